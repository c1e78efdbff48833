// bi-LSTM forward step with the WEIGHTS STATIONARY IN REGISTERS (bf16 engine, BASELINE.json configs[2]).
//
// Replaces the recurrence of model_v2.py:652-661, 694-823 (BasicLSTMCell under bidirectional_dynamic_rnn) for the
// shapes it is instantiated for; same arithmetic contract as lstm_step_fwd_bf16 (bf16 operands, fp32 accumulate,
// fp32 gates / c / h).
//
// Why: the tiled step kernel brings BOTH operands of z = [x_t | h_{t-1}] * kernel into LDS for every 256 x 128 output
// tile -- 1.2 GB of L2 -> LDS traffic per step at the metric shape, and a CU ingests ~65 GB/s.  Here a workgroup (one per
// CU, four waves, one per SIMD, 512 registers each) owns 32 NCT hidden units x 4 gates of ONE direction and keeps that
// slice of the kernel in its registers for the whole launch (K/2 registers per lane at NCT = 2: the B fragments of
// v_mfma_f32_32x32x16_bf16 for every k-step); only the activations stream:
//   * rows come in tiles of 32 sorted sequences, as 8 KB ring slots [32 rows][128 k] filled by LDS-DMA
//     (buffer_load_dwordx4 ... lds, XOR-swizzled source addresses, counted vmcnt + one s_barrier per slot), eleven
//     slots ahead of the MFMAs;
//   * every wave multiplies the SAME A fragments (ds_read_b128) with ITS columns: no B traffic at all;
//   * the tile's pre-activations go through a 33 KB LDS slab [32 rows][4 gates][units] and the gate math of tile i - 1
//     runs inside the k-loop of tile i, in the matrix pipe's shadow, every global access 16 bytes of a row
//     (4 rows x 256 B per wave-instruction);
//   * the bias rides in the GEMM: the input shadow carries TWO ones columns (in, in + 1) and the fragment-order weight
//     shadow holds bf16(bias) and bf16(bias - bf16(bias)) in those rows (relative error 2^-17): no bias registers.
// Operand traffic per step: the activations x (column blocks per direction) + one pass over the weights = ~0.4 GB at
// the metric shape, a third of the tiled kernel's.  Workgroups that share rows share an XCD (blockIdx & 7).
#include "gemm_bf16.h"
#include <type_traits>
#include "lstm_common.h"

namespace fvta {

// two floats -> packed bf16 pair (round to nearest even, v_cvt_pk_bf16_f32): a in the low half
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// (An EIGHT-wave form of this kernel -- two waves per SIMD of 256 registers, 32 columns each, nine weight fragments per
//  wave in LDS, ring of six slots -- was built and measured: 4.10 ms against 3.75 ms per text-cell forward; what it gains in
//  issue slots it loses to the shallow ring and the 8-wave barriers.  DESIGN.md appendix.)
// X3 (the split engine, precision bf16x3): every operand value is two bf16 terms (il32 layout, gemm_bf16.h).  A wave owns ONE
// column tile (NCT = 1: 8 units x 4 gates) and keeps its hi AND lo weight fragments -- the register budget of two column
// tiles -- a k-step reads a_hi and a_lo once and issues three MFMAs: hi hi into acc[0], hi lo + lo hi into acc[1] (summed
// when the tile's pre-activations go to the slab).  A 128-element ring slot row is 64 k of [hi 32 | lo 32] pairs: four
// k-steps per slot instead of eight.
template <int NX16_, int ND16_, int NCT_, bool X3_ = false>
struct WregCfg {
  static constexpr int NX16 = NX16_, ND16 = ND16_, NCT = NCT_, NW = 4;
  static constexpr bool X3 = X3_;
  static_assert(!X3_ || NCT_ == 1, "split engine: one column tile per wave (its hi and lo fragments fill the register file)");
  static constexpr int NWF = X3 ? 2 : NCT;            // weight fragments per k-step
  static constexpr int NP = X3 ? 3 : NCT;             // MFMAs per k-step = places for a gate stage
  static constexpr int NACC = X3 ? 2 : NCT;           // accumulators
  static constexpr int KS = X3 ? 4 : 8;               // k-steps per ring slot
  static constexpr int NK16 = NX16 + ND16;            // k-steps of 16
  static constexpr int NU = 8 * NCT, UB = NW * NU;    // units per wave / per workgroup
  static constexpr int RW = 32 / NW, DP = 8 / NW;     // rows of a tile whose gate math a wave does / DMA pieces per wave and slot
  static constexpr int D = 16 * ND16, IN_I = 16 * NX16;
  static constexpr int XM = X3 ? 2 : 1;               // stored terms per value: operand rows are XM times as long
  static constexpr int CB = D / UB;                   // column blocks per direction
  static constexpr int SX = (NX16 + KS - 1) / KS, SH = ND16 / KS, S = SX + SH;  // ring slots per row tile (x part, h part)
  static constexpr int XLAST = NX16 - KS * (SX - 1);  // k-steps of the last x slot (1 .. KS)
  // the ring holds RT whole tiles (so that a slot's LDS address is tile base + a compile-time offset); the DMA stream runs
  // LOOK slots ahead of the hand-over
  static constexpr int RT = S >= 12 ? 1 : 12 / S, RING = RT * S, LOOK = RING - 2;
  static constexpr int SLOT_ELEMS = 32 * 128;         // bf16 per slot: 32 rows x 128 elements
  static constexpr int TILE_ELEMS = S * SLOT_ELEMS;
  static constexpr int W_AGPR_FRAGS = (256 - 16 * NACC) / 4;  // weight fragments kept in AGPRs (beside the accumulators)
  static constexpr int ZS = 4 * UB + 4;               // floats per row of the pre-activation slab
  // the LAST W_LDS_FRAGS weight fragments of a wave live in LDS (the space the ring and the slab leave), read one k-step
  // ahead of their MFMAs: registers that the gate stages need at the widest shape
  static constexpr int W_FRAGS = NK16 * NWF;
  static constexpr int W_LDS_FRAGS = W_FRAGS > 84 ? (X3 ? 11 : 7) : 0, W_REG_FRAGS = W_FRAGS - W_LDS_FRAGS;
  static constexpr int Z_OFF = RING * SLOT_ELEMS * 2, WL_OFF = Z_OFF + 32 * ZS * 4;  // byte offsets of the slab / the weight tail
  static constexpr int LDS_BYTES = WL_OFF + NW * W_LDS_FRAGS * 1024;
  static_assert(LDS_BYTES <= 163840, "LDS");
  static constexpr int LPR = UB / 4, RPP = 64 / LPR, PASSES = RW / RPP;  // epilogue: lanes per row, rows per pass
  // k-step q of a tile: its ring slot, its position in the slot, its fragment buffer
  static constexpr int slot_of(int q) { return q < NX16 ? q / KS : SX + (q - NX16) / KS; }
  static constexpr int ks_of(int q) { return q < NX16 ? q % KS : (q - NX16) % KS; }
  // A fragments are read PF k-steps ahead of their MFMAs into NB = PF + 1 rotating buffers; the k-step sequence of a tile
  // is padded to NV, a multiple of NB (the padding steps read and multiply nothing), so that the rotation continues
  // seamlessly into the next tile's first PF k-steps
  static constexpr int PF = NX16 >= 3 ? 3 : 2, NB = PF + 1, NV = (NK16 + NB - 1) / NB * NB;
  static constexpr int buf_of(int q) { return q % NB; }
  // gate-math stages (see the kernel): per pass 1 read + 4 cells x CELL_STAGES + 1 store.  A k-step has NP places for a
  // stage (one behind each of its MFMAs); stage st runs at place HLO + st * HW / NSTAGES -- from the k-step after the
  // hand-over of slot 1 (every wave's slab write lies before it); the second pass's first slab read must come before the
  // hand-over of the next tile's slot 0 (k-step NV - PF, which precedes this tile's slab write)
  static constexpr int CELL_STAGES = 7, PASS_STAGES = 2 + 4 * CELL_STAGES, NSTAGES = PASSES * PASS_STAGES;
  static constexpr int QLO = SX > 1 ? KS : NX16, HLO = NP * QLO, HW = NP * NK16 - HLO;
  static constexpr int stage_begin(int h) { return h <= HLO ? 0 : (h >= NP * NK16 ? NSTAGES : ((h - HLO) * NSTAGES + HW - 1) / HW); }
  static constexpr int stage_place(int st) { return HLO + st * HW / NSTAGES; }
  // (the last slab read is cell 3's, requested at stage k = 2 of cell 2 of the last pass; shapes with too few k-steps for
  //  that put an extra barrier in front of the slab write instead)
  static constexpr bool SLAB_SAFE = stage_place((PASSES - 1) * PASS_STAGES + 1 + 2 * CELL_STAGES + 2) / NP <= NV - PF - 1;
  static_assert(ND16 % 8 == 0, "hidden size must be a multiple of 128");
  static_assert(D % UB == 0, "column blocks");
  static_assert(W_REG_FRAGS * 4 <= 400, "weight slice must fit the register file");
  static_assert(NX16 >= PF, "the first slot holds the k-steps read ahead across a tile boundary");
  static_assert(!X3 || NX16 % 2 == 0, "il32 groups: the input width is a multiple of 32");
  static_assert(PASSES >= 1 && PASSES <= 2 && LOOK >= 2 && RING <= 12, "geometry");
};

// ---- fragment-order weight shadow ---------------------------------------------------------------
// wf[cb][wave (NW)][ks][ct][lane][8]: the B fragment (32 columns x 16 k) of wave `wave` of column block `cb` for k-step ks,
// column tile ct, as ONE contiguous 1 KiB piece.  Column idx = 32 ct + (lane & 31) of the wave is gate idx / NU of unit
// cb UB + wave NU + idx % NU; k = 16 ks + 8 (lane >> 5) + e in the internal row order [x | 1 | 1 | 0.. | h]; rows `in`
// and `in + 1` hold the bias split in two bf16 terms.  One thread per (cb, wave, ks, ct, lane).
// X3: wf[cb][wave][ks][term][lane][8], ONE column tile per wave (NU = 8), term 0 = bf16(v), term 1 = bf16(v - bf16(v)); the
// bias sits in row `in` alone (hi and lo terms like every weight; the input's ones column there has hi = 1, lo = 0).
template <int NCT, bool X3 = false>
__global__ void cvt_weights_frag_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                        bf16_t* __restrict__ wf, int in, int in_i, int d, int nk16) {
  constexpr int NW = 4, NU = X3 ? 8 : 8 * NCT, UB = NW * NU;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int CB = d / UB;
  const size_t total = (size_t)CB * NW * nk16 * NCT * 64;
  if (gid >= total) return;
  const int lane = (int)(gid & 63);
  size_t r = gid >> 6;
  const int ct = (int)(r % NCT);   // X3: the term
  r /= NCT;
  const int ks = (int)(r % nk16);
  r /= nk16;
  const int wave = (int)(r % NW), cb = (int)(r / NW);
  const int idx = (X3 ? 0 : ct * 32) + (lane & 31);
  const int n = (idx / NU) * d + cb * UB + wave * NU + idx % NU;  // kernel column g d + u
  const int N4 = 4 * d;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * ks + 8 * (lane >> 5) + e;
    float v = 0.f;
    if (k < in)
      v = W[(size_t)k * N4 + n];
    else if (k >= in_i)
      v = W[(size_t)(in + k - in_i) * N4 + n];
    else if (k == in)
      v = X3 ? bias[n] : bf2f(f2bf(bias[n]));
    else if (k == in + 1)
      v = X3 ? 0.f : bias[n] - bf2f(f2bf(bias[n]));
    o[e] = (short)((X3 && ct == 1) ? f2bf(v - bf2f(f2bf(v))) : f2bf(v));
  }
  *reinterpret_cast<bf16x8*>(wf + gid * 8) = o;
}

// Diagnostics (-DFVTA_WREG_STAMP builds only, tools/r03_wreg_stamps.py): shader-clock sums of one workgroup's wave 0 at
// step t = 5: [0] kernel start, [1] after the weight load + first hand-over, [2] end, [3] tiles, [4] sum over hand-overs of
// the vmcnt wait, [5] of the barrier wait, [6] sum over tiles of the slab write section, [7] hand-overs, [8] DMA issue,
// [9] gate stages, [10] own_rows / prev_rows
__device__ unsigned long long g_wreg_stamps[64];
int wreg_read_stamp(int i, long long* v) {
  if (i < 0 || i >= 64) return FVTA_ERR_INVALID_ARG;
  unsigned long long x = 0;
  if (hipMemcpyFromSymbol(&x, HIP_SYMBOL(g_wreg_stamps), 8, (size_t)i * 8, hipMemcpyDeviceToHost) != hipSuccess) return FVTA_ERR_INVALID_ARG;
  *v = (long long)x;
  return FVTA_OK;
}
#ifdef FVTA_WREG_STAMP
#define WREG_CLOCK() __builtin_readcyclecounter()
#else
#define WREG_CLOCK() 0ull
#endif

// ---- the step kernel ------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(64 * C::NW, C::NW / 4) void lstm_fwd_wreg_bf16(StepArgs a, int RG) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  float* zs = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + C::Z_OFF);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA destinations and row bases are wave-uniform
  // workgroups are dealt round-robin over the 8 XCDs: the CB column blocks that stream the same rows share one
  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3;
  const int pair = xcd + 8 * (wslot / C::CB), cb = wslot % C::CB;
  const int dir = pair & 1, rg = pair >> 1;
  const int t = a.t;
  // compile-time ablations (timing experiments, -DFVTA_WREG_ABL=bits; results are garbage): 1 no gate stages, 2 no MFMAs,
  // 4 no activation DMA, 8 no weight load, 16 no stores, 32 no gate math (stores only), 64 no slab write
#ifdef FVTA_WREG_ABL
  constexpr int abl = FVTA_WREG_ABL;
#else
  constexpr int abl = 0;
#endif
  constexpr int d = C::D, IN_I = C::IN_I, NCT = C::NCT;
  const int nact = a.plan.nactive[t];
  const int ntiles = (nact + 31) >> 5;
  if (rg >= ntiles) return;
  const int nmine = (ntiles - rg + RG - 1) / RG;  // row tiles rg, rg + RG, ...
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  unsigned long long st_wait = 0, st_bar = 0, st_slab = 0, st_n = 0, st_issue = 0, st_stage = 0, st_rows = 0;
  const unsigned long long st_t0 = WREG_CLOCK();

  // ---- the weight slice: NK16 x NCT fragments, static indices only (registers)
  constexpr int NWF = C::NWF;  // weight fragments per k-step: the wave's column tiles, or (split engine) hi and lo of its one
  bf16x8_t w[C::NK16][NWF];
  f32x4* wlds = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + C::WL_OFF) + wave * C::W_LDS_FRAGS * 64 + lane;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.Wf[dir]) + (size_t)(cb * C::NW + wave) * C::NK16 * NWF * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < C::NK16; ++ks)
#pragma unroll
      for (int ct = 0; ct < NWF; ++ct) {
        Pack8 p;
        p.f = (abl & 8) ? f32x4{0.f, 0.f, 0.f, 0.f} : src[(ks * NWF + ct) * 64];
        if (ks * NWF + ct < C::W_REG_FRAGS)
          w[ks][ct] = p.b;
        else
          wlds[(ks * NWF + ct - C::W_REG_FRAGS) * 64] = p.f;  // (read back by this wave only)
      }
  }

  // ---- activation stream: per-lane source offsets of this wave's DP DMA pieces per slot (rows 4 DP wave .. )
  constexpr int XB = IN_I * 2 * C::XM, HB = d * 2 * C::XM;  // bytes per operand row (split engine: two terms per value)
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.xs + trow * (IN_I * C::XM), (unsigned)nact * XB);
  const __amdgpu_buffer_rsrc_t rh =
      make_rsrc(t > 0 ? a.hs + (trow - a.B) * (d * C::XM) : a.hs, t > 0 ? (unsigned)nact * HB : 0u);  // h_{-1} = 0
  unsigned voff_x[C::DP], voff_h[C::DP];
#pragma unroll
  for (int j = 0; j < C::DP; ++j) {
    const int r = 4 * (C::DP * wave + j) + (lane >> 4), c = (lane & 15) ^ (r & 15);
    voff_x[j] = (unsigned)r * XB + 16u * c;
    voff_h[j] = (unsigned)r * HB + 16u * c;
  }
  // slot cs of the workgroup's tile number `ord` -> its place in the ring (tile ord % RT)
  auto issue = [&](auto cs_c, int ord) {
    constexpr int cs = decltype(cs_c)::value;
    if constexpr ((abl & 4) != 0) return;
    // the row base goes into the VECTOR offset (scalar offset 0): rows past the active prefix, every tile past the
    // workgroup's last, and -- through its zero-record descriptor -- step 0's h_{-1} then fall off the descriptor and read
    // as zeros whichever way the range check treats a scalar offset
    const unsigned m0c = 32u * (unsigned)(rg + RG * ord);
    bf16_t* dst = smem + (ord % C::RT) * C::TILE_ELEMS + cs * C::SLOT_ELEMS + (C::DP * wave) * 512;
    if constexpr (cs < C::SX) {
      const unsigned rowb = m0c * XB + cs * 256;
#pragma unroll
      for (int j = 0; j < C::DP; ++j) {
        unsigned v = voff_x[j] + rowb;
        if constexpr (cs == C::SX - 1 && C::XLAST < C::KS) {  // the last x slot is narrower than 128 elements: its tail chunks read zeros
          const int r = 4 * (C::DP * wave + j) + (lane >> 4), c = (lane & 15) ^ (r & 15);
          v = (c < (16 / C::KS) * C::XLAST) ? v : GLDS_OOB;     // (16 / KS chunks per k-step; split engine: XLAST is even)
        }
        glds16(rx, dst + j * 512, v, 0);
      }
    } else {
      const unsigned rowb = m0c * HB + (cs - C::SX) * 256;
#pragma unroll
      for (int j = 0; j < C::DP; ++j) glds16(rh, dst + j * 512, voff_h[j] + rowb, 0);
    }
  };
  static_for<0, C::LOOK>([&](auto g_c) {  // slots 0 .. LOOK-1 of the stream
    constexpr int g = decltype(g_c)::value;
    issue(std::integral_constant<int, g % C::S>{}, g / C::S);
  });

  // ---- gate math of a tile, cut into STAGES of a few vector instructions each.  The stages of tile i - 1 are dealt over
  // the MFMAs of tile i (C::stage_begin), at most one behind each MFMA, so that they issue in its shadow (~24 free issue
  // cycles): the hand-written MFMA statements fix the instruction order, and a whole pass in one place would leave the
  // matrix pipe idle for its ~2,500 cycles.  lane = (row of the pass, four consecutive units).
  //   per pass: stage 0 reads cell 0's pre-activations from the slab, stages 1 + 7 e + k do cell e's sigmoids / tanhs
  //   (k = 0 .. 6; the next cell's pre-activations are requested at k = 2), the last stage stores c, h, the bf16 h shadow
  //   and the packed gates: every global access 16 bytes of a row.
  // What a tile's passes need from memory -- c_{t-1} and the output offsets of their rows -- is requested one whole tile
  // ahead (own_rows at the tile's first k-step): vmcnt retires in order, so a wait for a YOUNG load would drain the DMA
  // stream's look-ahead.
  const int e_rsub = lane / C::LPR, e_q = lane % C::LPR;
  const int u_lane = cb * C::UB + 4 * e_q;  // first of the lane's four units
  const float* cprev_base = a.cs ? (t > 0 ? a.cs + (trow - a.B) * (size_t)d : nullptr) : a.cstate + (size_t)dir * a.B * d;
  float* c_base = a.cs ? a.cs + trow * (size_t)d : a.cstate + (size_t)dir * a.B * d;
  f32x4 cp_next[C::PASSES], cp_cur[C::PASSES];
  int64_t oo_cur[C::PASSES];
  // The rows' output offsets travel with the previous cell states: ONE 8-byte vector load per pass and tile, issued a whole
  // tile before its use (own_rows).  (As wave-uniform scalar loads at the top of the tile that stores -- RPP per pass, selected
  // by row -- every wave sat out two scalar-cache round trips per tile behind an s_waitcnt lgkmcnt(0): 0.25 ms of the 3.8 ms
  // text-cell forward.)
  int64_t oo_next[C::PASSES];
  const int64_t* __restrict__ oo_g = a.plan.oo + trow;
  auto own_rows = [&](int m0t) {
#pragma unroll
    for (int p = 0; p < C::PASSES; ++p) {
      const int i = min(m0t + C::RW * wave + C::RPP * p + e_rsub, nact - 1);  // clamped: always a valid row
      oo_next[p] = oo_g[i];
      if (t > 0)
        cp_next[p] = *reinterpret_cast<const f32x4*>(cprev_base + (size_t)i * d + u_lane);
      else
        cp_next[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 cv, hv;
  unsigned gpk[8];
  f32x4 gf[C::X3 ? 4 : 1];  // split engine: the four cells' fp32 gates (i, j, f, o), saved unit-major
  f32x4 zg[4];  // the pass's pre-activations: gate g of the lane's four units (conflict-free 16-byte slab reads)
  float zjk, ti, tf, to, tj, ig, jg, fg, og, cc, te;
  auto read_pass = [&](auto p_c) {
    constexpr int p = decltype(p_c)::value;
    const float* zr = zs + (C::RW * wave + C::RPP * p + e_rsub) * C::ZS + 4 * e_q;
#pragma unroll
    for (int g = 0; g < 4; ++g) zg[g] = *reinterpret_cast<const f32x4*>(zr + g * C::UB);
  };
  // (the empty asm statements pin a stage's results to its place in the MFMA stream: the compiler would otherwise sink the
  //  whole computation down to the stores.  The sigmoid / tanh forms are fvta_sigmoid / fvta_tanh cut in two.)
  auto run_stage = [&](auto s_c, int m0p) {
    constexpr int st = decltype(s_c)::value, p = st / C::PASS_STAGES, r = st % C::PASS_STAGES;
    if constexpr (r == 0) {
      read_pass(std::integral_constant<int, p>{});
    } else if constexpr (r == C::PASS_STAGES - 1) {
      const int i = m0p + C::RW * wave + C::RPP * p + e_rsub;
      if constexpr ((abl & 4096) != 0) {  // timing experiment: the same stores into a small L2-resident region
        float* dump = reinterpret_cast<float*>(a.hs) + ((size_t)blockIdx.x * 4 + wave) * 2048 + lane * 4;
        st16(dump, cv, false);
        st16(dump + 256, hv, false);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(dump + 512 + lane * 2 - lane * 4) = u32x2{pk_bf16(hv[0], hv[1]), pk_bf16(hv[2], hv[3])};
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        st16(dump + 1024, __builtin_bit_cast(f32x4, u32x4{gpk[0], gpk[1], gpk[2], gpk[3]}), false);
        st16(dump + 1280, __builtin_bit_cast(f32x4, u32x4{gpk[4], gpk[5], gpk[6], gpk[7]}), false);
      } else
      if (((abl & 8192) || i < nact) && !(abl & 16)) {
        if constexpr (!(abl & 128)) st16(c_base + (size_t)i * d + u_lane, cv, a.nt != 0);
        const int64_t oo = oo_cur[p];
        if (((abl & 8192) || oo >= a.out_skip) && !(abl & 256)) {   // (inactive rows: -1; rows below out_skip: their readers take the shadow)
          float* o = a.out + oo + u_lane;
          if ((abl & 8192) || (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
            st16(o, hv, a.nt != 0);
          } else {  // an output row that is not 16-byte aligned
            o[0] = hv[0]; o[1] = hv[1]; o[2] = hv[2]; o[3] = hv[3];
          }
        }
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        if constexpr (C::X3) {
          // il32: the workgroup's 32 units are ONE group of the shadow row -- [hi 32 | lo 32] at element 2 cb UB
          bf16_t* hp = a.hs + (trow + i) * (size_t)(2 * d) + 2 * cb * C::UB + 4 * e_q;
          const unsigned h01 = pk_bf16(hv[0], hv[1]), h23 = pk_bf16(hv[2], hv[3]);
          const float l0 = hv[0] - __uint_as_float(h01 << 16), l1 = hv[1] - __uint_as_float(h01 & 0xffff0000u);
          const float l2 = hv[2] - __uint_as_float(h23 << 16), l3 = hv[3] - __uint_as_float(h23 & 0xffff0000u);
          *reinterpret_cast<u32x2*>(hp) = u32x2{h01, h23};
          *reinterpret_cast<u32x2*>(hp + 32) = u32x2{pk_bf16(l0, l1), pk_bf16(l2, l3)};
          if (a.gates) {  // fp32 gates, unit-major [u][i,j,f,o]: 64 bytes per lane
            float* gp = a.gates + (trow + i) * (size_t)(4 * d) + 4 * u_lane;
#pragma unroll
            for (int e = 0; e < 4; ++e) st16(gp + 4 * e, gf[e], a.nt != 0);
          }
        } else {
        if constexpr (!(abl & 512)) *reinterpret_cast<u32x2*>(a.hs + (trow + i) * (size_t)d + u_lane) = u32x2{pk_bf16(hv[0], hv[1]), pk_bf16(hv[2], hv[3])};
        }
        if (!C::X3 && a.gatesb && !(abl & 1024)) {  // unit-major [u][i,j,f,o]
          float* gp = reinterpret_cast<float*>(a.gatesb + (trow + i) * (size_t)(4 * d) + 4 * u_lane);
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          st16(gp, __builtin_bit_cast(f32x4, u32x4{gpk[0], gpk[1], gpk[2], gpk[3]}), a.nt != 0);
          st16(gp + 4, __builtin_bit_cast(f32x4, u32x4{gpk[4], gpk[5], gpk[6], gpk[7]}), a.nt != 0);
        }
      }
    } else {
      constexpr int e = (r - 1) / C::CELL_STAGES, k = (r - 1) % C::CELL_STAGES;
      if constexpr ((abl & 32) != 0) {
        if constexpr (k == 0) { cv[e] = zg[0][e]; hv[e] = zg[1][e]; gpk[2 * e] = __float_as_uint(zg[2][e]); gpk[2 * e + 1] = __float_as_uint(zg[3][e]); }
      } else
      if constexpr (k == 0) {
        ti = __expf(-zg[0][e]);
        tf = __expf(-(zg[2][e] + 1.0f));  // forget_bias
        asm volatile("" : "+v"(ti), "+v"(tf));
      } else if constexpr (k == 1) {
        to = __expf(-zg[3][e]);
        zjk = zg[1][e];
        tj = __expf(-2.0f * fabsf(zjk));
        asm volatile("" : "+v"(to), "+v"(tj), "+v"(zjk));
      } else if constexpr (k == 2) {
        ig = __builtin_amdgcn_rcpf(1.0f + ti);
        fg = __builtin_amdgcn_rcpf(1.0f + tf);
        asm volatile("" : "+v"(ig), "+v"(fg));
      } else if constexpr (k == 3) {
        og = __builtin_amdgcn_rcpf(1.0f + to);
        jg = copysignf((1.0f - tj) * __builtin_amdgcn_rcpf(1.0f + tj), zjk);
        asm volatile("" : "+v"(og), "+v"(jg));
      } else if constexpr (k == 4) {
        cc = cp_cur[p][e] * fg + ig * jg;
        te = __expf(-2.0f * fabsf(cc));
        asm volatile("" : "+v"(cc), "+v"(te));
        cv[e] = cc;
      } else if constexpr (k == 5) {
        float h = copysignf((1.0f - te) * __builtin_amdgcn_rcpf(1.0f + te), cc) * og;
        asm volatile("" : "+v"(h));
        hv[e] = h;
      } else if constexpr (C::X3) {
        gf[e] = f32x4{ig, jg, fg, og};
      } else {
        unsigned g0 = pk_bf16(ig, jg), g1 = pk_bf16(fg, og);
        asm volatile("" : "+v"(g0), "+v"(g1));
        gpk[2 * e] = g0;
        gpk[2 * e + 1] = g1;
      }
    }
  };

  // ---- accumulators (AGPRs) and the MFMA, by hand: the weight fragments live partly in AGPRs, partly in VGPRs, and the
  // matrix pipe reads either directly (the compiler's own allocation shuttles AGPR-resident operands through VGPRs, four
  // v_accvgpr_read per MFMA).  The asm statements are opaque to the hazard recogniser: an s_nop run covers the XDL write ->
  // VALU / LDS read distance before the accumulators are read.
  f32x16 acc[C::NACC];
  Pack8 wl[NWF];  // LDS-resident weight fragments of the NEXT k-step
  // place pl of k-step q: which weight fragment wi, which accumulator ai, and whether it opens the accumulator.
  //   bf16 engine: place = column tile (wi = ai = pl).   Split engine: place 0 = a_hi w_hi -> acc[0]; 1 = a_hi w_lo -> acc[1];
  //   2 = a_lo w_hi -> acc[1] (the caller hands the matching A fragment)
  auto mfma = [&](auto q_c, auto pl_c, const bf16x8_t afr) {
    constexpr int q = decltype(q_c)::value, pl = decltype(pl_c)::value;
    constexpr int wi = C::X3 ? (pl == 1 ? 1 : 0) : pl, ai = C::X3 ? (pl == 0 ? 0 : 1) : pl;
    constexpr bool opens = q == 0 && (!C::X3 || pl < 2);
    constexpr bool in_agpr = (q * NWF + wi) < C::W_AGPR_FRAGS;
    if constexpr ((abl & 2) != 0) return;
    if constexpr (q * NWF + wi >= C::W_REG_FRAGS) {
      if constexpr (opens)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[ai]) : "v"(afr), "v"(wl[wi].b));
      else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[ai]) : "v"(afr), "v"(wl[wi].b));
    } else if constexpr (opens) {
      if constexpr (in_agpr)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[ai]) : "v"(afr), "a"(w[q][wi]));
      else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[ai]) : "v"(afr), "v"(w[q][wi]));
    } else {
      if constexpr (in_agpr)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[ai]) : "v"(afr), "a"(w[q][wi]));
      else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[ai]) : "v"(afr), "v"(w[q][wi]));
    }
  };

  // ---- the k-step pipeline of a tile.  k-step q (of NK16) lies in ring slot slot_of(q); its A fragment is read PF (three)
  // k-steps ahead of its MFMAs (NB rotating fragment buffers), also across slot and tile boundaries: the hand-over of
  // slot g -- wait for its DMA, barrier, refill of the slot consumed two slots ago -- is therefore done while the MFMAs of
  // slot g - 1's last PF k-steps are still to come, and the matrix pipe never waits for a fresh LDS read.
  // ap[ks]: this lane's fragment address for position ks of a slot in the CURRENT tile's ring place (row l31, 16-byte
  // chunk (2 ks + hf) ^ (l31 & 15)); the slot is a compile-time offset on top.
  // split engine: position ks of a slot holds k-step ks's hi fragment at chunk 8 (ks / 2) + 2 (ks % 2) + hf of the row and its
  // lo fragment four chunks further (apl)
  const bf16_t* ap[C::KS];
  const bf16_t* apl[C::X3 ? C::KS : 1];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    const int ch = C::X3 ? 8 * (ks >> 1) + 2 * (ks & 1) + hf : 2 * ks + hf;
    ap[ks] = smem + l31 * 128 + ((ch ^ (l31 & 15)) << 3);
    if constexpr (C::X3) apl[ks] = smem + l31 * 128 + (((ch + 4) ^ (l31 & 15)) << 3);
  }
  Pack8 fr[C::NB];
  Pack8 frl[C::X3 ? C::NB : 1];  // split engine: the lo fragments
  auto handover = [&](auto s_c, int ord) {  // after it slot s_c of tile `ord` may be read
    constexpr int s = decltype(s_c)::value;
    const unsigned long long c0 = WREG_CLOCK();
    wait_vmcnt<C::DP * (C::LOOK - 1)>();  // every younger DMA piece may still fly; everything older has landed
    const unsigned long long c1 = WREG_CLOCK();
    __builtin_amdgcn_s_barrier();     // visible to all waves; the slot consumed two slots ago is free
    asm volatile("" ::: "memory");
    const unsigned long long c2 = WREG_CLOCK();
    issue(std::integral_constant<int, (s + C::LOOK) % C::S>{}, ord + (s + C::LOOK) / C::S);
    st_wait += c1 - c0;
    st_bar += c2 - c1;
    st_issue += WREG_CLOCK() - c2;
    st_n += 1;
  };

  handover(std::integral_constant<int, 0>{}, 0);
  const unsigned long long st_t1 = WREG_CLOCK();
#pragma unroll
  for (int q = 0; q < C::PF; ++q) {
    fr[C::buf_of(q)].f = *reinterpret_cast<const f32x4*>(ap[q]);
    if constexpr (C::X3) frl[C::buf_of(q)].f = *reinterpret_cast<const f32x4*>(apl[q]);
  }
  int prev_m0 = 1 << 30;  // no previous tile yet: the first tile's stages run on an undefined slab and store nothing
#pragma unroll
  for (int p = 0; p < C::PASSES; ++p) {
    cp_cur[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    oo_cur[p] = -1;
  }
  for (int it = 0; it < nmine; ++it) {
    const int m0 = 32 * (rg + RG * it);
    // ring place of the next tile relative to this one (elements)
    const int tb_delta = (((it + 1) % C::RT) - (it % C::RT)) * C::TILE_ELEMS;
    auto kstep = [&](auto q_c) {
      constexpr int q = decltype(q_c)::value;
      constexpr int n = q + C::PF;
      if constexpr (n < C::NK16) {
        if constexpr (C::ks_of(n) == 0) handover(std::integral_constant<int, C::slot_of(n)>{}, it);
        fr[C::buf_of(n)].f = *reinterpret_cast<const f32x4*>(ap[C::ks_of(n)] + C::slot_of(n) * C::SLOT_ELEMS);
        if constexpr (C::X3) frl[C::buf_of(n)].f = *reinterpret_cast<const f32x4*>(apl[C::ks_of(n)] + C::slot_of(n) * C::SLOT_ELEMS);
      } else if constexpr (n >= C::NV) {  // the next tile's first k-steps (past the workgroup's last tile: zeros nobody uses)
        if constexpr (n == C::NV) handover(std::integral_constant<int, 0>{}, it + 1);
        fr[C::buf_of(n - C::NV)].f = *reinterpret_cast<const f32x4*>(ap[n - C::NV] + tb_delta);
        if constexpr (C::X3) frl[C::buf_of(n - C::NV)].f = *reinterpret_cast<const f32x4*>(apl[n - C::NV] + tb_delta);
      }
      if constexpr (q >= C::NK16) return;  // a padding step of the fragment rotation
      else {
      if constexpr (q == 0) {
        const unsigned long long r0 = WREG_CLOCK();
        own_rows(m0);
        st_rows += WREG_CLOCK() - r0;
      }
      // the k-step's NP MFMAs (step 0: the h slots hold zeros), the gate stages dealt to each place right behind it
      static_for<0, C::NP>([&](auto pl_c) {
        constexpr int pl = decltype(pl_c)::value;
        if constexpr (C::X3 && pl == 2)
          mfma(q_c, pl_c, frl[C::buf_of(q)].b);
        else
          mfma(q_c, pl_c, fr[C::buf_of(q)].b);
        if constexpr (C::stage_begin(C::NP * q + pl) < C::stage_begin(C::NP * q + pl + 1)) {
          const unsigned long long g0 = WREG_CLOCK();
          if constexpr (!(abl & 1)) static_for<C::stage_begin(C::NP * q + pl), C::stage_begin(C::NP * q + pl + 1)>([&](auto s_c) { run_stage(s_c, prev_m0); });
          st_stage += WREG_CLOCK() - g0;
        }
      });
      }
    };
    static_for<0, C::NV>([&](auto q_c) {
      kstep(q_c);
      constexpr int q1 = decltype(q_c)::value + 1;
      if constexpr (q1 < C::NK16) {  // LDS-resident weight fragments of the next k-step
#pragma unroll
        for (int ct = 0; ct < NWF; ++ct)
          if (q1 * NWF + ct >= C::W_REG_FRAGS) wl[ct].f = wlds[(q1 * NWF + ct - C::W_REG_FRAGS) * 64];
      }
    });
    // the tile's pre-activations -> slab [row][gate][unit of the workgroup]
    const unsigned long long sl0 = WREG_CLOCK();
    if constexpr (!C::SLAB_SAFE) {  // few k-steps: the previous tile's last slab reads may not lie before the last hand-over
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if constexpr (C::NACC > 1)
      asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[0]), "+a"(acc[C::NACC - 1]));
    else
      asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[0]));
    if constexpr (!(abl & 64)) {
    if constexpr (C::X3) {  // one column tile: hi hi + (hi lo + lo hi)
      float* zc = zs + (l31 / C::NU) * C::UB + wave * C::NU + l31 % C::NU;
#pragma unroll
      for (int r = 0; r < 16; ++r) zc[((r & 3) + 8 * (r >> 2) + 4 * hf) * C::ZS] = acc[0][r] + acc[1][r];
    } else {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int idx = ct * 32 + l31;
      float* zc = zs + (idx / C::NU) * C::UB + wave * C::NU + idx % C::NU;
#pragma unroll
      for (int r = 0; r < 16; ++r) zc[((r & 3) + 8 * (r >> 2) + 4 * hf) * C::ZS] = acc[ct][r];
    }
    }
    }
    st_slab += WREG_CLOCK() - sl0;
    prev_m0 = m0;
#pragma unroll
    for (int p = 0; p < C::PASSES; ++p) {
      cp_cur[p] = cp_next[p];
      oo_cur[p] = oo_next[p];
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      ap[ks] += tb_delta;
      if constexpr (C::X3) apl[ks] += tb_delta;
    }
  }
  // ---- the last tile's gate math
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  static_for<0, C::NSTAGES>([&](auto s_c) { run_stage(s_c, prev_m0); });
  wait_vmcnt<0>();  // the ring's trailing DMA pieces must not outlive the workgroup's LDS allocation
#ifdef FVTA_WREG_STAMP
  if (t == 5 && (int)blockIdx.x == a.dbg && tid == 0) {
    g_wreg_stamps[0] = st_t0; g_wreg_stamps[1] = st_t1; g_wreg_stamps[2] = WREG_CLOCK(); g_wreg_stamps[3] = nmine;
    g_wreg_stamps[4] = st_wait; g_wreg_stamps[5] = st_bar; g_wreg_stamps[6] = st_slab; g_wreg_stamps[7] = st_n;
    g_wreg_stamps[8] = st_issue; g_wreg_stamps[9] = st_stage; g_wreg_stamps[10] = st_rows;
  }
#endif
}

// ---- host side --------------------------------------------------------------------------------------
template <class K>
static void wreg_allow_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

static int wreg_cus() {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  return cus;
}

// waves per workgroup of the configuration a shape runs on
// column tiles per wave for a shape, 0: not built
// FVTA_LSTM_WREG (A/B measurements): bit 0 the weights-stationary forward, bit 1 the weights-stationary backward of steps
// with few rows, bit 2 the pipelined weights-stationary backward (d = 512, every row count); default 7, 0: the tiled step
// kernels for every shape
static int g_wreg_override = -1;  // fvta_lstm_kernel_select (tests, A/B measurements)
int wreg_mode() {
  static const int env_mode = [] {
    const char* e = getenv("FVTA_LSTM_WREG");
    return (e && e[0] >= '0' && e[0] <= '7') ? e[0] - '0' : 7;
  }();
  return g_wreg_override >= 0 ? g_wreg_override : env_mode;
}
int wreg_set_mode(int mode) {
  const int prev = wreg_mode();
  g_wreg_override = mode < 0 ? -1 : (mode & 7);
  return prev;
}

int wreg_nct(int in_i, int d) {
  if (!(wreg_mode() & 1)) return 0;
  const int nx = in_i / 16, nd = d / 16;
  if (in_i % 16 || d % 128) return 0;
  if (nd == 32 && (nx == 14 || nx == 8)) return 2;
  if (nd == 64 && (nx == 14 || nx == 8)) return 1;
  if (nd == 8 && (nx == 8 || nx == 2)) return 2;
  return 0;
}
// the split engine's shapes (one column tile per wave; d = 1024 does not fit: 156 fragments)
bool wreg_x3_built(int in_i, int d) {
  if (!(wreg_mode() & 1) || in_i % 32 || d % 128) return false;
  const int nx = in_i / 16, nd = d / 16;
  return (nd == 32 && (nx == 14 || nx == 8)) || (nd == 8 && (nx == 8 || nx == 2));
}

void launch_cvt_weights_frag(const float* W, const float* bias, bf16_t* wf, int in, int in_i, int d, int xm, hipStream_t s) {
  const int nk16 = (in_i + d) / 16;
  if (xm == 2) {
    if (!wreg_x3_built(in_i, d)) return;
    const size_t total = (size_t)d * nk16 * 8 * 2;  // one thread per 8 weights of the [4d][K] kernel, two terms
    hipLaunchKernelGGL((cvt_weights_frag_kernel<2, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, bias, wf, in, in_i, d, nk16);
    return;
  }
  const int nct = wreg_nct(in_i, d);
  if (!nct) return;
  const size_t total = (size_t)d * nk16 * 8;  // one thread per 8 weights of the [4d][K] kernel
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (nct == 2)
    hipLaunchKernelGGL(cvt_weights_frag_kernel<2>, dim3(grid), dim3(256), 0, s, W, bias, wf, in, in_i, d, nk16);
  else
    hipLaunchKernelGGL(cvt_weights_frag_kernel<1>, dim3(grid), dim3(256), 0, s, W, bias, wf, in, in_i, d, nk16);
}

template <class C>
static void launch_wreg(const StepArgs& a, hipStream_t s) {
  wreg_allow_lds(lstm_fwd_wreg_bf16<C>, C::LDS_BYTES);
  // row groups: as many as fill the CUs, (2 directions x RG) a multiple of 8 (one (direction, row group) pair per XCD
  // slot), and no more than there are row tiles
  int rg = wreg_cus() / (2 * C::CB);
  rg = rg / 4 * 4;
  if (rg < 4) rg = 4;
  const int tiles = (a.B + 31) / 32;
  while (rg > 4 && rg - 4 >= tiles) rg -= 4;
  hipLaunchKernelGGL(lstm_fwd_wreg_bf16<C>, dim3(2 * rg * C::CB), dim3(64 * C::NW), C::LDS_BYTES, s, a, rg);
}

bool launch_step_fwd_wreg(const StepArgs& a, hipStream_t s) {
  const int in_i = a.Kp - a.d;
  if (a.xm == 2) {  // the split engine
    if (!wreg_x3_built(in_i, a.d) || !a.Wf[0] || !a.hs) return false;
    const int nx = in_i / 16, nd = a.d / 16;
    if (nd == 32 && nx == 14) launch_wreg<WregCfg<14, 32, 1, true>>(a, s);
    else if (nd == 32 && nx == 8) launch_wreg<WregCfg<8, 32, 1, true>>(a, s);
    else if (nd == 8 && nx == 8) launch_wreg<WregCfg<8, 8, 1, true>>(a, s);
    else if (nd == 8 && nx == 2) launch_wreg<WregCfg<2, 8, 1, true>>(a, s);
    else return false;
    return true;
  }
  const int nct = wreg_nct(in_i, a.d);
  if (!nct || !a.Wf[0] || !a.hs) return false;
  const int nx = in_i / 16, nd = a.d / 16;
  if (nd == 32 && nx == 14) launch_wreg<WregCfg<14, 32, 2>>(a, s);
  else if (nd == 32 && nx == 8) launch_wreg<WregCfg<8, 32, 2>>(a, s);
  else if (nd == 64 && nx == 14) launch_wreg<WregCfg<14, 64, 1>>(a, s);
  else if (nd == 64 && nx == 8) launch_wreg<WregCfg<8, 64, 1>>(a, s);
  else if (nd == 8 && nx == 8) launch_wreg<WregCfg<8, 8, 2>>(a, s);
  else if (nd == 8 && nx == 2) launch_wreg<WregCfg<2, 8, 2>>(a, s);
  else return false;
  return true;
}

}  // namespace fvta
