// bi-LSTM forward step, weights stationary in registers, TWO WAVES PER SIMD (bf16 engine, d = 512).
//
// Same arithmetic contract and data layout as lstm_fwd_wreg_bf16 (lstm_wreg.hip; model_v2.py:652-661, 694-823:
// BasicLSTMCell under bidirectional_dynamic_rnn): bf16 operands, fp32 accumulate, fp32 gates / c / h, the bias as two
// ones columns of the input shadow.
//
// Why a second form: with ONE wave per SIMD (the 512-register kernel) every vector-memory instruction stalls the SIMD's only
// instruction stream for ~120 cycles (tools/probes/store_issue_probe.hip), and a wave issues 24 of them per row tile:
// 2,900 cycles of issue stall beside 2,944 cycles of matrix time.  A wave beside a stalled partner runs the matrix pipe at
// full rate (tools/probes/spec_probe.hip) -- but at 256 registers a wave cannot hold 46 B fragments of a 32-column slice.
// Here the TWO WAVES OF A SIMD SHARE 32 COLUMNS AND SPLIT K: wave (cg, kh) of the workgroup's eight owns column group cg
// (8 hidden units x 4 gates) and the k-steps 2j + kh -- NK16 / 2 fragments (92 registers at K = 736) in AGPRs, nothing in
// LDS.  The workgroup owns 32 units (CB = 16 column blocks per direction: the activations stream twice as often as in the
// four-wave form, all L2 hits: workgroups that share rows share an XCD).
//   * ring, hand-over, swizzle: as in lstm_wreg.hip (one DMA piece per wave and slot); both waves of a pair consume every
//     slot, each its own k-steps;
//   * the two partial pre-activation tiles go to TWO slabs [32 rows][4 gates][32 units]; the gate math adds them;
//   * gate math of a row tile: ONE pass of 8 rows x 32 units per wave, done by the four waves whose kh equals the tile's
//     parity (row-contiguous 16-byte accesses, as many bytes per vector-memory instruction as the four-wave form), in
//     stages dealt over the MFMAs of the next tile.  The tile loop is unrolled by two and the whole body instantiated per
//     kh, so that duty, ring place and every load are compile-time unconditional (a load under a run-time condition costs a
//     vmcnt(0): DESIGN.md section 0).
#pragma clang diagnostic ignored "-Wunused-lambda-capture"
#include "gemm_bf16.h"
#include <type_traits>
#include "lstm_common.h"

namespace fvta {
namespace w8 {

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

template <int NX16_, int ND16_>
struct Cfg {
  static constexpr int NX16 = NX16_, ND16 = ND16_, NW = 8, NCG = 4;
  static constexpr int NK16 = NX16 + ND16, NKW = NK16 / 2;  // k-steps of 16: all / per wave
  static constexpr int NU = 8, UB = NCG * NU;               // units per column group / per workgroup
  static constexpr int D = 16 * ND16, IN_I = 16 * NX16, CB = D / UB;
  static constexpr int SX = (NX16 + 7) / 8, SH = ND16 / 8, S = SX + SH;
  static constexpr int XLAST = NX16 - 8 * (SX - 1);
  static constexpr int RT = 2, NP = 2 * S;                   // ring places (whole tiles) / DMA pieces per issuing wave and tile
  static constexpr int SLOT_ELEMS = 32 * 128, TILE_ELEMS = S * SLOT_ELEMS;
  static constexpr int ZS = 4 * UB + 4;                      // floats per slab row
  static constexpr int Z_OFF = RT * TILE_ELEMS * 2, SLAB_FLOATS = 32 * ZS;
  static constexpr int LDS_BYTES = Z_OFF + 2 * SLAB_FLOATS * 4;
  // wave step j = the k-steps 2j (kh = 0) and 2j + 1 (kh = 1): same ring slot, neighbouring positions
  static constexpr int slot_of(int j) { return 2 * j < NX16 ? (2 * j) / 8 : SX + (2 * j - NX16) / 8; }
  static constexpr int ke_of(int j) { return (2 * j < NX16 ? (2 * j) % 8 : (2 * j - NX16) % 8) / 2; }  // position pair in the slot
  static constexpr int PF = 3, NB = PF + 1, NV = (NKW + NB - 1) / NB * NB;
  // gate stages: 1 slab read + 4 cells x 7 + 1 store, behind the MFMAs of wave steps JLO .. NKW - 1 of the NEXT tile (the
  // tile barrier lies between every wave's slab write and the read)
  static constexpr int CELL_STAGES = 7, NSTAGES = 2 + 4 * CELL_STAGES, JLO = 0, HW = NKW - JLO;
  static constexpr int stage_begin(int j) { return j <= JLO ? 0 : (j >= NKW ? NSTAGES : ((j - JLO) * NSTAGES + HW - 1) / HW); }
  static_assert(NX16 % 2 == 0, "x part: whole wave steps");
  static_assert(NP <= NKW, "one DMA piece per wave step");
  static_assert(ND16 % 8 == 0, "hidden size must be a multiple of 128");
  static_assert(D % UB == 0 && LDS_BYTES <= 163840, "geometry");
  static_assert(NKW * 4 + 16 <= 128, "weight slice + accumulator in the AGPR half");
};

template <class C, int KH>
__device__ __forceinline__ void body(const StepArgs& a, int RG, bf16_t* smem) {
  float* zs = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + C::Z_OFF);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave & 3;
  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3;
  const int pair = xcd + 8 * (wslot / C::CB), cb = wslot % C::CB;
  const int dir = pair & 1, rg = pair >> 1;
  const int t = a.t;
  constexpr int d = C::D, IN_I = C::IN_I;
  // compile-time ablations (timing experiments, -DFVTA_W8_ABL=bits; results are garbage): 1 no gate stages, 2 no MFMAs,
  // 4 no activation DMA, 16 no stores, 64 no slab write
#ifdef FVTA_W8_ABL
  constexpr int abl = FVTA_W8_ABL;
#else
  constexpr int abl = 0;
#endif
  const int nact = a.plan.nactive[t];
  const int ntiles = (nact + 31) >> 5;
  if (rg >= ntiles) return;
  const int nmine = (ntiles - rg + RG - 1) / RG;
  const int npairs = (nmine + 1) >> 1;  // (an odd count: the last tile of the last pair lies past the rows -- reads zeros, stores nothing)
  const size_t trow = ((size_t)dir * a.J + t) * a.B;

  // ---- the wave's weight slice: fragments of the k-steps 2j + KH of column group cg (cvt_weights_frag_kernel<1> layout)
  bf16x8_t w[C::NKW];
  {
    const f32x4 __attribute__((address_space(1)))* src =
        (const f32x4 __attribute__((address_space(1)))*)(uintptr_t)a.Wf[dir] + (size_t)(cb * C::NCG + cg) * C::NK16 * 64 + lane;
#pragma unroll
    for (int j = 0; j < C::NKW; ++j) {
      Pack8 p;
      p.f = src[(2 * j + KH) * 64];
      w[j] = p.b;
    }
  }

  // ---- activation stream: the ring holds TWO WHOLE TILES; while tile i is multiplied, the waves that are off gate duty (one
  // per SIMD: kh == parity of i) bring in tile i + 1 -- two pieces (4 rows x 256 B each: rows 8 cg .. 8 cg + 7) per slot, one
  // piece per wave step -- and ONE barrier per tile hands over everything: tile i + 1 landed (the issuers wait for their own
  // pieces), the slab written, tile i's place free.  (Per-slot hand-overs, as in the four-wave kernel, measured 4.0 ms per
  // text-cell forward against its 3.5: six eight-wave barriers per tile, each waiting for the wave on gate duty.)
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.xs + trow * IN_I, (unsigned)nact * IN_I * 2);
  const __amdgpu_buffer_rsrc_t rh = make_rsrc(t > 0 ? a.hs + (trow - a.B) * d : a.hs, t > 0 ? (unsigned)nact * d * 2 : 0u);
  unsigned voff_x[2], voff_h[2];
  int dma_c[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = 8 * cg + 4 * h + (lane >> 4);
    dma_c[h] = (lane & 15) ^ (r & 15);
    voff_x[h] = (unsigned)r * (IN_I * 2) + 16u * dma_c[h];
    voff_h[h] = (unsigned)r * (d * 2) + 16u * dma_c[h];
  }
  auto issue = [&](auto p_c, auto place_c, int ord) {  // piece p_c (slot p / 2, row half p % 2) of tile `ord` into ring place place_c
    constexpr int cs = decltype(p_c)::value / 2, h = decltype(p_c)::value % 2, place = decltype(place_c)::value;
    if constexpr ((abl & 4) != 0) return;
    const unsigned m0c = 32u * (unsigned)(rg + RG * ord);
    bf16_t* dst = smem + place * C::TILE_ELEMS + cs * C::SLOT_ELEMS + (2 * cg + h) * 512;
    if constexpr (cs < C::SX) {
      unsigned v = voff_x[h] + m0c * (IN_I * 2) + cs * 256;
      if constexpr (cs == C::SX - 1 && C::XLAST < 8) v = (dma_c[h] < 2 * C::XLAST) ? v : GLDS_OOB;
      glds16(rx, dst, v, 0);
    } else {
      glds16(rh, dst, voff_h[h] + m0c * (d * 2) + (cs - C::SX) * 256, 0);
    }
  };
  if constexpr (KH == 1) {  // tile 0 (as if a tile -1 of parity 1 had run)
    static_for<0, C::NP>([&](auto p_c) { issue(p_c, std::integral_constant<int, 0>{}, 0); });
    wait_vmcnt<0>();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- gate math (the duty waves of a tile: kh == tile parity; wave cg does rows 8 cg .. 8 cg + 7, lane = (row, 4 units))
  const int e_rsub = lane >> 3, e_q = lane & 7;
  const int u_lane = cb * C::UB + 4 * e_q;
  float* c_base = a.cs ? a.cs + trow * (size_t)d : a.cstate + (size_t)dir * a.B * d;
  // (a wave is on duty every other tile: what own_rows requests at the head of a duty tile is used by the stages of the NEXT
  //  tile and replaced a tile after that -- one copy is enough)
  // c_{t-1} is loaded UNCONDITIONALLY (step 0: from a valid address, this step's own slab) and masked to zero at its use:
  // under `t > 0 ? load : 0` the compiler branches around the load and waits vmcnt(0) inside the branch
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t cp_cur = {0u, 0u, 0u, 0u};
  int64_t oo_cur = -1;
  const float* cp_src = a.cs ? (t > 0 ? a.cs + (trow - a.B) * (size_t)d : c_base) : c_base;
  const unsigned cp_mask = t > 0 ? 0xffffffffu : 0u;
  const int64_t* __restrict__ oo_g = a.plan.oo + trow;
  auto own_rows = [&](int m0t) {
    const int i = min(m0t + 8 * cg + e_rsub, nact - 1);
    oo_cur = oo_g[i];
    cp_cur = *(const u32x4_t __attribute__((address_space(1)))*)(uintptr_t)(cp_src + (size_t)i * d + u_lane);
  };
  f32x4 cv, hv;
  unsigned gpk[8];
  f32x4 zg[4];
  float zjk, ti, tf, to, tj, ig, jg, fg, og, cc, te;
  auto run_stage = [&](auto s_c, int m0p) {
    constexpr int st = decltype(s_c)::value;
    if constexpr (st == 0) {
      const float* zr = zs + (8 * cg + e_rsub) * C::ZS + 4 * e_q;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        zg[g] = *reinterpret_cast<const f32x4*>(zr + g * C::UB) + *reinterpret_cast<const f32x4*>(zr + C::SLAB_FLOATS + g * C::UB);
    } else if constexpr (st == C::NSTAGES - 1) {
      const int i = m0p + 8 * cg + e_rsub;
      if (i < nact && !(abl & 16)) {
        st16(c_base + (size_t)i * d + u_lane, cv, a.nt != 0);
        const int64_t oo = oo_cur;
        if (oo >= a.out_skip) {
          float* o = a.out + oo + u_lane;
          if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) {
            st16(o, hv, a.nt != 0);
          } else {
            o[0] = hv[0]; o[1] = hv[1]; o[2] = hv[2]; o[3] = hv[3];
          }
        }
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(a.hs + (trow + i) * (size_t)d + u_lane) = u32x2{pk_bf16(hv[0], hv[1]), pk_bf16(hv[2], hv[3])};
        if (a.gatesb) {
          float* gp = reinterpret_cast<float*>(a.gatesb + (trow + i) * (size_t)(4 * d) + 4 * u_lane);
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          st16(gp, __builtin_bit_cast(f32x4, u32x4{gpk[0], gpk[1], gpk[2], gpk[3]}), a.nt != 0);
          st16(gp + 4, __builtin_bit_cast(f32x4, u32x4{gpk[4], gpk[5], gpk[6], gpk[7]}), a.nt != 0);
        }
      }
    } else {
      constexpr int e = (st - 1) / C::CELL_STAGES, k = (st - 1) % C::CELL_STAGES;
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
        cc = __uint_as_float(cp_cur[e] & cp_mask) * fg + ig * jg;
        te = __expf(-2.0f * fabsf(cc));
        asm volatile("" : "+v"(cc), "+v"(te));
        cv[e] = cc;
      } else if constexpr (k == 5) {
        float h = copysignf((1.0f - te) * __builtin_amdgcn_rcpf(1.0f + te), cc) * og;
        asm volatile("" : "+v"(h));
        hv[e] = h;
      } else {
        unsigned g0 = pk_bf16(ig, jg), g1 = pk_bf16(fg, og);
        asm volatile("" : "+v"(g0), "+v"(g1));
        gpk[2 * e] = g0;
        gpk[2 * e + 1] = g1;
      }
    }
  };

  // ---- accumulator and MFMA by hand (weights and accumulator in AGPRs, read by the matrix pipe directly)
  f32x16 acc;
  auto mfma = [&acc, &w](auto j_c, const bf16x8_t afr) {  // (named captures: asm operands alone do not capture)
    constexpr int j = decltype(j_c)::value;
    if constexpr ((abl & 2) != 0) return;
    if constexpr (j == 0)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(afr), "a"(w[j]));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(afr), "a"(w[j]));
  };

  // ap[i]: this lane's fragment offset for position pair i of a slot: row l31, 16-byte chunk (2 (2 i + KH) + hf) ^ (l31 & 15)
  const bf16_t* ap[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ap[i] = smem + l31 * 128 + (((2 * (2 * i + KH) + hf) ^ (l31 & 15)) << 3);
  Pack8 fr[C::NB];
  int prev_m0 = 1 << 30;  // no previous tile yet: its stages run on an undefined slab and store nothing

  auto tile = [&](auto par_c, int ord) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool issuer = (KH == PAR);    // off gate duty in this tile: brings in the next tile, requests its own rows' c_{t-1}
    constexpr bool duty_now = (KH != PAR);  // does the gate math of the previous tile
    const int m0 = 32 * (rg + RG * ord);
#pragma unroll
    for (int j = 0; j < C::PF; ++j)
      fr[j % C::NB].f = *reinterpret_cast<const f32x4*>(ap[C::ke_of(j)] + PAR * C::TILE_ELEMS + C::slot_of(j) * C::SLOT_ELEMS);
    static_for<0, C::NKW>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value, n = j + C::PF;
      if constexpr (n < C::NKW)
        fr[n % C::NB].f = *reinterpret_cast<const f32x4*>(ap[C::ke_of(n)] + PAR * C::TILE_ELEMS + C::slot_of(n) * C::SLOT_ELEMS);
      if constexpr (issuer) {
        if constexpr (j == 0) own_rows(m0);
        if constexpr (j < C::NP) issue(j_c, std::integral_constant<int, 1 - PAR>{}, ord + 1);
      }
      mfma(j_c, fr[j % C::NB].b);
      if constexpr (duty_now && !(abl & 1) && C::stage_begin(j) < C::stage_begin(j + 1))
        static_for<C::stage_begin(j), C::stage_begin(j + 1)>([&](auto s_c) { run_stage(s_c, prev_m0); });
    });
    // the wave's partial pre-activations -> its K-half's slab [row][gate][unit of the workgroup]
    asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc));
    if constexpr (!(abl & 64)) {
      float* zc = zs + KH * C::SLAB_FLOATS + (l31 / C::NU) * C::UB + cg * C::NU + l31 % C::NU;
#pragma unroll
      for (int r = 0; r < 16; ++r) zc[((r & 3) + 8 * (r >> 2) + 4 * hf) * C::ZS] = acc[r];
    }
    prev_m0 = m0;
    // ---- the tile barrier: the next tile has landed, the slab is written, this tile's place is free
    if constexpr (issuer) wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  for (int ip = 0; ip < npairs; ++ip) {
    tile(std::integral_constant<int, 0>{}, 2 * ip);
    tile(std::integral_constant<int, 1>{}, 2 * ip + 1);
  }
  // ---- the last tile's gate math (parity 1)
  if constexpr (KH == 1 && !(abl & 1)) static_for<0, C::NSTAGES>([&](auto s_c) { run_stage(s_c, prev_m0); });
  wait_vmcnt<0>();  // the ring's trailing DMA pieces must not outlive the workgroup's LDS allocation
}

}  // namespace w8

template <class C>
__global__ __launch_bounds__(512) void lstm_fwd_wreg8_bf16(StepArgs a, int RG) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8))
    w8::body<C, 1>(a, RG, smem);
  else
    w8::body<C, 0>(a, RG, smem);
}

static int w8_cus() {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  return cus;
}

template <class C>
static void launch_w8(const StepArgs& a, hipStream_t s) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_wreg8_bf16<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  int rg = w8_cus() / (2 * C::CB);
  rg = rg / 4 * 4;
  if (rg < 4) rg = 4;
  const int tiles = (a.B + 31) / 32;
  while (rg > 4 && rg - 4 >= tiles) rg -= 4;
  hipLaunchKernelGGL(lstm_fwd_wreg8_bf16<C>, dim3(2 * rg * C::CB), dim3(512), C::LDS_BYTES, s, a, rg);
}

// shapes the K-split form is built for (their weight shadow is in cvt_weights_frag_kernel<1> order: wreg_nct)
bool wreg8_shape(int in_i, int d) {
  const int nx = in_i / 16, nd = d / 16;
  return in_i % 16 == 0 && d % 128 == 0 && nd == 32 && (nx == 14 || nx == 8);
}

bool launch_step_fwd_wreg8(const StepArgs& a, hipStream_t s) {
  const int in_i = a.Kp - a.d;
  if (!wreg8_shape(in_i, a.d) || !a.Wf[0] || !a.hs) return false;
  if (in_i / 16 == 14)
    launch_w8<w8::Cfg<14, 32>>(a, s);
  else
    launch_w8<w8::Cfg<8, 32>>(a, s);
  return true;
}

}  // namespace fvta
