// bi-LSTM backward step with the RECURRENT WEIGHTS STATIONARY IN REGISTERS (bf16 engine), for steps with few rows.
//
// Same arithmetic contract as lstm_bwd_fused_bf16 (model_v2.py:652-661, 694-823 differentiated: dh_t = d_out_t +
// dz_{t+1} * Wh^T, then the gate gradient; bf16 operands, fp32 accumulate, fp32 gate math, dz written as bf16 unit-major).
//
// Why: the tiled kernel's launch lasts as long as ONE workgroup's chain of 4d/32 k-tiles plus its epilogue, however few
// rows there are -- the photo cell (2,560 rows at the metric shape, 40 steps) and the short steps of ragged batches sit on
// that chain (55-100 us per launch).  Here a workgroup (one per CU, four waves, 512 registers each) owns 32 NCT hidden
// units of one direction and keeps ITS Wh columns in registers for the whole launch: wave w holds the B fragments of
// v_mfma_f32_32x32x16_bf16 for the quarter [w d, (w + 1) d) of the 4d-long dz row -- d NCT / 4 registers per lane, the
// whole AGPR file at d = 512 -- and multiplies row tiles of 32 sequences whose dz_{t+1} rows come straight from global
// memory into the A fragments (16 bytes per lane and k-step: the row-major dz row IS the fragment order).  The four
// partial tiles meet in LDS (one barrier per row tile, double buffered), then every thread does the gate gradient of
// 4 NCT units of one row with 16-byte accesses.  dz_{t+1} is read once per column block (d / (32 NCT) times) -- that
// is why the tiled kernel keeps the steps with many rows (launch_bwd_wreg's row limit).
#include "gemm_bf16.h"
#include <type_traits>
#include "lstm_common.h"

namespace fvta {

template <int B, int E, class F>
__device__ __forceinline__ void wb_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    wb_static_for<B + 1, E>(f);
  }
}

template <int ND16, int NCT_>
struct WbwdCfg {
  static constexpr int NW = 4, NCT = NCT_, D = 16 * ND16, UB = 32 * NCT, CB = D / UB;
  static constexpr int NKS = D / 16;            // k-steps of a wave: its quarter of the dz row
#ifndef FVTA_WBWD_PF
#define FVTA_WBWD_PF 8
#endif
  static constexpr int PF = NKS < FVTA_WBWD_PF ? NKS : FVTA_WBWD_PF;  // A fragments in flight
  static constexpr int RS = UB + 4;             // floats per row of a wave's partial tile (16-byte aligned, spreads banks)
  static constexpr int LDS_BYTES = 2 * NW * 32 * RS * 4;
  static constexpr int W_FRAGS = NKS * NCT;
  static constexpr int W_AGPR_FRAGS = W_FRAGS < 64 ? W_FRAGS : 64;  // (a slice beyond the AGPR file would sit in VGPRs)
  static_assert(W_FRAGS * 4 <= 256, "weight slice: 256 registers per lane");
  static_assert(D % UB == 0, "column blocks");
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// two floats -> packed bf16 pair (round to nearest even, v_cvt_pk_bf16_f32): a in the low half
__device__ __forceinline__ unsigned wb_pk_bf16(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

template <class C>
__global__ __launch_bounds__(64 * C::NW, 1) void lstm_bwd_wreg_bf16(FusedBwdArgs a, int RG) {
  constexpr int NCT = C::NCT, RS = C::RS;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][NW][32][RS] partial dh tiles
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
  // blockIdx -> (direction, column block, row group); row group fastest: the column blocks of one row group read the
  // same dz rows and share an XCD (blockIdx & 7 = rg & 7, RG a multiple of 4 and 2 RG CB of 8)
  const int lin = (int)blockIdx.x;
  const int rg = lin % RG, cb = (lin / RG) % C::CB, dir = lin / (RG * C::CB);
  const int t = a.t, d = C::D, K = 4 * d;
  const int nact = a.plan.nactive[t];
  const int ntiles = (nact + 31) / 32;
  if (rg >= ntiles) return;
  const int nnext = (t + 1 < a.J) ? a.plan.nactive[t + 1] : 0;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;

  // ---- this wave's weight fragments: units cb UB + 32 ct + l31, k = w d + 16 ks + 8 hf + (0..7) of the Wh rows of wb
  bf16x8_t w[C::NKS][NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const bf16_t* src = a.Wb[dir] + (size_t)(a.in_i + cb * C::UB + 32 * ct + l31) * K + wave * d + 8 * hf;
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) w[ks][ct] = *reinterpret_cast<const bf16x8_t*>(src + 16 * ks);
  }
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + (trow + a.B) * (size_t)K, (unsigned)nnext * K * 2);  // rows >= nnext read 0
  const unsigned voff_lane = (unsigned)l31 * (K * 2) + (unsigned)(wave * d + 8 * hf) * 2;

  // ---- gate gradient: thread (row er, unit group c8) takes units cb UB + 32 g + 4 c8 .. + 3, g < NCT
  const int er = tid >> 3, c8 = tid & 7;
  const float* __restrict__ cs_p = a.cs + (trow - a.B) * d;  // step t - 1 (unused at t == 0)
  float* __restrict__ dcs = a.dc + (size_t)dir * a.B * d;
  struct In {
    f32x4 g0, g1, cp, dout, dcv;  // g0, g1: the packed bf16 gates (i, j, f, o) of four units
  };
  auto ldnt = [](const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); };  // read once

  // Everything a tile needs from memory is requested one tile ahead: the gate-gradient inputs of tile i + 1 at the top of
  // tile i, and the A fragments run PF k-steps ahead of their MFMAs ACROSS tile boundaries (the last PF k-steps of a tile
  // load the first PF of the next; rows past nnext read zeros through the descriptor), so a wave never drains its loads.
  auto load_in = [&](int rt, In (&x)[NCT]) {
    const int ic = min(32 * rt + er, nact - 1);  // clamped: always a valid row (also for a tile past the end)
    const int64_t oo = a.plan.oo[trow + ic];
#pragma unroll
    for (int g = 0; g < NCT; ++g) {
      const int u = cb * C::UB + 32 * g + 4 * c8;
      const float* gp = reinterpret_cast<const float*>(a.gatesb + (trow + ic) * (size_t)K + 4 * u);
      x[g].g0 = ldnt(gp);
      x[g].g1 = ldnt(gp + 4);
      x[g].cp = t > 0 ? ldnt(cs_p + (size_t)ic * d + u) : f32x4{0.f, 0.f, 0.f, 0.f};
      const float* dp = a.d_out + oo + u;
      if ((reinterpret_cast<uintptr_t>(dp) & 15) == 0)
        x[g].dout = ldnt(dp);
      else
        x[g].dout = f32x4{dp[0], dp[1], dp[2], dp[3]};  // an output row that is not 16-byte aligned
      // dc of a row that was not active at step t + 1 is zero BY DEFINITION (nothing has written it in this call: the
      // engine does not zero the buffer); an unconditional load + select, no branch around the load
      const f32x4 dcl = *reinterpret_cast<const f32x4*>(dcs + (size_t)ic * d + u);
      x[g].dcv = (32 * rt + er) < nnext ? dcl : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lda = [&](unsigned voff, int ks) {
    return __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rz, voff + 32u * ks, 0, 0));
  };
  In in[NCT], inn[NCT];
  bf16x8_t fr[C::PF];
  load_in(rg, in);
#pragma unroll
  for (int ks = 0; ks < C::PF; ++ks) fr[ks] = lda(voff_lane + (unsigned)(32 * rg) * (K * 2), ks);
  int buf = 0;
  for (int rt = rg; rt < ntiles; rt += RG) {
    const int m0 = 32 * rt;
    const int i = m0 + er;
    load_in(rt + RG, inn);
    const bool gemm = m0 < nnext;  // (rows are sorted by length: the tiles with a successor step come first)
    if (gemm) {
      // ---- this wave's partial dh tile: 32 rows x UB units over its quarter of k
      f32x16 acc[NCT];
      const unsigned voff = voff_lane + (unsigned)m0 * (K * 2), voffn = voff + (unsigned)(32 * RG) * (K * 2);
      // (explicit captures: an asm operand inside a generic lambda does not trigger an implicit one)
      auto mfma = [&acc, &w](auto ks_c, auto ct_c, const bf16x8_t cur) {
        constexpr int ks = decltype(ks_c)::value, ct = decltype(ct_c)::value;
        constexpr bool in_agpr = ks * NCT + ct < C::W_AGPR_FRAGS;  // the matrix pipe reads either file directly
        if constexpr (ks == 0) {
          if constexpr (in_agpr)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc[ct]) : "v"(cur), "a"(w[ks][ct]));
          else
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc[ct]) : "v"(cur), "v"(w[ks][ct]));
        } else {
          if constexpr (in_agpr)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[ct]) : "v"(cur), "a"(w[ks][ct]));
          else
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[ct]) : "v"(cur), "v"(w[ks][ct]));
        }
      };
      wb_static_for<0, C::NKS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        const bf16x8_t cur = fr[ks % C::PF];
        if constexpr (ks + C::PF < C::NKS)
          fr[ks % C::PF] = lda(voff, ks + C::PF);
        else
          fr[ks % C::PF] = lda(voffn, ks + C::PF - C::NKS);
        wb_static_for<0, NCT>([&](auto ct_c) { mfma(ks_c, ct_c, cur); });
      });
      // (the asm statements are opaque to the hazard recogniser: cover the matrix pipe's write -> LDS store distance)
      if constexpr (NCT == 2)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));
      else
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]));
      float* my = red + (size_t)((buf * C::NW + wave) * 32) * RS;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) my[((r & 3) + 8 * (r >> 2) + 4 * hf) * RS + 32 * ct + l31] = acc[ct][r];
      __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < NCT; ++g) {
      const int u = cb * C::UB + 32 * g + 4 * c8;
      f32x4 dh4 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (gemm) {
#pragma unroll
        for (int ww = 0; ww < C::NW; ++ww)
          dh4 += *reinterpret_cast<const f32x4*>(&red[(size_t)((buf * C::NW + ww) * 32 + er) * RS + 32 * g + 4 * c8]);
      }
      // (the gate gradient of lstm_bwd_fused_bf16, same forms)
      const u32x4_t ga = __builtin_bit_cast(u32x4_t, in[g].g0), gb = __builtin_bit_cast(u32x4_t, in[g].g1);
      u32x4_t za, zb;  // dz of the four units, packed bf16 (i, j | f, o)
      f32x4 dco;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned w0 = e < 2 ? ga[2 * e] : gb[2 * (e - 2)], w1 = e < 2 ? ga[2 * e + 1] : gb[2 * (e - 2) + 1];
        const float ig = bf2f((bf16_t)(w0 & 0xffff)), jg = bf2f((bf16_t)(w0 >> 16)), fg = bf2f((bf16_t)(w1 & 0xffff)),
                    og = bf2f((bf16_t)(w1 >> 16));
        const float dh = in[g].dout[e] + dh4[e];
        const float tc = fvta_tanh(in[g].cp[e] * fg + ig * jg);
        const float dc = in[g].dcv[e] + dh * og * (1.f - tc * tc);
        const float dzi = dc * jg * ig * (1.f - ig), dzj = dc * ig * (1.f - jg * jg), dzf = dc * in[g].cp[e] * fg * (1.f - fg),
                    dzo = dh * tc * og * (1.f - og);
        const unsigned z0 = (unsigned)f2bf(dzi) | ((unsigned)f2bf(dzj) << 16), z1 = (unsigned)f2bf(dzf) | ((unsigned)f2bf(dzo) << 16);
        if (e < 2) {
          za[2 * e] = z0, za[2 * e + 1] = z1;
        } else {
          zb[2 * (e - 2)] = z0, zb[2 * (e - 2) + 1] = z1;
        }
        dco[e] = dc * fg;
      }
      if (i < nact) {
        float* zp = reinterpret_cast<float*>(a.dzb + (trow + i) * (size_t)K + 4 * u);
        *reinterpret_cast<f32x4*>(zp) = __builtin_bit_cast(f32x4, za);
        *reinterpret_cast<f32x4*>(zp + 4) = __builtin_bit_cast(f32x4, zb);
        *reinterpret_cast<f32x4*>(dcs + (size_t)i * d + u) = dco;
      }
    }
    if (gemm) buf ^= 1;
#pragma unroll
    for (int g = 0; g < NCT; ++g) in[g] = inn[g];
  }
}


// =====================================================================================================================
// The PIPELINED form for d = 512 (the metric shape's text and photo cells), every row count: lstm_bwd_ring_bf16.
//
// Same ownership as above -- a workgroup (one per CU, four waves, one per SIMD) owns 64 hidden units of one direction,
// wave w keeps the B fragments of the quarter [w d, (w + 1) d) of the dz row in the accumulator file for the whole
// launch -- but nothing in it runs in series:
//   * dz_{t+1} rows stream through a WAVE-PRIVATE LDS ring (a wave only ever reads its own quarter of k: no workgroup
//     barrier on the operand path): 8 slots of [32 rows][64 k] = one row tile, filled by LDS-DMA in pieces of 8 rows x
//     128 B (XOR-swizzled source addresses), six slots ahead of the slot being multiplied, behind an EXACTLY counted
//     s_waitcnt vmcnt: every vector-memory instruction of the loop is issued unconditionally (buffer loads / stores,
//     rows past the active prefix fall off the descriptors), each hand-over carries four DMA pieces and two of the
//     epilogue's loads / stores, so "slot s has landed" is "all but the 30 youngest operations are done";
//   * the gate gradient of tile i - 1 is cut into 36 stages of a few vector instructions, placed one behind an MFMA of
//     tile i (the MFMA statements are hand-written and fix the order); its inputs (saved gates, c_{t-1}, d_out, dc) were
//     requested during tile i - 1's own multiplication, a whole tile ahead, into the other of two register sets (the
//     tile loop is unrolled by two: no copies, no waits);
//   * the four partial tiles meet in LDS once per tile (two barriers back to back at the tile boundary).
// dz_{t+1} crosses L2 -> CU once per column block (8x) -- as 256-byte-per-row DMA pieces that hit the XCD's L2 (the eight
// column blocks of a row group share an XCD) instead of the 32-byte segments of the register loads above.
struct RingCfg {
  static constexpr int NW = 4, NCT = 2, D = 512, K = 4 * D, UB = 32 * NCT, CB = D / UB;
  static constexpr int KQ = K / NW, NKS = KQ / 16;          // a wave's share of the dz row: 512 k = 32 k-steps per tile
  static constexpr int SLOT_KS = 4, SLOTS = NKS / SLOT_KS;  // ring slots per tile = slots of the ring (8)
  static constexpr int SLOT_ELEMS = 32 * 16 * SLOT_KS;      // bf16 per slot: 32 rows x 64 k (4 KB)
  static constexpr int RING_ELEMS = SLOTS * SLOT_ELEMS;     // per wave (32 KB)
#ifndef FVTA_RING_LOOK
#define FVTA_RING_LOOK (SLOTS - 2)
#endif
  static constexpr int LOOK = FVTA_RING_LOOK;               // slots the DMA stream runs ahead of the hand-over (<= SLOTS - 2)
  static constexpr int PF = 3, NB = PF + 1;                 // A fragments are read PF k-steps ahead of their MFMAs
  static constexpr int EPI = 2;                             // epilogue loads / stores per hand-over
  static constexpr int VMW = (LOOK - 1) * (SLOT_KS + EPI);  // vector-memory operations younger than the awaited slot (30)
  static constexpr int RS = UB;                             // floats per row of a partial tile
  static constexpr int RED_OFF = NW * RING_ELEMS * 2;       // byte offset of the partial tiles [NW][32][RS]
  static constexpr int LDS_BYTES = RED_OFF + NW * 32 * RS * 4;
  static constexpr int NSTAGES = 4 + 8 * 4;                 // partial reads (2), sums (2), 8 cells x 4
#ifndef FVTA_RING_SPREAD
#define FVTA_RING_SPREAD 4
#endif
  static constexpr int stage_place(int st) { return st * FVTA_RING_SPREAD / 3; }  // MFMA place (2 q + ct) a stage follows
  static constexpr int stage_begin(int p) {  // first stage whose place is >= p
    int st = 0;
    while (st < NSTAGES && stage_place(st) < p) ++st;
    return st;
  }
};
static_assert(RingCfg::SLOTS == 8 && RingCfg::NKS % RingCfg::NB == 0, "geometry");
static_assert(RingCfg::LDS_BYTES <= 163840, "LDS");
static_assert(RingCfg::VMW <= 63, "vmcnt is a 6-bit counter");
// group 0's results are stored at the hand-over of slot 6 (k-step 21), group 1's from slot 7 (k-step 25) on
static_assert(RingCfg::stage_place(4 + 4 * 4 - 1) < 2 * (4 * 6 - RingCfg::PF) &&
              RingCfg::stage_place(RingCfg::NSTAGES - 1) < 2 * (4 * 7 - RingCfg::PF), "stage placement");

struct RingIn {  // gate-gradient inputs of one thread: units 32 g + 4 c8 .. + 3 of its row, g = 0, 1
  u32x4_t g0[2], g1[2];  // packed bf16 gates (i, j | f, o) of two units each
  f32x4 cp[2], dout[2], dcv[2];
};

template <class C>
__global__ __launch_bounds__(64 * C::NW, 1) void lstm_bwd_ring_bf16(FusedBwdArgs a, int RG) {
  constexpr int NCT = C::NCT, K = C::K, D = C::D, RS = C::RS;
  // compile-time ablations (timing experiments, -DFVTA_RING_ABL=bits; results are garbage): 1 no gate stages, 2 no MFMAs,
  // 4 no operand DMA, 8 no weight load, 16 no stores, 32 no epilogue loads, 64 no partial-tile exchange (LDS + barriers)
#ifdef FVTA_RING_ABL
  constexpr int abl = FVTA_RING_ABL;
#else
  constexpr int abl = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + C::RED_OFF);
  const int tid = (int)threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA destinations are wave-uniform
  // workgroups are dealt round-robin over the 8 XCDs: the CB column blocks that stream the same rows share one
  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3;
  const int pair = xcd + 8 * (wslot / C::CB), cb = wslot % C::CB;
  const int dir = pair & 1, rg = pair >> 1;
  const int t = a.t;
  const int nact = a.plan.nactive[t];
  const int ntiles = (nact + 31) >> 5;
  if (rg >= ntiles) return;
  const int nmine = (ntiles - rg + RG - 1) / RG;  // row tiles rg, rg + RG, ...
  const int nnext = (t + 1 < a.J) ? a.plan.nactive[t + 1] : 0;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;

  // The workgroups of a launch walk the k extent of their rows at the same pace; every row of dz is 4 KB, so at any moment
  // they would all be reading the SAME 128-byte column of every row -- the same few L2 channels.  Each column block
  // therefore walks its wave's eight 128-byte k-chunks in its own rotation (the sum order of a dot product changes, nothing
  // else): ring slot s holds k-chunk (s + rot) & 7, and the weight fragments are loaded in that order.
#ifndef FVTA_RING_ROT
#define FVTA_RING_ROT 1
#endif
  const int rot = FVTA_RING_ROT ? ((cb + (FVTA_RING_ROT > 1 ? rg : 0)) & 7) : 0;
  auto kslot = [&](int s) { return (s + rot) & 7; };
  // ---- this wave's weight fragments: units cb UB + 32 ct + l31, k = wave d + 16 ks + 8 hf + (0..7) of the Wh rows of wb
  bf16x8_t w[C::NKS][NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const bf16_t* src = a.Wb[dir] + (size_t)(a.in_i + cb * C::UB + 32 * ct + l31) * K + wave * C::KQ + 8 * hf;
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks)  // (k-step ks of a tile pass multiplies k-chunk kslot(ks / SLOT_KS): the rotation below)
      w[ks][ct] = *reinterpret_cast<const bf16x8_t*>(src + ((abl & 8) ? 0 : 64 * kslot(ks / C::SLOT_KS) + 16 * (ks % C::SLOT_KS)));
  }
  asm volatile("" ::: "memory");

  // ---- descriptors: rows past the active prefix fall off (loads return 0, stores are dropped)
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + (trow + a.B) * (size_t)K, (unsigned)nnext * K * 2);  // dz_{t+1}
  const __amdgpu_buffer_rsrc_t rgt = make_rsrc(a.gatesb + trow * (size_t)K, (unsigned)nact * K * 2);
  const __amdgpu_buffer_rsrc_t rcs = make_rsrc(t > 0 ? a.cs + (trow - a.B) * (size_t)D : a.cs, t > 0 ? (unsigned)nact * D * 4 : 0u);
  const __amdgpu_buffer_rsrc_t rdc = make_rsrc(a.dc + (size_t)dir * a.B * D, (unsigned)nact * D * 4);
  // (loads: only rows that were active at step t + 1 hold a dc of this call -- the others read 0 off this descriptor's end;
  //  the engine does not zero the buffer)
  const __amdgpu_buffer_rsrc_t rdc_ld = make_rsrc(a.dc + (size_t)dir * a.B * D, (unsigned)nnext * D * 4);
  const __amdgpu_buffer_rsrc_t rzo = make_rsrc(a.dzb + trow * (size_t)K, (unsigned)nact * K * 2);  // dz_t

  // ---- the operand stream: piece j of a slot = rows 8 j + (lane >> 3), 16-byte chunk (lane & 7) of the LDS row; the
  // chunk's SOURCE is logical chunk (lane & 7) ^ ((row >> 1) & 7) (conflict-free ds_read_b128 of the fragments)
  unsigned voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 8 * j + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    voff[j] = (unsigned)r * (K * 2) + (unsigned)(wave * C::KQ + 8 * c) * 2;
  }
  bf16_t* ring = smem + wave * C::RING_ELEMS;
  auto issue_dma = [&](auto s_c, int ord) {  // slot s of the workgroup's tile number ord (past the last: zeros)
    constexpr int s = decltype(s_c)::value;
    if constexpr ((abl & 4) != 0) return;
    const unsigned base = (unsigned)(32 * (rg + RG * ord)) * (unsigned)(K * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(rz, ring + s * C::SLOT_ELEMS + j * 512, voff[j] + base, 128 * kslot(s));  // (an immediate offset would move the LDS address too)
  };

  // ---- gate gradient: thread (row er, unit group c8) takes units cb UB + 32 g + 4 c8 .. + 3, g < 2
  const int er = tid >> 3, c8 = tid & 7;
  const int u0 = cb * C::UB + 4 * c8;
  const unsigned vo_g = (unsigned)er * (K * 2) + (unsigned)u0 * 8;  // gates / dz: 8 bytes per unit
  const unsigned vo_f = (unsigned)er * (D * 4) + (unsigned)u0 * 4;  // c, dc
  auto ld4 = [](const __amdgpu_buffer_rsrc_t r, unsigned vo, int imm) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, vo, imm, 0);
  };
  int64_t oo_sel = 0;  // output offset of this lane's row of the tile being multiplied
  // ONE 8-byte vector load per tile and lane, issued a whole tile ahead of its use (so that waiting for it drains nothing
  // that has not landed anyway).  (As wave-uniform scalar loads, selected by row, at the head of the tile they belong to,
  // every wave sat out eight scalar-cache round trips per tile, one s_waitcnt lgkmcnt(0) after the other.)
  const int64_t* __restrict__ oo_g = a.plan.oo + trow;
  int64_t oo_nxt = 0;
  auto row_offsets_issue = [&](int m0n) { oo_nxt = oo_g[min(m0n + 8 * wave + (lane >> 3), nact - 1)]; };  // clamped: always an active row
  // epilogue operation n (0 .. 15) of a tile pass: 10 loads of the tile being multiplied (into `in`), 6 stores of the tile
  // whose gate gradient has just run (from zw / dco)
  unsigned zw[2][8];  // packed dz words of group g: [2 e] = (i, j), [2 e + 1] = (f, o) of unit e
  f32x4 dco[2];
  auto epi_load = [&](auto n_c, RingIn& in, int m0) {
    constexpr int n = decltype(n_c)::value, g = n / 5, what = n % 5;
    if constexpr ((abl & 32) != 0) return;
    const unsigned rg_ = (unsigned)m0 * (K * 2), rf_ = (unsigned)m0 * (D * 4);
    if constexpr (what == 0) in.g0[g] = ld4(rgt, vo_g + rg_, 256 * g);
    else if constexpr (what == 1) in.g1[g] = ld4(rgt, vo_g + rg_, 256 * g + 16);
    else if constexpr (what == 2) in.cp[g] = __builtin_bit_cast(f32x4, ld4(rcs, vo_f + rf_, 128 * g));
    else if constexpr (what == 3) {
      const float* dp = a.d_out + oo_sel + u0 + 32 * g;
      if ((reinterpret_cast<uintptr_t>(dp) & 15) == 0)
        in.dout[g] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dp));
      else
        in.dout[g] = f32x4{dp[0], dp[1], dp[2], dp[3]};  // an output row that is not 16-byte aligned (more operations: the count stays safe)
    } else in.dcv[g] = __builtin_bit_cast(f32x4, ld4(rdc_ld, vo_f + rf_, 128 * g));
  };
  auto epi_store = [&](auto n_c, int m0p) {
    constexpr int n = decltype(n_c)::value, g = n / 3, what = n % 3;
    if constexpr ((abl & 16) != 0) return;
    const unsigned rg_ = (unsigned)m0p * (K * 2), rf_ = (unsigned)m0p * (D * 4);
    if constexpr (what == 0)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{zw[g][0], zw[g][1], zw[g][2], zw[g][3]}, rzo, vo_g + rg_, 256 * g, 0);
    else if constexpr (what == 1)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{zw[g][4], zw[g][5], zw[g][6], zw[g][7]}, rzo, vo_g + rg_, 256 * g + 16, 0);
    else
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, dco[g]), rdc, vo_f + rf_, 128 * g, 0);
  };
  // the two operations of hand-over s: slots 1 .. 5 carry the loads, 6, 7 and 0 the stores
  auto epi_pair = [&](auto s_c, RingIn& in, int m0, int m0p) {
    constexpr int s = decltype(s_c)::value;
    if constexpr (s >= 1 && s <= 5) {
      epi_load(std::integral_constant<int, 2 * (s - 1)>{}, in, m0);
      epi_load(std::integral_constant<int, 2 * (s - 1) + 1>{}, in, m0);
    } else {
      constexpr int b = s == 6 ? 0 : (s == 7 ? 2 : 4);
      epi_store(std::integral_constant<int, b>{}, m0p);
      epi_store(std::integral_constant<int, b + 1>{}, m0p);
    }
  };

  // ---- the gate gradient of a tile in stages (lstm_bwd_fused_bf16's forms); the empty asm statements pin a stage's
  // results to its place in the MFMA stream
  f32x4 ps[4];        // partial sums being read
  f32x4 dh4[2];
  float ig, jg, fg, og, dh, cc, te, tc, dc;
  auto run_stage = [&](auto st_c, const RingIn& in) {
    constexpr int st = decltype(st_c)::value;
    if constexpr (st < 4) {
      constexpr int g = st & 1;
      if constexpr (st < 2) {
        if constexpr (st == 1) dh4[0] = (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) ps[ww] = *reinterpret_cast<const f32x4*>(&red[(size_t)(ww * 32 + er) * RS + 32 * g + 4 * c8]);
      } else if constexpr (st == 2) {
        dh4[1] = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        asm volatile("" : "+v"(dh4[0]), "+v"(dh4[1]));
      }
    } else {
      constexpr int c = (st - 4) / 4, k = (st - 4) % 4, g = c / 4, e = c % 4;
      if constexpr (k == 0) {
        const unsigned w0 = e < 2 ? in.g0[g][2 * e] : in.g1[g][2 * (e - 2)], w1 = e < 2 ? in.g0[g][2 * e + 1] : in.g1[g][2 * (e - 2) + 1];
        ig = __uint_as_float(w0 << 16), jg = __uint_as_float(w0 & 0xffff0000u);
        fg = __uint_as_float(w1 << 16), og = __uint_as_float(w1 & 0xffff0000u);
        dh = in.dout[g][e] + dh4[g][e];
        cc = in.cp[g][e] * fg + ig * jg;
        te = __expf(-2.0f * fabsf(cc));
        asm volatile("" : "+v"(ig), "+v"(jg), "+v"(fg), "+v"(og), "+v"(dh), "+v"(cc), "+v"(te));
      } else if constexpr (k == 1) {
        tc = copysignf((1.0f - te) * __builtin_amdgcn_rcpf(1.0f + te), cc);
        dc = in.dcv[g][e] + dh * og * (1.f - tc * tc);
        asm volatile("" : "+v"(tc), "+v"(dc));
      } else if constexpr (k == 2) {
        const float dzi = dc * jg * ig * (1.f - ig), dzj = dc * ig * (1.f - jg * jg);
        unsigned z0 = wb_pk_bf16(dzi, dzj);
        asm volatile("" : "+v"(z0));
        zw[g][2 * e] = z0;
      } else {
        const float dzf = dc * in.cp[g][e] * fg * (1.f - fg), dzo = dh * tc * og * (1.f - og);
        unsigned z1 = wb_pk_bf16(dzf, dzo);
        float o = dc * fg;
        asm volatile("" : "+v"(z1), "+v"(o));
        zw[g][2 * e + 1] = z1;
        dco[g][e] = o;
      }
    }
  };

  // ---- MFMA by hand: the weight fragments stay in the accumulator file, the matrix pipe reads them there
  f32x16 acc[NCT];
  auto mfma = [&acc, &w](auto ks_c  /* explicit: an asm operand in a generic lambda does not capture implicitly */, auto ct_c, const bf16x8_t cur) {
    constexpr int ks = decltype(ks_c)::value, ct = decltype(ct_c)::value;
    if constexpr ((abl & 2) != 0) {
      asm volatile("" : "+v"(acc[ct]) : "v"(cur), "a"(w[ks][ct]));
    } else if constexpr (ks == 0)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc[ct]) : "v"(cur), "a"(w[ks][ct]));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[ct]) : "v"(cur), "a"(w[ks][ct]));
  };
  const bf16_t* ap[C::SLOT_KS];  // this lane's fragment address of position ks of a slot: row l31, chunk (2 ks + hf) ^ ((l31 >> 1) & 7)
#pragma unroll
  for (int ks = 0; ks < C::SLOT_KS; ++ks) ap[ks] = ring + l31 * 64 + (((2 * ks + hf) ^ ((l31 >> 1) & 7)) << 3);
  Pack8 fr[C::NB];

  // hand-over of slot s of tile `ord`: afterwards the slot may be read.  Issues the DMA of the slot LOOK ahead and the
  // two epilogue operations of this hand-over
  auto handover = [&](auto s_c, int ord, RingIn& in, int m0, int m0p) {
    constexpr int s = decltype(s_c)::value;
    wait_vmcnt<C::VMW>();
    issue_dma(std::integral_constant<int, (s + C::LOOK) % C::SLOTS>{}, ord + (s + C::LOOK) / C::SLOTS);
    epi_pair(s_c, in, m0, m0p);
    asm volatile("" ::: "memory");
  };

  // ---- one tile pass: the MFMAs of tile `it` (its inputs -> inC), the gate gradient of the tile before (inP, rows m0p)
  auto body = [&](int it, const RingIn& inP, RingIn& inC, int m0p) {
    const int m0 = 32 * (rg + RG * it);
    oo_sel = oo_nxt;  // (requested during the tile pass before)
    wb_static_for<0, C::NKS>([&](auto q_c) {
      constexpr int q = decltype(q_c)::value, n = q + C::PF;
      if constexpr (n % C::SLOT_KS == 0) {
        constexpr int s = (n / C::SLOT_KS) % C::SLOTS;
        handover(std::integral_constant<int, s>{}, it + n / C::NKS, inC, m0, m0p);
      }
      fr[n % C::NB].f = *reinterpret_cast<const f32x4*>(ap[n % C::SLOT_KS] + ((n / C::SLOT_KS) % C::SLOTS) * C::SLOT_ELEMS);
      if constexpr (q == 0) row_offsets_issue(m0 + 32 * RG);
      wb_static_for<0, NCT>([&](auto ct_c) {
        constexpr int p = 2 * q + decltype(ct_c)::value;
        mfma(q_c, ct_c, fr[q % C::NB].b);
        if constexpr (!(abl & 1)) wb_static_for<C::stage_begin(p), C::stage_begin(p + 1)>([&](auto st_c) { run_stage(st_c, inP); });
      });
    });
    // the tile's partial sums -> LDS.  First barrier: every wave has read the previous tile's partials (its stages 0, 1
    // ran at the head of this pass); second: this tile's are visible
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));
    if constexpr ((abl & 64) != 0) return;
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float* my = red + (size_t)(wave * 32) * RS;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) my[((r & 3) + 8 * (r >> 2) + 4 * hf) * RS + 32 * ct + l31] = acc[ct][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // the last tile's gate gradient, nothing to hide behind
  auto finish = [&](const RingIn& inP, int m0p) {
    wb_static_for<0, C::NSTAGES>([&](auto st_c) { run_stage(st_c, inP); });
    wb_static_for<0, 6>([&](auto n_c) { epi_store(n_c, m0p); });
    wait_vmcnt<0>();  // the ring's trailing DMA pieces must not outlive the workgroup's LDS allocation
  };

  // ---- prologue: what the hand-overs of slots 2 .. 7 of a tile pass in front of the first would have issued -- the DMA
  // of slots 0 .. 5 of tile 0, each with two epilogue operations (loads of rows past the end: zeros; stores: dropped)
  RingIn inA, inB;
  const int m0_none = 32 * ntiles;  // a row tile past the active prefix
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) zw[g][2 * e] = zw[g][2 * e + 1] = 0u, dco[g][e] = 0.f;
  row_offsets_issue(m0_none);
  oo_sel = oo_nxt;
  row_offsets_issue(32 * rg);  // the first tile's
  epi_pair(std::integral_constant<int, 1>{}, inB, m0_none, m0_none);
  asm volatile("" ::: "memory");
  wb_static_for<2, 8>([&](auto s_c) {
    constexpr int s = decltype(s_c)::value;
    issue_dma(std::integral_constant<int, s - 2>{}, 0);
    epi_pair(s_c, inB, m0_none, m0_none);
    asm volatile("" ::: "memory");
  });
  handover(std::integral_constant<int, 0>{}, 0, inB, m0_none, m0_none);
#pragma unroll
  for (int q = 0; q < C::PF; ++q) fr[q].f = *reinterpret_cast<const f32x4*>(ap[q]);

  int m0p = m0_none;
  for (int it = 0;; it += 2) {
    body(it, inB, inA, m0p);
    m0p = 32 * (rg + RG * it);
    if (it + 1 >= nmine) {
      finish(inA, m0p);
      break;
    }
    body(it + 1, inA, inB, m0p);
    m0p = 32 * (rg + RG * (it + 1));
    if (it + 2 >= nmine) {
      finish(inB, m0p);
      break;
    }
  }
}

// ------------------------------------------------------------------------------------------------------- host ----
static int wbwd_cus() {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  return cus;
}

template <class C>
static void launch_wbwd(const FusedBwdArgs& a, int rows, hipStream_t s) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_wreg_bf16<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES);
  int rg = wbwd_cus() / (2 * C::CB);
  rg = rg / 4 * 4;
  if (rg < 4) rg = 4;
  const int tiles = (rows + 31) / 32;
  while (rg > 4 && rg - 4 >= tiles) rg -= 4;
  hipLaunchKernelGGL(lstm_bwd_wreg_bf16<C>, dim3(2 * rg * C::CB), dim3(64 * C::NW), C::LDS_BYTES, s, a, rg);
}

// Steps with at most this many rows (the host's count of active sequences when it has one, else the call's B) run here:
// dz_{t+1} crosses L2 -> CU once per column block, which the tiled kernel's 256-row tiles do d/256 times only.
#ifndef FVTA_WBWD_MAX_ROWS
#define FVTA_WBWD_MAX_ROWS 6144
#endif
constexpr int WBWD_MAX_ROWS = FVTA_WBWD_MAX_ROWS;

// the pipelined kernel: d = 512, every row count (32-bit buffer offsets: the call's dz slab must stay under 2 GB)
static bool launch_bwd_ring(const FusedBwdArgs& a, int rows, hipStream_t s) {
  typedef RingCfg C;
  if (a.d != C::D || (size_t)(a.B + 4096) * C::K * 2 >= (1ull << 31)) return false;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_ring_bf16<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES);
  int rg = wbwd_cus() / (2 * C::CB);
  rg = rg / 4 * 4;
  if (rg < 4) rg = 4;
  if (rg > 32) rg = 32;  // (row tiles past the end are addressed up to 2 rg tiles beyond B: the 4096-row margin above)
  const int tiles = (rows + 31) / 32;
  while (rg > 4 && rg - 4 >= tiles) rg -= 4;
  hipLaunchKernelGGL(lstm_bwd_ring_bf16<C>, dim3(2 * rg * C::CB), dim3(64 * C::NW), C::LDS_BYTES, s, a, rg);
  return true;
}

#ifndef FVTA_RING_MAX_ROWS
#define FVTA_RING_MAX_ROWS 8192
#endif
constexpr int RING_MAX_ROWS = FVTA_RING_MAX_ROWS;

// launches of the text / photo cell's backward step by kernel since the last read: [tiled, pipelined, round-3 weights-stationary]
// (fvta_lstm_bwd_kernel_counts: bench.py labels its roofline by what actually ran)
long long g_bwd_step_counts[3] = {0, 0, 0};

bool launch_bwd_wreg(const FusedBwdArgs& a, hipStream_t s) {
  if (a.xm != 1 || !a.gatesb || a.in_i % 16) return false;
  const int rows = a.nact_hint >= 0 ? a.nact_hint : a.B;
  // d = 512: the pipelined kernel up to RING_MAX_ROWS active rows (the host's count when it has one), the tiled kernel
  // above.  At the dense metric shape (12,864 rows) the two take the same time alone (134-135 us per launch); the tiled one
  // leaves 52 CUs to the photo cell's side stream, the pipelined one (160 KB of LDS on every CU) does not: step 13.07 vs
  // 13.41 ms.  Ragged batches (SURVEY 8d lengths): 4.92 ms per step with the limit at 8192, 5.03 at 6144, 5.48 tiled only.
  if ((wreg_mode() & 4) && rows <= RING_MAX_ROWS && launch_bwd_ring(a, rows > 0 ? rows : 1, s)) {
    if (a.B > 64) ++g_bwd_step_counts[1];
    return true;
  }
  if (!(wreg_mode() & 2)) return false;
  if (rows > WBWD_MAX_ROWS) return false;
  if (a.d == 512) launch_wbwd<WbwdCfg<32, 2>>(a, rows > 0 ? rows : 1, s);
  else if (a.d == 1024) launch_wbwd<WbwdCfg<64, 1>>(a, rows > 0 ? rows : 1, s);
  else if (a.d == 128) launch_wbwd<WbwdCfg<8, 2>>(a, rows > 0 ? rows : 1, s);
  else return false;
  if (a.B > 64) ++g_bwd_step_counts[2];
  return true;
}

}  // namespace fvta
