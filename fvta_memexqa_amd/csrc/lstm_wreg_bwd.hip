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
      x[g].dcv = *reinterpret_cast<const f32x4*>(dcs + (size_t)ic * d + u);
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

bool launch_bwd_wreg(const FusedBwdArgs& a, hipStream_t s) {
  if (a.xm != 1 || !a.gatesb || !(wreg_mode() & 2) || a.in_i % 16) return false;
  const int rows = a.nact_hint >= 0 ? a.nact_hint : a.B;
  if (rows > WBWD_MAX_ROWS) return false;
  if (a.d == 512) launch_wbwd<WbwdCfg<32, 2>>(a, rows > 0 ? rows : 1, s);
  else if (a.d == 1024) launch_wbwd<WbwdCfg<64, 1>>(a, rows > 0 ? rows : 1, s);
  else if (a.d == 128) launch_wbwd<WbwdCfg<8, 2>>(a, rows > 0 ? rows : 1, s);
  else return false;
  return true;
}

}  // namespace fvta
