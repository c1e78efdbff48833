// Time warp of the context tensor: model_v2.py:953-1009 + time_indication_func 301-341.
//
// As written in the reference, hall_t1 and hall_t2 are the SAME expression (model_v2.py:979-980), so the
// [N,K,T,T,4d] construction collapses to a per-position scalar:
//   c[n,t]      = tanh( sum_k ( WC . (WH[:w]^T h[n,k,t]^2 + b_WH + lq[n]) + b_WC ) )
//               = tanh( sum_k sum_c v[c] h[n,k,t,c]^2 + K (s0 + WC . lq[n]) ),   v = WH[:w] WC,  s0 = b_WH . WC + b_WC
//   h'[n,k,t,:] = h[n,k,t,:] * c[n,t] * cnt(t),   cnt(t) = #{t' allowed by warp_type around t}
// O(N K T w) instead of the literal O(N K T^2 4d) (the "9 GB" of README.MD:229).  window_t gets no
// gradient (tf.ceil -> int cast, model_v2.py:335).
#include "fvta_common.h"

namespace fvta {

__device__ __forceinline__ f32x4 ld4t(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ float tw_count(int t, int T, int warp_type, int win) {
  switch (warp_type) {
    case 1: return (float)T;            // all time (model_v2.py:303-304)
    case 2: return 1.f;                 // diagonal (312-314)
    case 3: return (float)(t + 1);      // past, lower triangular (315-323)
    case 4: return (float)(T - t);      // future (324-330)
    default: return (float)(min(t + win, T - 1) - max(t - win, 0) + 1);  // 5: band of half-width ceil(window_t) (332-339)
  }
}

struct TwWork {
  float* v;      // [w]
  float* s0;     // [1]
  float* sq;     // [N]
  float* dsk;    // [N,K,T]
  float* dz;     // [N,T]
  float* dvp;    // [NWG][w] partials of dv
  float* dv;     // [w]
  float* dsq;    // [N]
  float* ds0;    // [1]
  int nwg;
  size_t bytes;
};
static TwWork tw_work(const fvta_timewarp_desc* d, void* p) {
  FvtaCarver c(p);
  TwWork w;
  w.nwg = 2048;   // wave slots of the fused kernels' dv partials: 256 CUs x 8 waves
  w.v = c.take<float>(d->w);
  w.s0 = c.take<float>(1);
  w.sq = c.take<float>(d->N);
  w.dsk = c.take<float>((size_t)d->N * d->K * d->T);
  w.dz = c.take<float>((size_t)d->N * d->T);
  w.dvp = c.take<float>((size_t)w.nwg * d->w);
  w.dv = c.take<float>(d->w);
  w.dsq = c.take<float>(d->N);
  w.ds0 = c.take<float>(1);
  w.bytes = c.off;
  return w;
}

// v[c] = sum_o WH[c][o] WC[o] ; s0 ; sq[n] = lq[n] . WC.   grid ceil((w + N + 1)/4), 256 threads (one wave per output)
__global__ void tw_vec_kernel(int N, int w, const float* __restrict__ WH, const float* __restrict__ WHb,
                              const float* __restrict__ WC, const float* __restrict__ WCb, const float* __restrict__ lq,
                              TwWork wk) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o > w + N) return;
  const float* src = o < w ? WH + (size_t)o * w : (o == w ? WHb : lq + (size_t)(o - w - 1) * w);
  float acc = 0.f;
  for (int c = lane; c < w; c += 64) acc += src[c] * WC[c];
  acc = wave_sum(acc);
  if (lane == 0) {
    if (o < w)
      wk.v[o] = acc;
    else if (o == w)
      wk.s0[0] = acc + WCb[0];
    else
      wk.sq[o - w - 1] = acc;
  }
}

// c[n,t] and the row scale.  One wave per (n,t).  grid ceil(N*T/4)
__global__ void tw_coef_kernel(fvta_timewarp_desc d, int win, const float* __restrict__ hall, TwWork wk,
                               float* __restrict__ c_out, float* __restrict__ scale) {
  const int pos = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pos >= d.N * d.T) return;
  const int n = pos / d.T, t = pos % d.T;
  float acc = 0.f;
  for (int k = 0; k < d.K; ++k) {
    const float* row = hall + (((size_t)n * d.K + k) * d.T + t) * d.w;
    for (int c = lane * 4; c < d.w; c += 256) {
      const f32x4 h = ld4t(row + c), v = ld4t(wk.v + c);
      const f32x4 p = h * h * v;
      acc += (p[0] + p[1]) + (p[2] + p[3]);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    const float c = tanhf(acc + (float)d.K * (wk.s0[0] + wk.sq[n]));
    c_out[pos] = c;
    scale[pos] = c * tw_count(t, d.T, d.warp_type, win);
  }
}

// out[n,k,t,:] = a[n,k,t,:] * scale[n,t]
__global__ void tw_apply_kernel(fvta_timewarp_desc d, const float* __restrict__ a, const float* __restrict__ scale,
                                float* __restrict__ out) {
  const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t w4 = d.w / 4, total = (size_t)d.N * d.K * d.T * w4;
  if (i4 >= total) return;
  const size_t row = i4 / w4;
  const int t = (int)(row % d.T), n = (int)(row / ((size_t)d.K * d.T));
  *reinterpret_cast<f32x4*>(out + i4 * 4) = ld4t(a + i4 * 4) * scale[(size_t)n * d.T + t];
}

// dsk[n,k,t] = d_warp[n,k,t,:] . h[n,k,t,:]   one wave per row
__global__ void tw_rowdot_kernel(fvta_timewarp_desc d, const float* __restrict__ hall, const float* __restrict__ d_warp,
                                 float* __restrict__ dsk) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (size_t)d.N * d.K * d.T) return;
  float acc = 0.f;
  for (int c = lane * 4; c < d.w; c += 256) {
    const f32x4 p = ld4t(hall + row * d.w + c) * ld4t(d_warp + row * d.w + c);
    acc += (p[0] + p[1]) + (p[2] + p[3]);
  }
  acc = wave_sum(acc);
  if (lane == 0) dsk[row] = acc;
}

// dz[n,t] = (sum_k dsk) * cnt(t) * (1 - c^2)
__global__ void tw_dz_kernel(fvta_timewarp_desc d, int win, const float* __restrict__ c_saved, TwWork wk,
                             const float* __restrict__ d_scale_att) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= d.N * d.T) return;
  const int n = pos / d.T, t = pos % d.T;
  float ds = d_scale_att ? d_scale_att[pos] : 0.f;  // time_warp_att: the attention's own gradient w.r.t. c[n,t] cnt(t)
  for (int k = 0; k < d.K; ++k) ds += wk.dsk[((size_t)n * d.K + k) * d.T + t];
  const float c = c_saved[pos];
  wk.dz[pos] = ds * tw_count(t, d.T, d.warp_type, win) * (1.f - c * c);
}

// d_hall = scale * d_warp + dz * 2 v h ;  dv partials.  grid nwg, 256 threads; rows strided over workgroups
__global__ __launch_bounds__(256) void tw_apply_bwd_kernel(fvta_timewarp_desc d, int win, const float* __restrict__ hall,
                                                           const float* __restrict__ d_warp,
                                                           const float* __restrict__ c_saved, TwWork wk,
                                                           float* __restrict__ d_hall) {
  const int tid = threadIdx.x;
  const size_t rows = (size_t)d.N * d.K * d.T;
  for (int c0 = tid * 4; c0 < d.w; c0 += 1024) {  // each thread owns channel quads c0, c0+1024, ...
    const f32x4 v = ld4t(wk.v + c0);
    f32x4 dvacc = {0.f, 0.f, 0.f, 0.f};
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
      const int t = (int)(row % d.T), n = (int)(row / ((size_t)d.K * d.T));
      const size_t pos = (size_t)n * d.T + t;
      const float dz = wk.dz[pos];
      const float sc = c_saved[pos] * tw_count(t, d.T, d.warp_type, win);
      const f32x4 h = ld4t(hall + row * d.w + c0);
      const f32x4 g = ld4t(d_warp + row * d.w + c0);
      *reinterpret_cast<f32x4*>(d_hall + row * d.w + c0) = g * sc + h * v * (2.f * dz);
      dvacc += h * h * dz;
    }
    *reinterpret_cast<f32x4*>(wk.dvp + (size_t)blockIdx.x * d.w + c0) = dvacc;
  }
}

// ---- fused forms (w a multiple of 256, K <= TW_KMAX): ONE pass over the context tensor per direction of the step.
// A WAVE owns a position (n, t): its K rows (K x w floats: 24 KB at the metric shape) sit in registers between the
// reduction over (k, c) that gives c[n,t] / dz[n,t] and the scaling that uses it, so hall (and d_warp) cross HBM once
// instead of twice -- the separate kernels above are all on the HBM roof already (4.6-5.4 TB/s), only bytes are left to save.
// Lane l holds channels 256 g + 4 l .. + 3 of a row (G4 = w / 256 quads): every access 16 bytes, 1 KB per wave-instruction.
constexpr int TW_KMAX = 8;

template <int G4>
__global__ __launch_bounds__(256) void tw_fwd_fused_kernel(fvta_timewarp_desc d, int win, const float* __restrict__ hall,
                                                           TwWork wk, float* __restrict__ c_out, float* __restrict__ scale,
                                                           float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6), nwv = gridDim.x * 4;
  f32x4 v[G4];
#pragma unroll
  for (int g = 0; g < G4; ++g) v[g] = ld4t(wk.v + 256 * g + 4 * lane);
  const float s0 = wk.s0[0];
  for (int pos = wv; pos < d.N * d.T; pos += nwv) {
    const int n = pos / d.T, t = pos % d.T;
    f32x4 h[TW_KMAX][G4];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k)
      if (k < d.K) {
        const float* row = hall + (((size_t)n * d.K + k) * d.T + t) * d.w + 4 * lane;
#pragma unroll
        for (int g = 0; g < G4; ++g) h[k][g] = ld4t(row + 256 * g);
      }
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k)
      if (k < d.K) {
#pragma unroll
        for (int g = 0; g < G4; ++g) {
          const f32x4 p = h[k][g] * h[k][g] * v[g];
          acc += (p[0] + p[1]) + (p[2] + p[3]);
        }
      }
    acc = wave_sum(acc);
    const float c = tanhf(acc + (float)d.K * (s0 + wk.sq[n]));
    const float sc = c * tw_count(t, d.T, d.warp_type, win);
    if (lane == 0) {
      c_out[pos] = c;
      scale[pos] = sc;
    }
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k)
      if (k < d.K) {
        float* row = out + (((size_t)n * d.K + k) * d.T + t) * d.w + 4 * lane;
#pragma unroll
        for (int g = 0; g < G4; ++g) *reinterpret_cast<f32x4*>(row + 256 * g) = h[k][g] * sc;
      }
  }
}

// backward: dz[n,t] = (d_scale_att + sum_k d_warp . h) cnt (1 - c^2);  d_hall = scale d_warp + dz 2 v h;  dv partials per
// wave slot (gridDim.x * 4 == wk.nwg slots).
template <int G4>
__global__ __launch_bounds__(256) void tw_bwd_fused_kernel(fvta_timewarp_desc d, int win, const float* __restrict__ hall,
                                                           const float* __restrict__ d_warp, const float* __restrict__ c_saved,
                                                           const float* __restrict__ d_scale_att, TwWork wk,
                                                           float* __restrict__ d_hall) {
  const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6), nwv = gridDim.x * 4;
  f32x4 v[G4], dvacc[G4];
#pragma unroll
  for (int g = 0; g < G4; ++g) {
    v[g] = ld4t(wk.v + 256 * g + 4 * lane);
    dvacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int pos = wv; pos < d.N * d.T; pos += nwv) {
    const int n = pos / d.T, t = pos % d.T;
    f32x4 h[TW_KMAX][G4], gw[TW_KMAX][G4];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k)
      if (k < d.K) {
        const size_t ro = (((size_t)n * d.K + k) * d.T + t) * d.w + 4 * lane;
#pragma unroll
        for (int g = 0; g < G4; ++g) {
          h[k][g] = ld4t(hall + ro + 256 * g);
          gw[k][g] = ld4t(d_warp + ro + 256 * g);
        }
      }
    // (per-row dots in k order, then the sum over k: the order of the separate kernels)
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k)
      if (k < d.K) {
        float dk = 0.f;
#pragma unroll
        for (int g = 0; g < G4; ++g) {
          const f32x4 p = h[k][g] * gw[k][g];
          dk += (p[0] + p[1]) + (p[2] + p[3]);
        }
        acc += wave_sum(dk);
      }
    const float c = c_saved[pos], cnt = tw_count(t, d.T, d.warp_type, win);
    const float dz = ((d_scale_att ? d_scale_att[pos] : 0.f) + acc) * cnt * (1.f - c * c);
    const float sc = c * cnt;
    if (lane == 0) wk.dz[pos] = dz;
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k)
      if (k < d.K) {
        float* row = d_hall + (((size_t)n * d.K + k) * d.T + t) * d.w + 4 * lane;
#pragma unroll
        for (int g = 0; g < G4; ++g) {
          *reinterpret_cast<f32x4*>(row + 256 * g) = gw[k][g] * sc + h[k][g] * v[g] * (2.f * dz);
          dvacc[g] += h[k][g] * h[k][g] * dz;
        }
      }
  }
#pragma unroll
  for (int g = 0; g < G4; ++g) *reinterpret_cast<f32x4*>(wk.dvp + (size_t)wv * d.w + 256 * g + 4 * lane) = dvacc[g];
}

// ---- the fused forms over the encoders' bf16 SHADOW ROWS (fvta_lstm_shadow_rows; the bf16 engine): the context tensor is
// never stored in fp32, so the warp reads row (n,k,t) as two bf16 half-rows through the address table (table[0][row]: channels
// [0, w/2), table[1][row]: [w/2, w); padding rows point at a zero half-row) and writes the WARPED rows as bf16 into a dense
// buffer -- which the focal attention reads through a second, static table (fvta_attn_fwd_shadow / _bwd_shadow).  Per element:
// 2 B read + 2 B written here and 2 B read by the attention, against 4 + 4 + 4 (and the 4 B the bi-LSTM no longer stores).
// The backward reads the same bf16 rows and the attention's fp32 gradient of the warped rows.  w / 2 a multiple of 256
// (a lane's 256-channel block lies in one half-row).
typedef unsigned short tw_bf16;
typedef tw_bf16 tw_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 tw_ld_bf4(unsigned long long addr, int elem) {
  const tw_bf16x4 r = *(const tw_bf16x4 __attribute__((address_space(1)))*)(addr + (unsigned long long)elem * 2ull);   // (global, not flat)
  return f32x4{__uint_as_float((unsigned)r[0] << 16), __uint_as_float((unsigned)r[1] << 16), __uint_as_float((unsigned)r[2] << 16),
               __uint_as_float((unsigned)r[3] << 16)};
}

// The table entries of a position's 2 K half-rows come by ONE vector load (lane l < 2 K holds entry l: half l / K, stream l % K),
// requested a whole position ahead and broadcast with v_readlane: no dependent scalar-load chain in front of the row loads.
__device__ __forceinline__ unsigned long long tw_tab_load(const unsigned long long* __restrict__ table, size_t nkt, int K, int T, int pos,
                                                          int lane) {
  const int n = pos / T, t = pos % T;
  const int k = lane % K, half = lane / K;
  const size_t idx = (size_t)(half & 1) * nkt + ((size_t)n * K + k) * T + t;   // (lanes >= 2 K re-read an entry of the position)
  return table[idx];
}
__device__ __forceinline__ unsigned long long tw_bcast64(unsigned long long v, int src) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, src), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}

template <int G4>
__global__ __launch_bounds__(256) void tw_fwd_shadow_kernel(fvta_timewarp_desc d, int win, const unsigned long long* __restrict__ table,
                                                            TwWork wk, float* __restrict__ c_out, float* __restrict__ scale,
                                                            tw_bf16* __restrict__ out) {
  const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6), nwv = gridDim.x * 4;
  const size_t nkt = (size_t)d.N * d.K * d.T;
  const int dp = d.w / 2, npos = d.N * d.T;
  f32x4 v[G4];
#pragma unroll
  for (int g = 0; g < G4; ++g) v[g] = ld4t(wk.v + 256 * g + 4 * lane);
  const float s0 = wk.s0[0];
  if (wv >= npos) return;
  unsigned long long tab_cur = tw_tab_load(table, nkt, d.K, d.T, wv, lane);
  for (int pos = wv; pos < npos; pos += nwv) {
    const int n = pos / d.T, t = pos % d.T;
    const unsigned long long tab_nxt = tw_tab_load(table, nkt, d.K, d.T, min(pos + nwv, npos - 1), lane);   // (clamped: always valid)
    f32x4 h[TW_KMAX][G4];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k) {
      const int kc = min(k, d.K - 1);   // (unconditional loads: streams k >= K re-read the last one and are not used)
      const unsigned long long a0 = tw_bcast64(tab_cur, kc), a1 = tw_bcast64(tab_cur, d.K + kc);
#pragma unroll
      for (int g = 0; g < G4; ++g) h[k][g] = 256 * g < dp ? tw_ld_bf4(a0, 256 * g + 4 * lane) : tw_ld_bf4(a1, 256 * g - dp + 4 * lane);
    }
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k)
      if (k < d.K) {
#pragma unroll
        for (int g = 0; g < G4; ++g) {
          const f32x4 p = h[k][g] * h[k][g] * v[g];
          acc += (p[0] + p[1]) + (p[2] + p[3]);
        }
      }
    acc = wave_sum(acc);
    const float c = tanhf(acc + (float)d.K * (s0 + wk.sq[n]));
    const float sc = c * tw_count(t, d.T, d.warp_type, win);
    if (lane == 0) {
      c_out[pos] = c;
      scale[pos] = sc;
    }
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k)
      if (k < d.K) {
        tw_bf16* row = out + (((size_t)n * d.K + k) * d.T + t) * d.w + 4 * lane;
#pragma unroll
        for (int g = 0; g < G4; ++g) {
          const f32x4 o = h[k][g] * sc;
          *reinterpret_cast<tw_bf16x4*>(row + 256 * g) = tw_bf16x4{f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
        }
      }
    tab_cur = tab_nxt;
  }
}

template <int G4>
__global__ __launch_bounds__(256) void tw_bwd_shadow_kernel(fvta_timewarp_desc d, int win, const unsigned long long* __restrict__ table,
                                                            const float* __restrict__ d_warp, const float* __restrict__ c_saved,
                                                            TwWork wk, float* __restrict__ d_hall) {
  const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6), nwv = gridDim.x * 4;
  const size_t nkt = (size_t)d.N * d.K * d.T;
  const int dp = d.w / 2;
  f32x4 v[G4], dvacc[G4];
#pragma unroll
  for (int g = 0; g < G4; ++g) {
    v[g] = ld4t(wk.v + 256 * g + 4 * lane);
    dvacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int npos = d.N * d.T;
  unsigned long long tab_cur = wv < npos ? tw_tab_load(table, nkt, d.K, d.T, wv, lane) : 0ull;
  for (int pos = wv; pos < npos; pos += nwv) {
    const int n = pos / d.T, t = pos % d.T;
    const unsigned long long tab_nxt = tw_tab_load(table, nkt, d.K, d.T, min(pos + nwv, npos - 1), lane);
    f32x4 h[TW_KMAX][G4], gw[TW_KMAX][G4];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k) {
      const int kc = min(k, d.K - 1);   // (unconditional loads)
      const unsigned long long a0 = tw_bcast64(tab_cur, kc), a1 = tw_bcast64(tab_cur, d.K + kc);
      const size_t ro = (((size_t)n * d.K + kc) * d.T + t) * d.w + 4 * lane;
#pragma unroll
      for (int g = 0; g < G4; ++g) {
        h[k][g] = 256 * g < dp ? tw_ld_bf4(a0, 256 * g + 4 * lane) : tw_ld_bf4(a1, 256 * g - dp + 4 * lane);
        gw[k][g] = ld4t(d_warp + ro + 256 * g);
      }
    }
    tab_cur = tab_nxt;
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k)
      if (k < d.K) {
        float dk = 0.f;
#pragma unroll
        for (int g = 0; g < G4; ++g) {
          const f32x4 p = h[k][g] * gw[k][g];
          dk += (p[0] + p[1]) + (p[2] + p[3]);
        }
        acc += wave_sum(dk);
      }
    const float c = c_saved[pos], cnt = tw_count(t, d.T, d.warp_type, win);
    const float dz = acc * cnt * (1.f - c * c);
    const float sc = c * cnt;
    if (lane == 0) wk.dz[pos] = dz;
#pragma unroll
    for (int k = 0; k < TW_KMAX; ++k)
      if (k < d.K) {
        float* row = d_hall + (((size_t)n * d.K + k) * d.T + t) * d.w + 4 * lane;
#pragma unroll
        for (int g = 0; g < G4; ++g) {
          *reinterpret_cast<f32x4*>(row + 256 * g) = gw[k][g] * sc + h[k][g] * v[g] * (2.f * dz);
          dvacc[g] += h[k][g] * h[k][g] * dz;
        }
      }
  }
#pragma unroll
  for (int g = 0; g < G4; ++g) *reinterpret_cast<f32x4*>(wk.dvp + (size_t)wv * d.w + 256 * g + 4 * lane) = dvacc[g];
}

// dv[c] = sum over workgroups; dsq[n] = K sum_t dz.  grid w + N, 256 threads: one output per block, strided partial
// sums folded by a fixed tree (a single thread walking all partials was a ~1000-deep chain of L2 latencies)
__global__ __launch_bounds__(256) void tw_reduce_kernel(fvta_timewarp_desc d, TwWork wk) {
  __shared__ float s_red[256];
  const int o = blockIdx.x, tid = threadIdx.x;
  float acc = 0.f;
  if (o < d.w) {
    for (int g = tid; g < wk.nwg; g += 256) acc += wk.dvp[(size_t)g * d.w + o];
  } else {
    const int n = o - d.w;
    for (int t = tid; t < d.T; t += 256) acc += wk.dz[(size_t)n * d.T + t];
  }
  s_red[tid] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) s_red[tid] += s_red[tid + st];
    __syncthreads();
  }
  if (tid == 0) {
    if (o < d.w) wk.dv[o] = s_red[0];
    else wk.dsq[o - d.w] = s_red[0] * (float)d.K;
  }
}

// parameter gradients (accumulated) and d_lq (accumulated).  grid w, 256 threads: block o handles column o
__global__ __launch_bounds__(256) void tw_param_bwd_kernel(fvta_timewarp_desc d, const float* __restrict__ WH,
                                                           const float* __restrict__ WHb, const float* __restrict__ WC,
                                                           const float* __restrict__ lq, TwWork wk,
                                                           float* __restrict__ d_lq, float* __restrict__ dWH,
                                                           float* __restrict__ dWHb, float* __restrict__ dWC,
                                                           float* __restrict__ dWCb) {
  __shared__ float s_red[4];
  const int o = blockIdx.x, tid = threadIdx.x, w = d.w;
  float ds0 = 0.f;  // = sum_n dsq[n]
  for (int n = 0; n < d.N; ++n) ds0 += wk.dsq[n];
  // dWC[o] = sum_c dv[c] WH[c][o] + ds0 WHb[o] + sum_n dsq[n] lq[n][o]
  float acc = 0.f;
  for (int c = tid; c < w; c += 256) acc += wk.dv[c] * WH[(size_t)c * w + o];
  for (int n = tid; n < d.N; n += 256) acc += wk.dsq[n] * lq[(size_t)n * w + o];
  acc = wave_sum(acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  const float wco = WC[o];
  if (tid == 0) {
    dWC[o] += (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]) + ds0 * WHb[o];
    dWHb[o] += ds0 * wco;
    if (o == 0) dWCb[0] += ds0;
  }
  for (int c = tid; c < w; c += 256) dWH[(size_t)c * w + o] += wk.dv[c] * wco;  // rows w..2w of WH see a zero feature
  for (int n = tid; n < d.N; n += 256) d_lq[(size_t)n * w + o] += wk.dsq[n] * wco;
}

}  // namespace fvta
using namespace fvta;

static int check_tw(const fvta_timewarp_desc* d) {
  FVTA_CHECK_ARG(d && d->N > 0 && d->K > 0 && d->T > 0 && d->w > 0 && d->w % 4 == 0, "timewarp: bad descriptor");
  if (d->warp_type < 1 || d->warp_type > 5) {
    fvta_set_error("time warping type not implemented (warp_type=%d)", d->warp_type);  // model_v2.py:341
    return FVTA_ERR_INVALID_ARG;
  }
  return FVTA_OK;
}

extern "C" size_t fvta_timewarp_workspace_bytes(const fvta_timewarp_desc* d) {
  if (check_tw(d)) return 0;
  return tw_work(d, nullptr).bytes;
}

extern "C" int fvta_timewarp_fwd(const fvta_timewarp_desc* d, const float* hall, const float* lq, const float* WH_W,
                                 const float* WH_b, const float* WC_W, const float* WC_b, float* warp_h, float* c_out,
                                 float* scale_out, void* workspace, fvta_stream_t stream_) {
  if (int e = check_tw(d)) return e;
  FVTA_CHECK_ARG(hall && lq && WH_W && WH_b && WC_W && WC_b && warp_h && c_out && scale_out && workspace,
                 "timewarp_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream_;
  TwWork wk = tw_work(d, workspace);
  const int win = (int)ceilf(d->window_t);
  hipLaunchKernelGGL(tw_vec_kernel, dim3((d->w + d->N + 1 + 3) / 4), dim3(256), 0, s, d->N, d->w, WH_W, WH_b, WC_W, WC_b, lq,
                     wk);
  if (d->w % 256 == 0 && d->w <= 2048 && d->K <= TW_KMAX) {  // one pass: c, the row scale and the scaled rows
    const dim3 g(wk.nwg / 4);
    switch (d->w / 256) {
      case 1: hipLaunchKernelGGL(tw_fwd_fused_kernel<1>, g, dim3(256), 0, s, *d, win, hall, wk, c_out, scale_out, warp_h); break;
      case 2: hipLaunchKernelGGL(tw_fwd_fused_kernel<2>, g, dim3(256), 0, s, *d, win, hall, wk, c_out, scale_out, warp_h); break;
      case 4: hipLaunchKernelGGL(tw_fwd_fused_kernel<4>, g, dim3(256), 0, s, *d, win, hall, wk, c_out, scale_out, warp_h); break;
      case 8: hipLaunchKernelGGL(tw_fwd_fused_kernel<8>, g, dim3(256), 0, s, *d, win, hall, wk, c_out, scale_out, warp_h); break;
      default: goto separate_fwd;
    }
    FVTA_CHECK_LAUNCH("timewarp_fwd");
    return FVTA_OK;
  }
separate_fwd:
  hipLaunchKernelGGL(tw_coef_kernel, dim3((d->N * d->T + 3) / 4), dim3(256), 0, s, *d, win, hall, wk, c_out, scale_out);
  const size_t total4 = (size_t)d->N * d->K * d->T * (d->w / 4);
  hipLaunchKernelGGL(tw_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, *d, hall, scale_out, warp_h);
  FVTA_CHECK_LAUNCH("timewarp_fwd");
  return FVTA_OK;
}

extern "C" int fvta_timewarp_bwd(const fvta_timewarp_desc* d, const float* hall, const float* lq, const float* WH_W,
                                 const float* WH_b, const float* WC_W, const float* WC_b, const float* c_saved,
                                 const float* d_warp, float* d_hall, float* d_lq, float* dWH_W, float* dWH_b,
                                 float* dWC_W, float* dWC_b, void* workspace, fvta_stream_t stream_) {
  return fvta_timewarp_bwd_att(d, hall, lq, WH_W, WH_b, WC_W, WC_b, c_saved, d_warp, nullptr, d_hall, d_lq, dWH_W, dWH_b,
                               dWC_W, dWC_b, workspace, stream_);
}

extern "C" int fvta_timewarp_bwd_att(const fvta_timewarp_desc* d, const float* hall, const float* lq, const float* WH_W,
                                     const float* WH_b, const float* WC_W, const float* WC_b, const float* c_saved,
                                     const float* d_warp, const float* d_scale_att, float* d_hall, float* d_lq,
                                     float* dWH_W, float* dWH_b, float* dWC_W, float* dWC_b, void* workspace,
                                     fvta_stream_t stream_) {
  if (int e = check_tw(d)) return e;
  FVTA_CHECK_ARG(hall && lq && WH_W && WH_b && WC_W && WC_b && c_saved && d_warp && d_hall && d_lq && dWH_W && dWH_b &&
                     dWC_W && dWC_b && workspace,
                 "timewarp_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream_;
  TwWork wk = tw_work(d, workspace);
  const int win = (int)ceilf(d->window_t);
  // v, s0, sq are recomputed (the workspace may have been reused since the forward call)
  hipLaunchKernelGGL(tw_vec_kernel, dim3((d->w + d->N + 1 + 3) / 4), dim3(256), 0, s, d->N, d->w, WH_W, WH_b, WC_W, WC_b, lq,
                     wk);
  const size_t rows = (size_t)d->N * d->K * d->T;
  bool fused = d->w % 256 == 0 && d->w <= 2048 && d->K <= TW_KMAX;
  if (fused) {  // one pass over hall and d_warp (wk.nwg wave slots of dv partials)
    const dim3 g(wk.nwg / 4);
    switch (d->w / 256) {
      case 1: hipLaunchKernelGGL(tw_bwd_fused_kernel<1>, g, dim3(256), 0, s, *d, win, hall, d_warp, c_saved, d_scale_att, wk, d_hall); break;
      case 2: hipLaunchKernelGGL(tw_bwd_fused_kernel<2>, g, dim3(256), 0, s, *d, win, hall, d_warp, c_saved, d_scale_att, wk, d_hall); break;
      case 4: hipLaunchKernelGGL(tw_bwd_fused_kernel<4>, g, dim3(256), 0, s, *d, win, hall, d_warp, c_saved, d_scale_att, wk, d_hall); break;
      default: fused = false;  // (w = 2048: two K x w row sets do not fit the register file -- the separate kernels)
    }
  }
  if (!fused) {
    hipLaunchKernelGGL(tw_rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, *d, hall, d_warp, wk.dsk);
    hipLaunchKernelGGL(tw_dz_kernel, dim3((d->N * d->T + 255) / 256), dim3(256), 0, s, *d, win, c_saved, wk, d_scale_att);
    hipLaunchKernelGGL(tw_apply_bwd_kernel, dim3(wk.nwg), dim3(256), 0, s, *d, win, hall, d_warp, c_saved, wk, d_hall);
  }
  hipLaunchKernelGGL(tw_reduce_kernel, dim3(d->w + d->N), dim3(256), 0, s, *d, wk);
  hipLaunchKernelGGL(tw_param_bwd_kernel, dim3(d->w), dim3(256), 0, s, *d, WH_W, WH_b, WC_W, lq, wk, d_lq, dWH_W, dWH_b,
                     dWC_W, dWC_b);
  FVTA_CHECK_LAUNCH("timewarp_bwd");
  return FVTA_OK;
}

// ---- over the bf16 shadow rows (see tw_fwd_shadow_kernel) -----------------------------------------------------------------
static int check_tw_shadow(const fvta_timewarp_desc* d) {
  if (int e = check_tw(d)) return e;
  FVTA_CHECK_ARG(d->w % 512 == 0 && d->w <= 1024 && d->K <= TW_KMAX,
                 "timewarp (shadow rows): w = %d must be 512 or 1024 and K = %d at most %d", d->w, d->K, TW_KMAX);
  return FVTA_OK;
}

extern "C" int fvta_timewarp_fwd_shadow(const fvta_timewarp_desc* d, const uint64_t* table, const float* lq, const float* WH_W,
                                        const float* WH_b, const float* WC_W, const float* WC_b, uint16_t* warp_rows,
                                        float* c_out, float* scale_out, void* workspace, fvta_stream_t stream_) {
  if (int e = check_tw_shadow(d)) return e;
  FVTA_CHECK_ARG(table && lq && WH_W && WH_b && WC_W && WC_b && warp_rows && c_out && scale_out && workspace,
                 "timewarp_fwd_shadow: null pointer");
  hipStream_t s = (hipStream_t)stream_;
  TwWork wk = tw_work(d, workspace);
  const int win = (int)ceilf(d->window_t);
  hipLaunchKernelGGL(tw_vec_kernel, dim3((d->w + d->N + 1 + 3) / 4), dim3(256), 0, s, d->N, d->w, WH_W, WH_b, WC_W, WC_b, lq,
                     wk);
  const dim3 g(1024);   // (no per-wave partials in the forward: four waves per SIMD)
  const unsigned long long* tab = reinterpret_cast<const unsigned long long*>(table);
  if (d->w == 512)
    hipLaunchKernelGGL(tw_fwd_shadow_kernel<2>, g, dim3(256), 0, s, *d, win, tab, wk, c_out, scale_out, warp_rows);
  else
    hipLaunchKernelGGL(tw_fwd_shadow_kernel<4>, g, dim3(256), 0, s, *d, win, tab, wk, c_out, scale_out, warp_rows);
  FVTA_CHECK_LAUNCH("timewarp_fwd_shadow");
  return FVTA_OK;
}

extern "C" int fvta_timewarp_bwd_shadow(const fvta_timewarp_desc* d, const uint64_t* table, const float* lq, const float* WH_W,
                                        const float* WH_b, const float* WC_W, const float* WC_b, const float* c_saved,
                                        const float* d_warp, float* d_hall, float* d_lq, float* dWH_W, float* dWH_b,
                                        float* dWC_W, float* dWC_b, void* workspace, fvta_stream_t stream_) {
  if (int e = check_tw_shadow(d)) return e;
  FVTA_CHECK_ARG(table && lq && WH_W && WH_b && WC_W && WC_b && c_saved && d_warp && d_hall && d_lq && dWH_W && dWH_b &&
                     dWC_W && dWC_b && workspace,
                 "timewarp_bwd_shadow: null pointer");
  hipStream_t s = (hipStream_t)stream_;
  TwWork wk = tw_work(d, workspace);
  const int win = (int)ceilf(d->window_t);
  hipLaunchKernelGGL(tw_vec_kernel, dim3((d->w + d->N + 1 + 3) / 4), dim3(256), 0, s, d->N, d->w, WH_W, WH_b, WC_W, WC_b, lq,
                     wk);
  const dim3 g(wk.nwg / 4);
  const unsigned long long* tab = reinterpret_cast<const unsigned long long*>(table);
  if (d->w == 512)
    hipLaunchKernelGGL(tw_bwd_shadow_kernel<2>, g, dim3(256), 0, s, *d, win, tab, d_warp, c_saved, wk, d_hall);
  else
    hipLaunchKernelGGL(tw_bwd_shadow_kernel<4>, g, dim3(256), 0, s, *d, win, tab, d_warp, c_saved, wk, d_hall);
  hipLaunchKernelGGL(tw_reduce_kernel, dim3(d->w + d->N), dim3(256), 0, s, *d, wk);
  hipLaunchKernelGGL(tw_param_bwd_kernel, dim3(d->w), dim3(256), 0, s, *d, WH_W, WH_b, WC_W, lq, wk, d_lq, dWH_W, dWH_b,
                     dWC_W, dWC_b);
  FVTA_CHECK_LAUNCH("timewarp_bwd_shadow");
  return FVTA_OK;
}
