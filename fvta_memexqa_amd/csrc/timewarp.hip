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
  w.nwg = 1024;
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
  hipLaunchKernelGGL(tw_rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, *d, hall, d_warp, wk.dsk);
  hipLaunchKernelGGL(tw_dz_kernel, dim3((d->N * d->T + 255) / 256), dim3(256), 0, s, *d, win, c_saved, wk, d_scale_att);
  hipLaunchKernelGGL(tw_apply_bwd_kernel, dim3(wk.nwg), dim3(256), 0, s, *d, win, hall, d_warp, c_saved, wk, d_hall);
  hipLaunchKernelGGL(tw_reduce_kernel, dim3(d->w + d->N), dim3(256), 0, s, *d, wk);
  hipLaunchKernelGGL(tw_param_bwd_kernel, dim3(d->w), dim3(256), 0, s, *d, WH_W, WH_b, WC_W, lq, wk, d_lq, dWH_W, dWH_b,
                     dWC_W, dWC_b);
  FVTA_CHECK_LAUNCH("timewarp_bwd");
  return FVTA_OK;
}
