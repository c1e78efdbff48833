// The reversed ("bidirect") direction of the 1-D attentions and the dense gradient of the similarity logits it needs.
//
// attention(..., bidirect=True) (model_v2.py:184-192, model.py:169-177) and attention_keeprank1(..., bidirect=True)
// (model.py:297-307) append q_a = mean_v softsel(hq, a_logits[v, :]) -- every context row attends the question, the
// attended questions are averaged over the rows -- to the max-pooled h_a.  Unlike h_a (one (t, j) pair per row carries
// gradient: attn_bwd.hip), q_a sends gradient into EVERY logit, so its backward is dense in (t, j):
//   x[t,j] = sum_c U h q + Rh.h + R2.h^2 + Cq.q + C2.q^2 + b      (attn_common.h)
//   dh[t]  = sum_j dA[t,j] U q[j] + (sum_j dA[t,j]) (Rh + 2 R2 h[t])        and symmetrically dq[j], dU, dRh, ...
// In the reference's graphs this branch only ever runs on short row lists (the K per-stream vectors, the question's or a
// choice's tokens: model.py:904, 967, 978), so one workgroup per batch row with the dA tile in LDS is all it takes; these
// are not hot kernels.  Fixed summation orders throughout (bitwise reproducible).
#include "attn_common.h"

namespace fvta {

constexpr int DENSE_MAX = 8192;  // V * JQ floats of one batch row in LDS

__device__ __forceinline__ float dn_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float dn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// p[v, j] = softmax_j(a[v, :]) into LDS (a wave per row, a lane per j: JQ <= 64), pbar[j] = mean_v p[v, j]
__device__ __forceinline__ void qside_probs(const float* __restrict__ a, float* s_p, float* s_pbar, int V, int JQ) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int v = wave; v < V; v += 4) {
    const float x = lane < JQ ? a[v * JQ + lane] : -INFINITY;
    const float m = dn_wave_max(x);
    const float e = lane < JQ ? expf(x - m) : 0.f;
    const float s = dn_wave_sum(e);
    if (lane < JQ) s_p[v * JQ + lane] = e / s;
  }
  __syncthreads();
  if (tid < JQ) {
    float acc = 0.f;
    for (int v = 0; v < V; ++v) acc += s_p[v * JQ + tid];
    s_pbar[tid] = acc / (float)V;
  }
  __syncthreads();
}

// grid R, 256 threads
__global__ __launch_bounds__(256) void attn_qside_fwd_kernel(const float* __restrict__ a_logits, const float* __restrict__ hq,
                                                            float* __restrict__ q_a, int V, int JQ, int w) {
  __shared__ float s_p[DENSE_MAX];
  __shared__ float s_pbar[64];
  const int64_t r = blockIdx.x;
  qside_probs(a_logits + r * V * JQ, s_p, s_pbar, V, JQ);
  const float* q = hq + r * JQ * w;
  for (int c = threadIdx.x; c < w; c += 256) {
    float acc = 0.f;
    for (int j = 0; j < JQ; ++j) acc += s_pbar[j] * q[(int64_t)j * w + c];
    q_a[r * w + c] = acc;
  }
}

// grid R, 256 threads: dA [R,V,JQ] overwritten, d_hq [R,JQ,w] accumulated
__global__ __launch_bounds__(256) void attn_qside_bwd_kernel(const float* __restrict__ a_logits, const float* __restrict__ hq,
                                                            const float* __restrict__ d_q_a, float* __restrict__ dA,
                                                            float* __restrict__ d_hq, int V, int JQ, int w) {
  __shared__ float s_p[DENSE_MAX];
  __shared__ float s_pbar[64], s_dpb[64];
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  qside_probs(a_logits + r * V * JQ, s_p, s_pbar, V, JQ);
  const float* q = hq + r * JQ * w;
  const float* g = d_q_a + r * w;
  for (int j = wave; j < JQ; j += 4) {  // d pbar[j] = g . hq[j]
    float acc = 0.f;
    for (int c = lane; c < w; c += 64) acc += g[c] * q[(int64_t)j * w + c];
    acc = dn_wave_sum(acc);
    if (lane == 0) s_dpb[j] = acc;
  }
  __syncthreads();
  const float invV = 1.f / (float)V;
  for (int v = wave; v < V; v += 4) {  // softmax backward per row; d p[v,j] = d pbar[j] / V
    const float p = lane < JQ ? s_p[v * JQ + lane] : 0.f;
    const float dp = lane < JQ ? s_dpb[lane] * invV : 0.f;
    const float dot = dn_wave_sum(p * dp);
    if (lane < JQ) dA[(r * V + v) * JQ + lane] = p * (dp - dot);
  }
  float* dq = d_hq + r * JQ * w;
  for (int c = tid; c < w; c += 256) {
    const float gc = g[c];
    for (int j = 0; j < JQ; ++j) dq[(int64_t)j * w + c] += s_pbar[j] * gc;
  }
}

__device__ __forceinline__ void dense_vecs(const float* __restrict__ W, int w, int simi, int feat_order, int c, float& U,
                                           float& Rh, float& R2, float& Cq, float& C2) {
  U = Rh = R2 = Cq = C2 = 0.f;
  if (simi == 1) {
    Rh = W[c]; Cq = W[w + c]; U = W[2 * w + c];
  } else if (simi == 2) {
    const float W1 = feat_order == 0 ? W[c] : W[w + c];
    const float W2 = feat_order == 0 ? W[w + c] : W[c];
    U = W1 - 2.f * W2; R2 = W2; C2 = W2;
  } else {
    Rh = W[c]; Cq = W[w + c];
    const float W2 = W[2 * w + c];
    U = W[3 * w + c] - 2.f * W2; R2 = W2; C2 = W2;
  }
}

// grid N, 256 threads.  pvec [N][5][w] per-n parameter-vector partials, pb [N] per-n bias partials
__global__ __launch_bounds__(256) void attn_logits_bwd_kernel(const float* __restrict__ hinfo, size_t hstride,
                                                             const float* __restrict__ hq, const float* __restrict__ W,
                                                             const float* __restrict__ dA, float* __restrict__ d_hinfo,
                                                             float* __restrict__ d_hq, float* __restrict__ pvec,
                                                             float* __restrict__ pb, int T, int JQ, int w, int simi,
                                                             int feat_order) {
  __shared__ float s_d[DENSE_MAX];
  __shared__ float s_rs[2048], s_cs[64];
  __shared__ float s_red[4];
  const int64_t n = blockIdx.x;
  const int tid = threadIdx.x;
  const float* h = hinfo + n * hstride;
  float* dh = d_hinfo + n * hstride;
  const float* q = hq + n * JQ * w;
  float* dq = d_hq + n * JQ * w;
  float tot = 0.f;
  for (int i = tid; i < T * JQ; i += 256) {
    const float v = dA[n * T * JQ + i];
    s_d[i] = v;
    tot += v;
  }
  tot = dn_wave_sum(tot);
  if ((tid & 63) == 0) s_red[tid >> 6] = tot;
  __syncthreads();
  if (tid == 0) pb[n] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  for (int t = tid; t < T; t += 256) {
    float acc = 0.f;
    for (int j = 0; j < JQ; ++j) acc += s_d[t * JQ + j];
    s_rs[t] = acc;
  }
  if (tid < JQ) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += s_d[t * JQ + tid];
    s_cs[tid] = acc;
  }
  __syncthreads();
  for (int c = tid; c < w; c += 256) {
    float U, Rh, R2, Cq, C2;
    dense_vecs(W, w, simi, feat_order, c, U, Rh, R2, Cq, C2);
    float dRh = 0.f, dR2 = 0.f, dCq = 0.f, dC2 = 0.f, dU = 0.f;
    for (int t = 0; t < T; ++t) {  // d h[t,c]
      const float hv = h[(int64_t)t * w + c];
      float acc = 0.f;
      for (int j = 0; j < JQ; ++j) acc += s_d[t * JQ + j] * q[(int64_t)j * w + c];
      const float rs = s_rs[t];
      dh[(int64_t)t * w + c] += U * acc + rs * (Rh + 2.f * R2 * hv);
      dRh += rs * hv;
      dR2 += rs * hv * hv;
    }
    for (int j = 0; j < JQ; ++j) {  // d q[j,c], d U
      const float qv = q[(int64_t)j * w + c];
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc += s_d[t * JQ + j] * h[(int64_t)t * w + c];
      const float cs = s_cs[j];
      dq[(int64_t)j * w + c] += U * acc + cs * (Cq + 2.f * C2 * qv);
      dU += acc * qv;
      dCq += cs * qv;
      dC2 += cs * qv * qv;
    }
    float* pv = pvec + n * VEC_COUNT * w;
    pv[VEC_U * w + c] = dU;
    pv[VEC_RH * w + c] = dRh;
    pv[VEC_R2 * w + c] = dR2;
    pv[VEC_CQ * w + c] = dCq;
    pv[VEC_C2 * w + c] = dC2;
  }
}

// fold the per-n partials in n order into dW (the reference's W layout) and db.  grid ceil(w/256) + 1
__global__ __launch_bounds__(256) void attn_logits_bwd_params_kernel(const float* __restrict__ pvec, const float* __restrict__ pb,
                                                                    float* __restrict__ dW, float* __restrict__ db, int N,
                                                                    int w, int simi, int feat_order) {
  if (blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x == 0) {
      float acc = 0.f;
      for (int n = 0; n < N; ++n) acc += pb[n];
      db[0] += acc;
    }
    return;
  }
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= w) return;
  float v[VEC_COUNT] = {0, 0, 0, 0, 0};
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int k = 0; k < VEC_COUNT; ++k) v[k] += pvec[((size_t)n * VEC_COUNT + k) * w + c];
  const float dU = v[VEC_U], dRh = v[VEC_RH], dR2 = v[VEC_R2], dCq = v[VEC_CQ], dC2 = v[VEC_C2];
  if (simi == 1) {
    dW[c] += dRh;
    dW[w + c] += dCq;
    dW[2 * w + c] += dU;
  } else if (simi == 2) {
    const float d1 = dU, d2 = -2.f * dU + dR2 + dC2;
    dW[c] += feat_order == 0 ? d1 : d2;
    dW[w + c] += feat_order == 0 ? d2 : d1;
  } else {
    dW[c] += dRh;
    dW[w + c] += dCq;
    dW[2 * w + c] += -2.f * dU + dR2 + dC2;
    dW[3 * w + c] += dU;
  }
}
}  // namespace fvta

using namespace fvta;

extern "C" int fvta_attn_qside_fwd(const float* a_logits, const float* hq, float* q_a, int32_t R, int32_t V, int32_t JQ,
                                   int32_t w, fvta_stream_t stream) {
  FVTA_CHECK_ARG(a_logits && hq && q_a && R > 0 && V > 0 && JQ > 0 && w > 0, "attn_qside_fwd: bad arguments");
  FVTA_CHECK_ARG(JQ <= 64 && (int64_t)V * JQ <= DENSE_MAX, "attn_qside_fwd: JQ=%d (<= 64), V*JQ=%lld (<= %d) unsupported", JQ,
                 (long long)V * JQ, DENSE_MAX);
  hipLaunchKernelGGL(attn_qside_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, a_logits, hq, q_a, V, JQ, w);
  FVTA_CHECK_LAUNCH("attn_qside_fwd");
  return FVTA_OK;
}

extern "C" int fvta_attn_qside_bwd(const float* a_logits, const float* hq, const float* d_q_a, float* dA, float* d_hq,
                                   int32_t R, int32_t V, int32_t JQ, int32_t w, fvta_stream_t stream) {
  FVTA_CHECK_ARG(a_logits && hq && d_q_a && dA && d_hq && R > 0 && V > 0 && JQ > 0 && w > 0, "attn_qside_bwd: bad arguments");
  FVTA_CHECK_ARG(JQ <= 64 && (int64_t)V * JQ <= DENSE_MAX, "attn_qside_bwd: JQ=%d (<= 64), V*JQ=%lld (<= %d) unsupported", JQ,
                 (long long)V * JQ, DENSE_MAX);
  hipLaunchKernelGGL(attn_qside_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, a_logits, hq, d_q_a, dA, d_hq, V, JQ, w);
  FVTA_CHECK_LAUNCH("attn_qside_bwd");
  return FVTA_OK;
}

extern "C" size_t fvta_attn_logits_bwd_workspace_bytes(const fvta_attn_desc* d) {
  if (!d || d->N <= 0 || d->w <= 0) return 0;
  return fvta_align_up(((size_t)d->N * VEC_COUNT * d->w + d->N) * sizeof(float), 256);
}

extern "C" int fvta_attn_logits_bwd(const fvta_attn_desc* d, const float* hinfo, const float* hq, const float* W,
                                    const float* dA, float* d_hinfo, float* d_hq, float* dW, float* db, void* workspace,
                                    fvta_stream_t stream) {
  FVTA_CHECK_ARG(d && hinfo && hq && W && dA && d_hinfo && d_hq && dW && db && workspace, "attn_logits_bwd: null pointer");
  FVTA_CHECK_ARG(d->K == 1 && d->add_tanh == 0 && d->simi >= 1 && d->simi <= 3,
                 "attn_logits_bwd: K == 1, no tanh, simiMatrix 1-3 (K=%d add_tanh=%d simi=%d)", d->K, d->add_tanh, d->simi);
  FVTA_CHECK_ARG(d->JQ <= 64 && d->T <= 2048 && (int64_t)d->T * d->JQ <= DENSE_MAX,
                 "attn_logits_bwd: JQ=%d (<= 64), T=%d (<= 2048), T*JQ <= %d", d->JQ, d->T, DENSE_MAX);
  float* pvec = (float*)workspace;
  float* pb = pvec + (size_t)d->N * VEC_COUNT * d->w;
  const size_t hstride = d->hinfo_stride ? (size_t)d->hinfo_stride : (size_t)d->T * d->w;
  hipLaunchKernelGGL(attn_logits_bwd_kernel, dim3(d->N), dim3(256), 0, (hipStream_t)stream, hinfo, hstride, hq, W, dA, d_hinfo,
                     d_hq, pvec, pb, d->T, d->JQ, d->w, d->simi, d->feat_order);
  hipLaunchKernelGGL(attn_logits_bwd_params_kernel, dim3((d->w + 255) / 256 + 1), dim3(256), 0, (hipStream_t)stream, pvec, pb,
                     dW, db, d->N, d->w, d->simi, d->feat_order);
  FVTA_CHECK_LAUNCH("attn_logits_bwd");
  return FVTA_OK;
}
