// AttentionGRUCell.__call__ (attention_gru_cell.py:50-70) for gfx950: one step of the DMN+
// attention-gated GRU,
//   r = sigmoid([x,h] Wg + bg);  hc = h Wc;  xi = x Wi + bi;  h_hat = tanh(r*hc + xi);
//   new_h = (1-g) h + g h_hat,          inputs = [x | g]  ([B, d+1], the gate is the LAST column).
// Its only consumer in the reference is the DMN+ baseline (model_dmnplus.py:130), with B = facts of a
// batch and d ~ 80: small, odd-strided ([B, d+1] rows are not 16-byte aligned) products, so they run on
// an LDS-tiled fp32 VALU GEMM with arbitrary element strides; the gate math is fused elementwise.
#include "fvta_common.h"

namespace fvta {

// C[M,N] (+)= A[M,K] * B[K,N] (+ bias[N]) with element strides (transposes are strides).  32x32 tile, 256 thr
struct SgArgs {
  const float *A, *B, *bias;
  float* C;
  int M, N, K;
  int64_t sam, sak, sbk, sbn, scm, scn;
  int accumulate;
};
__global__ __launch_bounds__(256) void sgemm_strided(SgArgs a) {
  __shared__ float As[32][33], Bs[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // rows ty, ty+8, ty+16, ty+24 ; column tx
  for (int k0 = 0; k0 < a.K; k0 += 32) {
    for (int r = ty; r < 32; r += 8) {
      const int m = m0 + r, k = k0 + tx;
      As[r][tx] = (m < a.M && k < a.K) ? a.A[m * a.sam + k * a.sak] : 0.f;
      const int kk = k0 + r, n = n0 + tx;
      Bs[r][tx] = (kk < a.K && n < a.N) ? a.B[kk * a.sbk + n * a.sbn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const float b = Bs[k][tx];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += As[ty + 8 * i][k] * b;
    }
    __syncthreads();
  }
  const int n = n0 + tx;
  if (n >= a.N) return;
  const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + 8 * i;
    if (m < a.M) {
      float* c = a.C + m * a.scm + n * a.scn;
      *c = (a.accumulate ? *c : 0.f) + acc[i] + bv;
    }
  }
}

static void sgemm(hipStream_t s, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                  const float* bias, float* C, int64_t scm, int64_t scn, int M, int N, int K, int accumulate) {
  SgArgs a{A, B, bias, C, M, N, K, sam, sak, sbk, sbn, scm, scn, accumulate};
  hipLaunchKernelGGL(sgemm_strided, dim3((M + 31) / 32, (N + 31) / 32), dim3(256), 0, s, a);
}

// saved [B][3d] = r_pre -> r, hc, xi -> h_hat
__global__ void attgru_gate_fwd(int B, int d, const float* __restrict__ inputs, const float* __restrict__ state,
                                float* __restrict__ saved, float* __restrict__ new_h) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * d) return;
  const int b = idx / d, u = idx % d;
  float* sv = saved + (size_t)b * 3 * d;
  const float r = 1.f / (1.f + expf(-sv[u]));
  const float hc = sv[d + u];
  const float h_hat = tanhf(r * hc + sv[2 * d + u]);
  const float g = inputs[(size_t)b * (d + 1) + d], h = state[idx];
  sv[u] = r;
  sv[2 * d + u] = h_hat;
  new_h[idx] = (1.f - g) * h + g * h_hat;
}

// ws [B][4d]: dr_pre, dhc, dxi, (h_hat - h) * d_new_h ; d_state initialised with the direct term
__global__ void attgru_gate_bwd(int B, int d, const float* __restrict__ inputs, const float* __restrict__ state,
                                const float* __restrict__ saved, const float* __restrict__ d_new_h,
                                float* __restrict__ ws, float* __restrict__ d_state) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * d) return;
  const int b = idx / d, u = idx % d;
  const float* sv = saved + (size_t)b * 3 * d;
  const float r = sv[u], hc = sv[d + u], h_hat = sv[2 * d + u];
  const float g = inputs[(size_t)b * (d + 1) + d], h = state[idx], dn = d_new_h[idx];
  const float dpre = dn * g * (1.f - h_hat * h_hat);
  float* w = ws + (size_t)b * 4 * d;
  w[u] = dpre * hc * r * (1.f - r);
  w[d + u] = dpre * r;
  w[2 * d + u] = dpre;
  w[3 * d + u] = dn * (h_hat - h);
  d_state[idx] = dn * (1.f - g);
}

// d_inputs[b][d] = sum_u ws[b][3d+u] ; the x part of d_inputs is zeroed here (GEMMs accumulate into it)
__global__ void attgru_dgate(int B, int d, const float* __restrict__ ws, float* __restrict__ d_inputs) {
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int u = threadIdx.x; u < d; u += 64) {
    acc += ws[(size_t)b * 4 * d + 3 * d + u];
    d_inputs[(size_t)b * (d + 1) + u] = 0.f;
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) d_inputs[(size_t)b * (d + 1) + d] = acc;
}

// dbias[n] += sum_b ws[b][col0 + n]
__global__ void attgru_colsum(int B, int d, const float* __restrict__ ws, int col0, float* __restrict__ dbias) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= d) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += ws[(size_t)b * 4 * d + col0 + n];
  dbias[n] += acc;
}

}  // namespace fvta
using namespace fvta;

extern "C" int fvta_attgru_fwd(int32_t B, int32_t d, const float* inputs, const float* state, const float* Wg,
                               const float* bg, const float* Wc, const float* Wi, const float* bi, float* new_h,
                               float* saved, fvta_stream_t stream_) {
  FVTA_CHECK_ARG(B > 0 && d > 0, "attgru_fwd: bad B/d");
  FVTA_CHECK_ARG(inputs && state && Wg && bg && Wc && Wi && bi && new_h && saved, "attgru_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream_;
  const int64_t si = d + 1, ss = 3 * d;
  // r_pre = x Wg[:d] + h Wg[d:] + bg   (attention_gru_cell.py:63)
  sgemm(s, inputs, si, 1, Wg, d, 1, bg, saved, ss, 1, B, d, d, 0);
  sgemm(s, state, d, 1, Wg + (size_t)d * d, d, 1, nullptr, saved, ss, 1, B, d, d, 1);
  sgemm(s, state, d, 1, Wc, d, 1, nullptr, saved + d, ss, 1, B, d, d, 0);   // hc  (:66, no bias)
  sgemm(s, inputs, si, 1, Wi, d, 1, bi, saved + 2 * d, ss, 1, B, d, d, 0);  // xi  (:68)
  hipLaunchKernelGGL(attgru_gate_fwd, dim3((B * d + 255) / 256), dim3(256), 0, s, B, d, inputs, state, saved, new_h);
  FVTA_CHECK_LAUNCH("attgru_fwd");
  return FVTA_OK;
}

extern "C" int fvta_attgru_bwd(int32_t B, int32_t d, const float* inputs, const float* state, const float* Wg,
                               const float* Wc, const float* Wi, const float* saved, const float* d_new_h,
                               float* d_inputs, float* d_state, float* dWg, float* dbg, float* dWc, float* dWi,
                               float* dbi, void* workspace, fvta_stream_t stream_) {
  FVTA_CHECK_ARG(B > 0 && d > 0, "attgru_bwd: bad B/d");
  FVTA_CHECK_ARG(inputs && state && Wg && Wc && Wi && saved && d_new_h && d_inputs && d_state && dWg && dbg && dWc &&
                     dWi && dbi && workspace,
                 "attgru_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream_;
  float* ws = (float*)workspace;
  const int64_t si = d + 1, sw = 4 * d;
  hipLaunchKernelGGL(attgru_gate_bwd, dim3((B * d + 255) / 256), dim3(256), 0, s, B, d, inputs, state, saved, d_new_h, ws,
                     d_state);
  hipLaunchKernelGGL(attgru_dgate, dim3(B), dim3(64), 0, s, B, d, ws, d_inputs);
  const float *dr = ws, *dhc = ws + d, *dxi = ws + 2 * d;
  // parameter gradients (accumulated): dW = A^T dY
  sgemm(s, inputs, 1, si, dr, sw, 1, nullptr, dWg, d, 1, d, d, B, 1);
  sgemm(s, state, 1, d, dr, sw, 1, nullptr, dWg + (size_t)d * d, d, 1, d, d, B, 1);
  sgemm(s, state, 1, d, dhc, sw, 1, nullptr, dWc, d, 1, d, d, B, 1);
  sgemm(s, inputs, 1, si, dxi, sw, 1, nullptr, dWi, d, 1, d, d, B, 1);
  hipLaunchKernelGGL(attgru_colsum, dim3((d + 255) / 256), dim3(256), 0, s, B, d, ws, 0, dbg);
  hipLaunchKernelGGL(attgru_colsum, dim3((d + 255) / 256), dim3(256), 0, s, B, d, ws, 2 * d, dbi);
  // input gradients (d_inputs x-part zeroed by attgru_dgate, d_state seeded with the direct term): dA = dY W^T
  sgemm(s, dr, sw, 1, Wg, 1, d, nullptr, d_inputs, si, 1, B, d, d, 1);
  sgemm(s, dxi, sw, 1, Wi, 1, d, nullptr, d_inputs, si, 1, B, d, d, 1);
  sgemm(s, dr, sw, 1, Wg + (size_t)d * d, 1, d, nullptr, d_state, d, 1, B, d, d, 1);
  sgemm(s, dhc, sw, 1, Wc, 1, d, nullptr, d_state, d, 1, B, d, d, 1);
  FVTA_CHECK_LAUNCH("attgru_bwd");
  return FVTA_OK;
}
