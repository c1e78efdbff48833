// Answer scorer + softmax cross-entropy: model_v2.py:1053-1083 and 1085-1096.
// Tiny ([N,4,5w] x [5w,1]); one workgroup per QA pair, deterministic reductions.
#include "fvta_common.h"

namespace fvta {

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// feature order of model_v2.py:1073/1075: gq, g1, gch, g1*gch, gq*gch [, (g1-gch)^2, (gq-gch)^2]
__global__ __launch_bounds__(256) void scorer_fwd_kernel(fvta_scorer_desc d, const float* __restrict__ gq,
                                                         const float* __restrict__ g1,
                                                         const float* __restrict__ gch, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ logits,
                                                         float* __restrict__ yp) {
  __shared__ float s_red[4];
  __shared__ float s_logit[64];
  const int n = blockIdx.x, tid = threadIdx.x, w = d.w;
  for (int c = 0; c < d.C; ++c) {
    float acc = 0.f;
    for (int ch = tid; ch < w; ch += 256) {
      const float q = gq[(size_t)n * w + ch], a = g1[(size_t)n * w + ch], g = gch[((size_t)n * d.C + c) * w + ch];
      float v = W[ch] * q + W[w + ch] * a + W[2 * w + ch] * g + W[3 * w + ch] * (a * g) + W[4 * w + ch] * (q * g);
      if (d.use_eu_output) v += W[5 * w + ch] * ((a - g) * (a - g)) + W[6 * w + ch] * ((q - g) * (q - g));
      acc += v;
    }
    acc = block_sum_256(acc, s_red) + b[0];
    if (d.use_eu_output && d.add_tanh) acc = tanhf(acc);
    if (tid == 0) s_logit[c] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    float mx = -INFINITY;
    for (int c = 0; c < d.C; ++c) mx = fmaxf(mx, s_logit[c]);
    float sum = 0.f;
    for (int c = 0; c < d.C; ++c) sum += expf(s_logit[c] - mx);
    for (int c = 0; c < d.C; ++c) {
      logits[n * d.C + c] = s_logit[c];
      yp[n * d.C + c] = expf(s_logit[c] - mx) / sum;
    }
  }
}

// mean over ALL N rows (model_v2.py:1090), fixed summation order
__global__ __launch_bounds__(256) void ce_loss_kernel(int N, int C, const float* __restrict__ logits,
                                                      const uint8_t* __restrict__ y, float* __restrict__ loss) {
  __shared__ float s_red[4];
  float acc = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[n * C + c]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(logits[n * C + c] - mx);
    const float lse = mx + logf(sum);
    for (int c = 0; c < C; ++c)
      if (y[n * C + c]) acc += lse - logits[n * C + c];
  }
  acc = block_sum_256(acc, s_red);
  if (threadIdx.x == 0) loss[0] = acc / (float)N;
}

__device__ __forceinline__ float dlogit_of(const fvta_scorer_desc& d, int n, int c, const float* logits,
                                           const float* yp, const uint8_t* y, float scale) {
  // TF-1's SoftmaxCrossEntropyWithLogits kernel hands back `softmax - labels` as its backprop output whatever the
  // labels sum to [TF-internal]; the gradient of -sum(y log softmax) proper is `softmax * sum(y) - y`.  They differ
  // on rows whose labels are all False: the padded rows of a short last batch (model_v2.py:1270) -- the reference
  // pushes softmax/N into the scorer from those rows.  xent_grad 0 (default) = the reference's behaviour.
  float ysum = 1.f;
  if (d.xent_grad == 1) {
    ysum = 0.f;
    for (int k = 0; k < d.C; ++k) ysum += y[n * d.C + k] ? 1.f : 0.f;
  }
  float dl = scale * (yp[n * d.C + c] * ysum - (y[n * d.C + c] ? 1.f : 0.f));
  if (d.use_eu_output && d.add_tanh) {
    const float l = logits[n * d.C + c];
    dl *= (1.f - l * l);
  }
  return dl;
}

__global__ __launch_bounds__(256) void scorer_bwd_inputs_kernel(fvta_scorer_desc d, const float* __restrict__ gq,
                                                                const float* __restrict__ g1,
                                                                const float* __restrict__ gch,
                                                                const float* __restrict__ W, const uint8_t* y,
                                                                const float* logits, const float* yp, float scale,
                                                                float* __restrict__ dgq, float* __restrict__ dg1,
                                                                float* __restrict__ dgch) {
  __shared__ float s_dl[64];
  const int n = blockIdx.x, tid = threadIdx.x, w = d.w;
  if (tid < d.C) s_dl[tid] = dlogit_of(d, n, tid, logits, yp, y, scale);
  __syncthreads();
  for (int ch = tid; ch < w; ch += 256) {
    const float q = gq[(size_t)n * w + ch], a = g1[(size_t)n * w + ch];
    float dq = 0.f, da = 0.f;
    for (int c = 0; c < d.C; ++c) {
      const float g = gch[((size_t)n * d.C + c) * w + ch], dl = s_dl[c];
      float tq = W[ch] + W[4 * w + ch] * g;
      float ta = W[w + ch] + W[3 * w + ch] * g;
      float tg = W[2 * w + ch] + W[3 * w + ch] * a + W[4 * w + ch] * q;
      if (d.use_eu_output) {
        const float ea = 2.f * W[5 * w + ch] * (a - g), eq = 2.f * W[6 * w + ch] * (q - g);
        ta += ea;
        tq += eq;
        tg -= ea + eq;
      }
      dq += dl * tq;
      da += dl * ta;
      dgch[((size_t)n * d.C + c) * w + ch] = dl * tg;
    }
    dgq[(size_t)n * w + ch] = dq;
    dg1[(size_t)n * w + ch] = da;
  }
}

// dW [F] += sum_n sum_c dl * feat ; db += sum dl.   grid ceil(w/64); 256 threads = 64 channels x 4 groups of n.
// Fixed summation order (bitwise reproducible); dl is computed once per block into LDS.
constexpr int SCORER_DL_MAX = 4096;
__global__ __launch_bounds__(256) void scorer_bwd_params_kernel(fvta_scorer_desc d, const float* __restrict__ gq,
                                                                const float* __restrict__ g1,
                                                                const float* __restrict__ gch, const uint8_t* y,
                                                                const float* logits, const float* yp, float scale,
                                                                float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float s_dl[SCORER_DL_MAX];
  __shared__ float s_p[4][7][64];
  const int tid = threadIdx.x, w = d.w, NC = d.N * d.C;
  const bool cached = NC <= SCORER_DL_MAX;
  if (cached)
    for (int i = tid; i < NC; i += 256) s_dl[i] = dlogit_of(d, i / d.C, i % d.C, logits, yp, y, scale);
  __syncthreads();
  const int cl = tid & 63, grp = tid >> 6, ch = blockIdx.x * 64 + cl;
  float s[7] = {0, 0, 0, 0, 0, 0, 0};
  auto term = [&](float dl, float q, float a, float g) {
    s[0] += dl * q;
    s[1] += dl * a;
    s[2] += dl * g;
    s[3] += dl * a * g;
    s[4] += dl * q * g;
    s[5] += dl * (a - g) * (a - g);
    s[6] += dl * (q - g) * (q - g);
  };
  if (ch < w) {
    // four n of the group per round, their loads issued together (the plain loop was one exposed load latency per n and
    // choice: 36 us on 16 workgroups); the order of the sums is unchanged
    int n = grp;
    if (d.C == 4)
      for (; n + 12 < d.N; n += 16) {
        float q[4], a[4], g[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ni = n + 4 * i;
          q[i] = gq[(size_t)ni * w + ch];
          a[i] = g1[(size_t)ni * w + ch];
#pragma unroll
          for (int c = 0; c < 4; ++c) g[i][c] = gch[((size_t)ni * 4 + c) * w + ch];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int ni = n + 4 * i;
            term(cached ? s_dl[ni * 4 + c] : dlogit_of(d, ni, c, logits, yp, y, scale), q[i], a[i], g[i][c]);
          }
      }
    for (; n < d.N; n += 4) {
      const float q = gq[(size_t)n * w + ch], a = g1[(size_t)n * w + ch];
      for (int c = 0; c < d.C; ++c)
        term(cached ? s_dl[n * d.C + c] : dlogit_of(d, n, c, logits, yp, y, scale), q, a, gch[((size_t)n * d.C + c) * w + ch]);
    }
  }
#pragma unroll
  for (int f = 0; f < 7; ++f) s_p[grp][f][cl] = s[f];
  __syncthreads();
  if (grp == 0 && ch < w) {
    const int nf = d.use_eu_output ? 7 : 5;
    for (int f = 0; f < nf; ++f) dW[f * w + ch] += (s_p[0][f][cl] + s_p[1][f][cl]) + (s_p[2][f][cl] + s_p[3][f][cl]);
  }
  if (blockIdx.x == 0) {  // d bias = sum of the logit gradients: strided partial sums, then a fixed tree (one thread walked all N C)
    __syncthreads();
    float sb = 0.f;
    for (int i = tid; i < NC; i += 256) sb += cached ? s_dl[i] : dlogit_of(d, i / d.C, i % d.C, logits, yp, y, scale);
    float* red = &s_p[0][0][0];
    red[tid] = sb;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if (tid < h) red[tid] += red[tid + h];
      __syncthreads();
    }
    if (tid == 0) db[0] += red[0];
  }
}

}  // namespace fvta
using namespace fvta;

static int check_scorer(const fvta_scorer_desc* d) {
  FVTA_CHECK_ARG(d && d->N > 0 && d->C > 0 && d->C <= 64 && d->w > 0, "scorer: bad descriptor");
  FVTA_CHECK_ARG(d->xent_grad == 0 || d->xent_grad == 1, "scorer: xent_grad must be 0 (TF kernel) or 1 (mathematical)");
  return FVTA_OK;
}

extern "C" int fvta_scorer_ce_fwd(const fvta_scorer_desc* d, const float* gq, const float* g1, const float* gch,
                                  const float* W, const float* b, const uint8_t* y, float* logits, float* yp,
                                  float* loss, fvta_stream_t stream_) {
  if (int e = check_scorer(d)) return e;
  FVTA_CHECK_ARG(gq && g1 && gch && W && b && logits && yp, "scorer_ce_fwd: null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(scorer_fwd_kernel, dim3(d->N), dim3(256), 0, stream, *d, gq, g1, gch, W, b, logits, yp);
  if (y && loss) hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), 0, stream, d->N, d->C, logits, y, loss);
  FVTA_CHECK_LAUNCH("scorer_fwd");
  return FVTA_OK;
}

extern "C" int fvta_scorer_ce_bwd(const fvta_scorer_desc* d, const float* gq, const float* g1, const float* gch,
                                  const float* W, const float* b, const uint8_t* y, const float* logits,
                                  const float* yp, float loss_scale, float* dgq, float* dg1, float* dgch, float* dW,
                                  float* db, fvta_stream_t stream_) {
  if (int e = check_scorer(d)) return e;
  FVTA_CHECK_ARG(gq && g1 && gch && W && y && logits && yp && dgq && dg1 && dgch && dW && db,
                 "scorer_ce_bwd: null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const float scale = loss_scale / (float)d->N;
  hipLaunchKernelGGL(scorer_bwd_inputs_kernel, dim3(d->N), dim3(256), 0, stream, *d, gq, g1, gch, W, y, logits, yp,
                     scale, dgq, dg1, dgch);
  hipLaunchKernelGGL(scorer_bwd_params_kernel, dim3((d->w + 63) / 64), dim3(256), 0, stream, *d, gq, g1, gch, y,
                     logits, yp, scale, dW, db);
  FVTA_CHECK_LAUNCH("scorer_bwd");
  return FVTA_OK;
}
