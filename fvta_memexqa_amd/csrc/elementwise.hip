// Stand-alone forms of the small graph helpers the reference exposes next to attention / attention_3d:
// softmax (model_v2.py:23-28), softsel (39-48), linear (75-100), exp_mask (utils.py:210-213).  Inside the model
// these are folded into the attention / scorer / embedding kernels; the stand-alone entry points exist so that code
// written against the reference's functional surface has something to call.  Forward only.
#include "fvta_common.h"

namespace fvta {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// softmax over the last axis: one wave per row.  grid ceil(rows/4), 256 threads
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows,
                                                          int J) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* xr = x + r * J;
  float m = -INFINITY;
  for (int j = lane; j < J; j += 64) m = fmaxf(m, xr[j]);
  m = wave_max(m);
  float s = 0.f;
  for (int j = lane; j < J; j += 64) s += expf(xr[j] - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int j = lane; j < J; j += 64) y[r * J + j] = expf(xr[j] - m) * inv;
}

// softsel: out[r, :] = sum_j softmax(logits[r, :])[j] * target[r, j, :].  One workgroup per row: the weights go to
// LDS once, then thread c walks j for its channels (coalesced over c).  grid rows, 256 threads, dyn LDS J floats
__global__ __launch_bounds__(256) void softsel_kernel(const float* __restrict__ target, const float* __restrict__ logits,
                                                     float* __restrict__ out, int J, int d) {
  extern __shared__ float s_p[];
  __shared__ float s_red[4];
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* lr = logits + r * J;
  float m = -INFINITY;
  for (int j = tid; j < J; j += 256) m = fmaxf(m, lr[j]);
  m = wave_max(m);
  if (lane == 0) s_red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float s = 0.f;
  for (int j = tid; j < J; j += 256) {
    const float e = expf(lr[j] - m);
    s_p[j] = e;
    s += e;
  }
  s = wave_sum(s);
  if (lane == 0) s_red[wv] = s;
  __syncthreads();
  const float inv = 1.f / ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
  const float* tr = target + r * (int64_t)J * d;
  for (int c = tid; c < d; c += 256) {
    float acc = 0.f;
    for (int j = 0; j < J; ++j) acc += s_p[j] * tr[(int64_t)j * d + c];
    out[r * d + c] = acc * inv;
  }
}

__global__ void exp_mask_kernel(const float* __restrict__ val, const uint8_t* __restrict__ mask, float* __restrict__ out,
                                int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = val[i] + (1.f - (mask[i] ? 1.f : 0.f)) * FVTA_NEG;  // utils.py:213, literally
}

// linear: y[M, out] = x[M, in] * W[in, out] + b (+ tanh).  64 x 64 output tile, 16-deep k slices through LDS,
// fp32 FMA in k order (bit-stable).  Not a hot kernel: every linear of the model is fused into its consumer.
// x rows may live in blocks: row m starts at (m / rpb) * bstride + (m % rpb) * in  (rpb = M, bstride = 0: dense) -- one
// stream's rows inside the model.py graph's [N][all streams] arena.
__device__ __forceinline__ int64_t blk_row(int64_t m, int64_t rpb, int64_t bstride, int ld) {
  return (m / rpb) * bstride + (m % rpb) * ld;
}
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ b, float* __restrict__ y, int64_t M, int in,
                                                     int out, int add_tanh, int64_t rpb, int64_t bstride) {
  __shared__ float sx[64][17], sw[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 4 x 4 outputs each
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < in; k0 += 16) {
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int r = i >> 4, c = i & 15;
      sx[r][c] = (m0 + r < M && k0 + c < in) ? x[blk_row(m0 + r, rpb, bstride, in) + k0 + c] : 0.f;
    }
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      sw[r][c] = (k0 + r < in && n0 + c < out) ? W[(int64_t)(k0 + r) * out + n0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sx[ty * 4 + i][k];
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[j] = sw[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * bb[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t m = m0 + ty * 4 + i;
      const int n = n0 + tx * 4 + j;
      if (m < M && n < out) {
        const float v = acc[i][j] + (b ? b[n] : 0.f);
        y[m * out + n] = add_tanh ? tanhf(v) : v;
      }
    }
}

// linear backward, input side: dx[M, in] (+)= dyt[M, out] * W[in, out]^T with dyt = dy (1 - y^2) under add_tanh.
// Same 64 x 64 tiling as the forward (W read transposed).
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* __restrict__ W, const float* __restrict__ y,
                                                            const float* __restrict__ dy, float* __restrict__ dx, int64_t M,
                                                            int in, int out, int add_tanh, int accumulate, int64_t rpb,
                                                            int64_t bstride) {
  __shared__ float sd[64][17], sw[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;  // columns of dx = inputs
  float acc[4][4] = {};
  for (int k0 = 0; k0 < out; k0 += 16) {
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int r = i >> 4, c = i & 15;
      float v = 0.f;
      if (m0 + r < M && k0 + c < out) {
        v = dy[(m0 + r) * out + k0 + c];
        if (add_tanh) {
          const float yy = y[(m0 + r) * out + k0 + c];
          v *= 1.f - yy * yy;
        }
      }
      sd[r][c] = v;
    }
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
      const int c = i >> 4, r = i & 15;  // W[n0 + c][k0 + r], read along its rows
      sw[r][c] = (k0 + r < out && n0 + c < in) ? W[(int64_t)(n0 + c) * out + k0 + r] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sd[ty * 4 + i][k];
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[j] = sw[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * bb[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t m = m0 + ty * 4 + i;
      const int n = n0 + tx * 4 + j;
      if (m < M && n < in) {
        float* o = dx + blk_row(m, rpb, bstride, in) + n;
        *o = accumulate ? *o + acc[i][j] : acc[i][j];
      }
    }
}

// linear backward, parameter side: dW[in, out] += x^T dyt, db[out] += sum_m dyt.  One workgroup per 64 x 64 tile of dW
// walks all M rows in order (fixed summation order); the workgroups of the first row of tiles also fold db.
__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ dy, float* __restrict__ dW,
                                                            float* __restrict__ db, int64_t M, int in, int out, int add_tanh,
                                                            int64_t rpb, int64_t bstride) {
  __shared__ float sx[16][65], sd[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.x * 64, o0 = blockIdx.y * 64;
  float acc[4][4] = {};
  float bacc = 0.f;  // threads 0..63 of the blockIdx.x == 0 tiles: db[o0 + tid]
  for (int64_t m0 = 0; m0 < M; m0 += 16) {
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      sx[r][c] = (m0 + r < M && i0 + c < in) ? x[blk_row(m0 + r, rpb, bstride, in) + i0 + c] : 0.f;
      float v = 0.f;
      if (m0 + r < M && o0 + c < out) {
        v = dy[(m0 + r) * out + o0 + c];
        if (add_tanh) {
          const float yy = y[(m0 + r) * out + o0 + c];
          v *= 1.f - yy * yy;
        }
      }
      sd[r][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sx[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[j] = sd[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * bb[j];
    }
    if (blockIdx.x == 0 && threadIdx.x < 64)
#pragma unroll
      for (int k = 0; k < 16; ++k) bacc += sd[k][threadIdx.x];
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = i0 + ty * 4 + i, c = o0 + tx * 4 + j;
      if (r < in && c < out) dW[(int64_t)r * out + c] += acc[i][j];
    }
  if (db && blockIdx.x == 0 && threadIdx.x < 64 && o0 + (int)threadIdx.x < out) db[o0 + threadIdx.x] += bacc;
}

// weighted sum without a softmax: out[r, :] = sum_j weights[r, j] * target[r, j, :]  (attention_tgif, model.py:236-238:
// the weights there are softmax(score) with exp_mask applied AFTERWARDS).  grid rows, 256 threads
__global__ __launch_bounds__(256) void wsum_kernel(const float* __restrict__ target, const float* __restrict__ weights,
                                                   float* __restrict__ out, int J, int d, int64_t t_ld) {
  const int64_t r = blockIdx.x;
  const float* tr = target + r * t_ld;
  const float* wr = weights + r * J;
  for (int c = threadIdx.x; c < d; c += 256) {
    float acc = 0.f;
    for (int j = 0; j < J; ++j) acc += wr[j] * tr[(int64_t)j * d + c];
    out[r * d + c] = acc;
  }
}
// backward of wsum: d_weights[r, j] = target[r, j, :] . d_out[r, :] (overwritten), d_target[r, j, :] += weights[r, j] d_out[r, :].
// grid (rows, ceil(J / 4)): a wave per j
__global__ __launch_bounds__(256) void wsum_bwd_kernel(const float* __restrict__ target, const float* __restrict__ weights,
                                                       const float* __restrict__ d_out, float* __restrict__ d_weights,
                                                       float* __restrict__ d_target, int J, int d, int64_t t_ld) {
  const int64_t r = blockIdx.x;
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= J) return;
  const float* tr = target + r * t_ld + (int64_t)j * d;
  float* dt = d_target ? d_target + r * t_ld + (int64_t)j * d : nullptr;
  const float* g = d_out + r * d;
  const float wv = weights[r * J + j];
  float acc = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float gc = g[c];
    acc += tr[c] * gc;
    if (dt) dt[c] += wv * gc;
  }
  acc = wave_sum(acc);
  if (lane == 0 && d_weights) d_weights[r * J + j] = acc;
}
// backward of softmax over the last axis: dx = p (dp - sum_j p dp).  One wave per row.  grid ceil(rows/4)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                          float* __restrict__ dx, int64_t rows, int J) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  float dot = 0.f;
  for (int j = lane; j < J; j += 64) dot += p[r * J + j] * dp[r * J + j];
  dot = wave_sum(dot);
  for (int j = lane; j < J; j += 64) dx[r * J + j] = p[r * J + j] * (dp[r * J + j] - dot);
}
// DMN+ episode attention features (model_dmnplus.py:93-98): out[n,f,:] = [fact*q, fact*m, |fact-q|, |fact-m|].
// grid N*F, 256 threads
__global__ __launch_bounds__(256) void dmn_features_kernel(const float* __restrict__ facts, const float* __restrict__ q,
                                                          const float* __restrict__ m, float* __restrict__ out, int F, int d) {
  const int64_t row = blockIdx.x;
  const int n = (int)(row / F);
  const float* f = facts + row * d;
  const float* qv = q + (size_t)n * d;
  const float* mv = m + (size_t)n * d;
  float* o = out + row * 4 * d;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float fv = f[c], a = qv[c], b = mv[c];
    o[c] = fv * a;
    o[d + c] = fv * b;
    o[2 * d + c] = fabsf(fv - a);
    o[3 * d + c] = fabsf(fv - b);
  }
}
// backward of dmn_features_kernel.  d_facts[n,f,c] += dO1 q + dO2 m + dO3 sgn(fact-q) + dO4 sgn(fact-m) (one thread per
// element of its (n, channel) column, walking the facts), d_q[n,c] += sum_f (dO1 fact - dO3 sgn(fact-q)), d_m likewise
// (|x| has derivative 0 at 0, as tf.abs).  grid (N, ceil(d/256))
__global__ __launch_bounds__(256) void dmn_features_bwd_kernel(const float* __restrict__ facts, const float* __restrict__ q,
                                                              const float* __restrict__ m, const float* __restrict__ d_out,
                                                              float* __restrict__ d_facts, float* __restrict__ d_q,
                                                              float* __restrict__ d_m, int F, int d) {
  const int n = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= d) return;
  const float a = q[(size_t)n * d + c], b = m[(size_t)n * d + c];
  float gq = 0.f, gm = 0.f;
  for (int f = 0; f < F; ++f) {
    const size_t row = (size_t)n * F + f;
    const float fv = facts[row * d + c];
    const float* o = d_out + row * 4 * d;
    const float g1 = o[c], g2 = o[d + c], g3 = o[2 * d + c], g4 = o[3 * d + c];
    const float sa = fv > a ? 1.f : (fv < a ? -1.f : 0.f), sb = fv > b ? 1.f : (fv < b ? -1.f : 0.f);
    d_facts[row * d + c] += g1 * a + g2 * b + g3 * sa + g4 * sb;
    gq += g1 * fv - g3 * sa;
    gm += g2 * fv - g4 * sb;
  }
  d_q[(size_t)n * d + c] += gq;
  d_m[(size_t)n * d + c] += gm;
}
// ---- LSTM input dropout (DropoutWrapper(cell, input_keep_prob), model_v2.py:657-661).  bidirectional_dynamic_rnn calls the
// wrapped cell in two loops, so every (sequence, position) input is dropped twice, independently: x2[0][e] for the forward
// direction, x2[1][e] for the backward one, x2[dir][e] = x[e] * keep(dir, e) / keep_prob.  keep() is a counter-based hash
// (splitmix64 of seed + golden * (dir * n + e + 1), top 32 bits < keep_prob * 2^32): TensorFlow's random stream cannot be
// reproduced, its distribution is; the oracle evaluates the same hash (oracle/fvta_fused.py dropout_keep_masks).
__global__ __launch_bounds__(256) void dropout_pair_fwd_kernel(const float* __restrict__ x, float* __restrict__ x2, int64_t n,
                                                              float scale, unsigned long long thr, unsigned long long seed) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float v = x[e] * scale;
  x2[e] = dropout_keep(seed, (unsigned long long)e, thr) ? v : 0.f;
  x2[n + e] = dropout_keep(seed, (unsigned long long)(n + e), thr) ? v : 0.f;
}
__global__ __launch_bounds__(256) void dropout_pair_bwd_kernel(const float* __restrict__ dx2, float* __restrict__ dx, int64_t n,
                                                              float scale, unsigned long long thr, unsigned long long seed,
                                                              int accumulate) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float g = (dropout_keep(seed, (unsigned long long)e, thr) ? dx2[e] * scale : 0.f) +
                  (dropout_keep(seed, (unsigned long long)(n + e), thr) ? dx2[n + e] * scale : 0.f);
  dx[e] = accumulate ? dx[e] + g : g;
}
// relu and its backward (tf.layers.dense(..., activation=tf.nn.relu), model_dmnplus.py:511-514)
__global__ __launch_bounds__(256) void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = fmaxf(x[i], 0.f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
// out[r, :] (+)= scale * sum_j x[r, j, :]  (tf.reduce_mean over an inner axis, model.py:874-885, :907; with scale 1
// the backward of a tile).  grid (rows, ceil(d/256))
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ x, float* __restrict__ out, int J, int d,
                                                          int64_t out_ld, float scale, int accumulate) {
  const int64_t r = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= d) return;
  const float* xr = x + r * (int64_t)J * d + c;
  float acc = 0.f;
  for (int j = 0; j < J; ++j) acc += xr[(int64_t)j * d];
  float* o = out + r * out_ld + c;
  *o = accumulate ? *o + scale * acc : scale * acc;
}
// out[r, j, :] (+)= scale * v[r, :]  (tf.tile along an inner axis; with scale 1/J the backward of reduce_mean).
// grid (rows * J, ceil(d/256))
__global__ __launch_bounds__(256) void rows_broadcast_kernel(const float* __restrict__ v, float* __restrict__ out, int J, int d,
                                                             int64_t v_ld, float scale, int accumulate) {
  const int64_t rj = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= d) return;
  const float val = scale * v[(rj / J) * v_ld + c];
  float* o = out + rj * d + c;
  *o = accumulate ? *o + val : val;
}
}  // namespace fvta

extern "C" int fvta_linear_bwd_blk(const float* x, const float* W, const float* y, const float* dy, float* dx, float* dW,
                                   float* db, int64_t M, int32_t in, int32_t out, int32_t add_tanh, int32_t accumulate_dx,
                                   int64_t rows_per_blk, int64_t blk_stride, fvta_stream_t stream) {
  FVTA_CHECK_ARG(W && dy && M > 0 && in > 0 && out > 0 && (!add_tanh || y), "linear_bwd: bad arguments");
  FVTA_CHECK_ARG((dW == nullptr) || x, "linear_bwd: dW wants x");
  FVTA_CHECK_ARG(rows_per_blk > 0 && (rows_per_blk >= M || blk_stride >= rows_per_blk * in), "linear_bwd: bad row blocks");
  if (dx) {
    hipLaunchKernelGGL(fvta::linear_bwd_dx_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)((in + 63) / 64)), dim3(256), 0,
                       (hipStream_t)stream, W, y, dy, dx, M, in, out, add_tanh, accumulate_dx, rows_per_blk, blk_stride);
  }
  if (dW) {
    hipLaunchKernelGGL(fvta::linear_bwd_dw_kernel, dim3((unsigned)((in + 63) / 64), (unsigned)((out + 63) / 64)), dim3(256), 0,
                       (hipStream_t)stream, x, y, dy, dW, db, M, in, out, add_tanh, rows_per_blk, blk_stride);
  }
  FVTA_CHECK_LAUNCH("linear_bwd");
  return FVTA_OK;
}

extern "C" int fvta_linear_bwd(const float* x, const float* W, const float* y, const float* dy, float* dx, float* dW,
                               float* db, int64_t M, int32_t in, int32_t out, int32_t add_tanh, int32_t accumulate_dx,
                               fvta_stream_t stream) {
  return fvta_linear_bwd_blk(x, W, y, dy, dx, dW, db, M, in, out, add_tanh, accumulate_dx, M, 0, stream);
}

extern "C" int fvta_rows_reduce(const float* x, float* out, int64_t rows, int32_t J, int32_t d, int64_t out_ld, float scale,
                                int32_t accumulate, fvta_stream_t stream) {
  FVTA_CHECK_ARG(x && out && rows > 0 && rows < (1ll << 31) && J > 0 && d > 0 && out_ld >= d, "rows_reduce: bad arguments");
  hipLaunchKernelGGL(fvta::rows_reduce_kernel, dim3((unsigned)rows, (unsigned)((d + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, out, J, d, out_ld, scale, accumulate);
  FVTA_CHECK_LAUNCH("rows_reduce");
  return FVTA_OK;
}

extern "C" int fvta_rows_broadcast(const float* v, float* out, int64_t rows, int32_t J, int32_t d, int64_t v_ld, float scale,
                                   int32_t accumulate, fvta_stream_t stream) {
  FVTA_CHECK_ARG(v && out && rows > 0 && J > 0 && rows * J < (1ll << 31) && d > 0 && v_ld >= d, "rows_broadcast: bad arguments");
  hipLaunchKernelGGL(fvta::rows_broadcast_kernel, dim3((unsigned)(rows * J), (unsigned)((d + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, v, out, J, d, v_ld, scale, accumulate);
  FVTA_CHECK_LAUNCH("rows_broadcast");
  return FVTA_OK;
}

extern "C" int fvta_dmn_features(const float* facts, const float* q, const float* m, float* out, int32_t N, int32_t F,
                                 int32_t d, fvta_stream_t stream) {
  FVTA_CHECK_ARG(facts && q && m && out && N > 0 && F > 0 && d > 0, "dmn_features: bad arguments");
  hipLaunchKernelGGL(fvta::dmn_features_kernel, dim3((unsigned)((size_t)N * F)), dim3(256), 0, (hipStream_t)stream, facts, q, m,
                     out, F, d);
  FVTA_CHECK_LAUNCH("dmn_features");
  return FVTA_OK;
}

extern "C" int fvta_dmn_features_bwd(const float* facts, const float* q, const float* m, const float* d_out, float* d_facts,
                                     float* d_q, float* d_m, int32_t N, int32_t F, int32_t d, fvta_stream_t stream) {
  FVTA_CHECK_ARG(facts && q && m && d_out && d_facts && d_q && d_m && N > 0 && F > 0 && d > 0, "dmn_features_bwd: bad arguments");
  hipLaunchKernelGGL(fvta::dmn_features_bwd_kernel, dim3((unsigned)N, (unsigned)((d + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, facts, q, m, d_out, d_facts, d_q, d_m, F, d);
  FVTA_CHECK_LAUNCH("dmn_features_bwd");
  return FVTA_OK;
}

extern "C" int fvta_dropout_pair_fwd(const float* x, float* x2, int64_t n, float keep_prob, uint64_t seed, fvta_stream_t stream) {
  FVTA_CHECK_ARG(x && x2 && n > 0 && keep_prob > 0.f && keep_prob <= 1.f, "dropout_pair_fwd: bad arguments (0 < keep_prob <= 1)");
  hipLaunchKernelGGL(fvta::dropout_pair_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x2, n,
                     1.0f / keep_prob, dropout_thr(keep_prob), (unsigned long long)seed);
  FVTA_CHECK_LAUNCH("dropout_pair_fwd");
  return FVTA_OK;
}

extern "C" int fvta_dropout_pair_bwd(const float* dx2, float* dx, int64_t n, float keep_prob, uint64_t seed, int32_t accumulate,
                                     fvta_stream_t stream) {
  FVTA_CHECK_ARG(dx2 && dx && n > 0 && keep_prob > 0.f && keep_prob <= 1.f, "dropout_pair_bwd: bad arguments (0 < keep_prob <= 1)");
  hipLaunchKernelGGL(fvta::dropout_pair_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx2, dx, n,
                     1.0f / keep_prob, dropout_thr(keep_prob), (unsigned long long)seed, accumulate);
  FVTA_CHECK_LAUNCH("dropout_pair_bwd");
  return FVTA_OK;
}

extern "C" int fvta_relu_fwd(const float* x, float* y, int64_t n, fvta_stream_t stream) {
  FVTA_CHECK_ARG(x && y && n > 0, "relu_fwd: bad arguments");
  hipLaunchKernelGGL(fvta::relu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  FVTA_CHECK_LAUNCH("relu_fwd");
  return FVTA_OK;
}

extern "C" int fvta_relu_bwd(const float* y, const float* dy, float* dx, int64_t n, fvta_stream_t stream) {
  FVTA_CHECK_ARG(y && dy && dx && n > 0, "relu_bwd: bad arguments");
  hipLaunchKernelGGL(fvta::relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, n);
  FVTA_CHECK_LAUNCH("relu_bwd");
  return FVTA_OK;
}

extern "C" int fvta_wsum_fwd_ld(const float* target, const float* weights, float* out, int64_t rows, int32_t J, int32_t d,
                                int64_t target_ld, fvta_stream_t stream) {
  FVTA_CHECK_ARG(target && weights && out && rows > 0 && rows < (1ll << 31) && J > 0 && d > 0 && target_ld >= (int64_t)J * d,
                 "wsum_fwd: bad arguments");
  hipLaunchKernelGGL(fvta::wsum_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, target, weights, out, J, d,
                     target_ld);
  FVTA_CHECK_LAUNCH("wsum");
  return FVTA_OK;
}

extern "C" int fvta_wsum_fwd(const float* target, const float* weights, float* out, int64_t rows, int32_t J, int32_t d,
                             fvta_stream_t stream) {
  return fvta_wsum_fwd_ld(target, weights, out, rows, J, d, (int64_t)J * d, stream);
}

extern "C" int fvta_wsum_bwd(const float* target, const float* weights, const float* d_out, float* d_weights, float* d_target,
                             int64_t rows, int32_t J, int32_t d, int64_t target_ld, fvta_stream_t stream) {
  FVTA_CHECK_ARG(target && weights && d_out && rows > 0 && rows < (1ll << 31) && J > 0 && d > 0 && target_ld >= (int64_t)J * d,
                 "wsum_bwd: bad arguments");
  hipLaunchKernelGGL(fvta::wsum_bwd_kernel, dim3((unsigned)rows, (unsigned)((J + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     target, weights, d_out, d_weights, d_target, J, d, target_ld);
  FVTA_CHECK_LAUNCH("wsum_bwd");
  return FVTA_OK;
}

extern "C" int fvta_softmax_bwd(const float* p, const float* dp, float* dx, int64_t rows, int32_t J, fvta_stream_t stream) {
  FVTA_CHECK_ARG(p && dp && dx && rows > 0 && J > 0, "softmax_bwd: bad arguments");
  hipLaunchKernelGGL(fvta::softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp, dx,
                     rows, J);
  FVTA_CHECK_LAUNCH("softmax_bwd");
  return FVTA_OK;
}

extern "C" int fvta_softmax_fwd(const float* logits, float* out, int64_t rows, int32_t J, fvta_stream_t stream) {
  FVTA_CHECK_ARG(logits && out && rows > 0 && J > 0, "softmax_fwd: bad arguments");
  hipLaunchKernelGGL(fvta::softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits,
                     out, rows, J);
  FVTA_CHECK_LAUNCH("softmax_rows");
  return FVTA_OK;
}

extern "C" int fvta_softsel_fwd(const float* target, const float* logits, float* out, int64_t rows, int32_t J, int32_t d,
                                fvta_stream_t stream) {
  FVTA_CHECK_ARG(target && logits && out && rows > 0 && J > 0 && d > 0, "softsel_fwd: bad arguments");
  FVTA_CHECK_ARG(J <= 16000 && rows < (1ll << 31), "softsel_fwd: J=%d > 16000 or too many rows", J);
  hipLaunchKernelGGL(fvta::softsel_kernel, dim3((unsigned)rows), dim3(256), (size_t)J * sizeof(float), (hipStream_t)stream,
                     target, logits, out, J, d);
  FVTA_CHECK_LAUNCH("softsel");
  return FVTA_OK;
}

extern "C" int fvta_exp_mask(const float* val, const uint8_t* mask, float* out, int64_t n, fvta_stream_t stream) {
  FVTA_CHECK_ARG(val && mask && out && n > 0, "exp_mask: bad arguments");
  hipLaunchKernelGGL(fvta::exp_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, val, mask,
                     out, n);
  FVTA_CHECK_LAUNCH("exp_mask");
  return FVTA_OK;
}

extern "C" int fvta_linear_fwd_blk(const float* x, const float* W, const float* b, float* y, int64_t M, int32_t in,
                                   int32_t out, int32_t add_tanh, int64_t rows_per_blk, int64_t blk_stride,
                                   fvta_stream_t stream) {
  FVTA_CHECK_ARG(x && W && y && M > 0 && in > 0 && out > 0, "linear_fwd: bad arguments");
  FVTA_CHECK_ARG(rows_per_blk > 0 && (rows_per_blk >= M || blk_stride >= rows_per_blk * in), "linear_fwd: bad row blocks");
  hipLaunchKernelGGL(fvta::linear_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)((out + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, x, W, b, y, M, in, out, add_tanh, rows_per_blk, blk_stride);
  FVTA_CHECK_LAUNCH("linear");
  return FVTA_OK;
}

extern "C" int fvta_linear_fwd(const float* x, const float* W, const float* b, float* y, int64_t M, int32_t in,
                               int32_t out, int32_t add_tanh, fvta_stream_t stream) {
  return fvta_linear_fwd_blk(x, W, b, y, M, in, out, add_tanh, M, 0, stream);
}
