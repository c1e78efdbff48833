// Parameter update over the flat fp32 parameter buffer: trainer.py:16
// (tf.train.AdadeltaOptimizer) and the commented-out Adam of trainer.py:17.
#include "fvta_common.h"

namespace fvta {
__global__ void adadelta_kernel(float* __restrict__ var, const float* __restrict__ grad, float* __restrict__ accum,
                                float* __restrict__ accum_update, int64_t n, float lr, float rho, float eps,
                                float gscale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = grad[i] * gscale;
  const float a = rho * accum[i] + (1.f - rho) * g * g;
  const float upd = sqrtf(accum_update[i] + eps) / sqrtf(a + eps) * g;
  accum[i] = a;
  accum_update[i] = rho * accum_update[i] + (1.f - rho) * upd * upd;
  var[i] -= lr * upd;
}
__global__ void adam_kernel(float* __restrict__ var, const float* __restrict__ grad, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps,
                            float gscale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = grad[i] * gscale;
  const float mi = b1 * m[i] + (1.f - b1) * g;
  const float vi = b2 * v[i] + (1.f - b2) * g * g;
  m[i] = mi;
  v[i] = vi;
  var[i] -= lr_t * mi / (sqrtf(vi) + eps);
}
// add_wd (model_v2.py:347-354): one term wd * l2_loss(var) = wd/2 * sum(var^2) per trainable of the scope.
// One workgroup per call so that the sum has ONE fixed order (bitwise reproducible); the slices are small (<= ~3M).
__global__ __launch_bounds__(1024) void weight_decay_kernel(const float* __restrict__ p, float* __restrict__ g, int64_t n,
                                                            float coef, float* __restrict__ loss) {
  __shared__ float s_red[1024];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float v = p[i];
    if (g) g[i] += coef * v;
    acc += v * v;
  }
  if (!loss) return;
  s_red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) s_red[threadIdx.x] += s_red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] += 0.5f * coef * s_red[0];
}
}  // namespace fvta

extern "C" int fvta_weight_decay(const float* var, float* grad, int64_t n, float coef, float* loss,
                                 fvta_stream_t stream) {
  FVTA_CHECK_ARG(var && n > 0 && (grad || loss), "weight_decay: bad arguments");
  hipLaunchKernelGGL(fvta::weight_decay_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, var, grad, n, coef, loss);
  FVTA_CHECK_LAUNCH("weight_decay");
  return FVTA_OK;
}

extern "C" int fvta_adadelta_step(float* var, const float* grad, float* accum, float* accum_update, int64_t n,
                                  float lr, float rho, float eps, float grad_scale, fvta_stream_t stream) {
  FVTA_CHECK_ARG(var && grad && accum && accum_update && n > 0, "adadelta_step: bad arguments");
  hipLaunchKernelGGL(fvta::adadelta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, var,
                     grad, accum, accum_update, n, lr, rho, eps, grad_scale);
  FVTA_CHECK_LAUNCH("adadelta");
  return FVTA_OK;
}

extern "C" int fvta_adam_step(float* var, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, int32_t t, float grad_scale, fvta_stream_t stream) {
  FVTA_CHECK_ARG(var && grad && m && v && n > 0 && t >= 1, "adam_step: bad arguments");
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
  hipLaunchKernelGGL(fvta::adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, var,
                     grad, m, v, n, (float)lr_t, beta1, beta2, eps, grad_scale);
  FVTA_CHECK_LAUNCH("adam");
  return FVTA_OK;
}
