// HBM read-bandwidth probe (diagnostics, not part of the library): what does this machine deliver for
//  A: fully coalesced 16 B / lane streaming reads (1 KiB contiguous per wave-instruction), and
//  B: the attention kernel's MFMA-operand order (16 rows x 64 B per wave-instruction),
// as a function of waves per CU and loads in flight per wave.   hipcc --offload-arch=gfx950 -O3 stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void read_coalesced(const f32x4* __restrict__ p, size_t n16, float* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// rows of W floats; a workgroup of NWAVE waves takes 16-row tiles: wave w owns channels [w*W/NWAVE, ...), lane (row l&15, kq l>>4)
// loads 16 B at channel 16*blk + 4*kq of its row: NB = W/NWAVE/16 loads per tile per lane, DEPTH tiles in flight.
template <int NB, int DEPTH>
__global__ void read_rows16(const float* __restrict__ p, int W, int tiles_total, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int per = (tiles_total + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * per, t1 = min(tiles_total, t0 + per);
  f32x4 acc = {0, 0, 0, 0};
  f32x4 buf[DEPTH][NB];
  auto load = [&](int t, f32x4(&b)[NB]) {
    const float* row = p + ((size_t)t * 16 + l15) * W + wave * NB * 16 + 4 * kq;
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i] = *reinterpret_cast<const f32x4*>(row + 16 * i);
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if (t0 + d < t1) load(t0 + d, buf[d]);
  for (int t = t0; t < t1; t += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (t + d < t1) {
        if (t + d + DEPTH - 1 < t1) load(t + d + DEPTH - 1, buf[(d + DEPTH - 1) % DEPTH]);
#pragma unroll
        for (int i = 0; i < NB; ++i) acc += buf[d][i];
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class F>
static float time_ms(F f, int n = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < n; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / n;
}

int main() {
  const size_t bytes = (size_t)1600 << 20;  // ~ the metric shape's context tensor
  float *p, *out;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(p, 0, bytes));
  const size_t n16 = bytes / 16;
  printf("coalesced 16 B/lane, nontemporal\n");
  for (int wg_per_cu : {1, 2, 4, 8}) for (int threads : {256, 512, 1024}) {
    const int grid = 256 * wg_per_cu;
    float ms4 = time_ms([&] { hipLaunchKernelGGL(read_coalesced<4>, dim3(grid), dim3(threads), 0, 0, (const f32x4*)p, n16, out); });
    float ms8 = time_ms([&] { hipLaunchKernelGGL(read_coalesced<8>, dim3(grid), dim3(threads), 0, 0, (const f32x4*)p, n16, out); });
    printf("  wg/cu %d threads %4d : unroll4 %.3f ms %.2f TB/s | unroll8 %.3f ms %.2f TB/s\n", wg_per_cu, threads, ms4,
           bytes / ms4 / 1e9, ms8, bytes / ms8 / 1e9);
  }
  const int W = 1024, tiles = (int)(bytes / 4 / W / 16);
  printf("rows16 order (16 rows x 64 B per wave-instruction), w = 1024\n");
  for (int wg_per_cu : {1, 2, 4}) {
    const int grid = 256 * wg_per_cu;
    float a = time_ms([&] { hipLaunchKernelGGL((read_rows16<8, 2>), dim3(grid), dim3(512), 0, 0, p, W, tiles, out); });
    float b = time_ms([&] { hipLaunchKernelGGL((read_rows16<8, 3>), dim3(grid), dim3(512), 0, 0, p, W, tiles, out); });
    float c = time_ms([&] { hipLaunchKernelGGL((read_rows16<8, 4>), dim3(grid), dim3(512), 0, 0, p, W, tiles, out); });
    float d = time_ms([&] { hipLaunchKernelGGL((read_rows16<4, 4>), dim3(grid), dim3(1024), 0, 0, p, W, tiles, out); });
    printf("  wg/cu %d : 8 waves depth2 %.2f TB/s | depth3 %.2f | depth4 %.2f | 16 waves depth4 %.2f\n", wg_per_cu, bytes / a / 1e9,
           bytes / b / 1e9, bytes / c / 1e9, bytes / d / 1e9);
  }
  return 0;
}
