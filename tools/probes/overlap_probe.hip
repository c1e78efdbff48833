// What tile period does a workgroup of 8 waves reach when every wave loads its slice of a 16-row x 4 KiB tile in the
// attention kernel's lane order (8 x buffer_load_dwordx4 per wave and tile, three register buffers) and then spends
// C cycles of dependent VALU work on it -- as a function of WHERE the loads are issued and HOW OFTEN the waves meet at a
// barrier?  (diagnostics for DESIGN.md 4.1; not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NBAR barriers per tile; SPREAD: 0 = the 8 loads of the tile two ahead in one burst at the tile start, 1 = one load per
// eighth of the compute; work = dependent FMA chain length per eighth (4 independent chains)
template <int NBAR, int SPREAD>
__global__ __launch_bounds__(512, 1) void mimic(const float* __restrict__ p, int W, int tiles_total, int work, float* out) {
  constexpr int NB = 8;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int per = (tiles_total + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * per, t1 = min(tiles_total, t0 + per);
  f32x4 bufA[NB], bufB[NB], bufC[NB];
  f32x4 acc = {0.f, 1.f, 2.f, 3.f};
  auto addr = [&](int t) { return p + ((size_t)t * 16 + l15) * W + wave * NB * 16 + 4 * kq; };
  auto load_all = [&](int t, f32x4(&b)[NB]) {
    const float* row = addr(t);
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i] = *reinterpret_cast<const f32x4*>(row + 16 * i);
  };
  auto chunk = [&](const f32x4& v) {  // `work` steps of 4 independent dependent-FMA chains seeded by the tile's data
    f32x4 a = acc + v;
    for (int i = 0; i < work; ++i) a = a * 1.0001f + 0.5f;
    acc = a;
  };
  auto tile = [&](int t, f32x4(&cur)[NB], f32x4(&nxt)[NB]) {
    const bool more = t + 2 < t1;
    if (!SPREAD && more) load_all(t + 2, nxt);
    const float* row = addr(more ? t + 2 : t);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      chunk(cur[i]);
      if (SPREAD && more) {
        __builtin_amdgcn_sched_barrier(0);
        nxt[i] = *reinterpret_cast<const f32x4*>(row + 16 * i);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (NBAR >= 2 && i == 3) lds_barrier();
    }
    if (NBAR >= 1) lds_barrier();
  };
  if (t0 < t1) load_all(t0, bufA);
  if (t0 + 1 < t1) load_all(t0 + 1, bufB);
  for (int t = t0; t < t1; t += 3) {
    tile(t, bufA, bufC);
    if (t + 1 < t1) tile(t + 1, bufB, bufA);
    if (t + 2 < t1) tile(t + 2, bufC, bufB);
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// ---- the same loop with loads the compiler does not track (inline asm, pinned registers, counted waits) ----------
template <int BUF>
__device__ __forceinline__ void issue8(const float* base, unsigned voff, f32x4 (&d)[8]) {
  if constexpr (BUF == 0) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:0" : "={v[160:163]}"(d[0]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:64" : "={v[164:167]}"(d[1]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:128" : "={v[168:171]}"(d[2]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:192" : "={v[172:175]}"(d[3]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:256" : "={v[176:179]}"(d[4]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:320" : "={v[180:183]}"(d[5]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:384" : "={v[184:187]}"(d[6]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:448" : "={v[188:191]}"(d[7]) : "v"(voff), "s"(base));
  } else if constexpr (BUF == 1) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:0" : "={v[192:195]}"(d[0]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:64" : "={v[196:199]}"(d[1]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:128" : "={v[200:203]}"(d[2]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:192" : "={v[204:207]}"(d[3]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:256" : "={v[208:211]}"(d[4]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:320" : "={v[212:215]}"(d[5]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:384" : "={v[216:219]}"(d[6]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:448" : "={v[220:223]}"(d[7]) : "v"(voff), "s"(base));
  } else {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:0" : "={v[224:227]}"(d[0]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:64" : "={v[228:231]}"(d[1]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:128" : "={v[232:235]}"(d[2]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:192" : "={v[236:239]}"(d[3]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:256" : "={v[240:243]}"(d[4]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:320" : "={v[244:247]}"(d[5]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:384" : "={v[248:251]}"(d[6]) : "v"(voff), "s"(base));
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:448" : "={v[252:255]}"(d[7]) : "v"(voff), "s"(base));
  }
}
template <int BUF>
__device__ __forceinline__ void wait8(int younger, f32x4 (&b)[8]) {
  if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (BUF == 0) {
    asm volatile("" : "+{v[160:163]}"(b[0]), "+{v[164:167]}"(b[1]), "+{v[168:171]}"(b[2]), "+{v[172:175]}"(b[3]), "+{v[176:179]}"(b[4]), "+{v[180:183]}"(b[5]), "+{v[184:187]}"(b[6]), "+{v[188:191]}"(b[7]));
  } else if constexpr (BUF == 1) {
    asm volatile("" : "+{v[192:195]}"(b[0]), "+{v[196:199]}"(b[1]), "+{v[200:203]}"(b[2]), "+{v[204:207]}"(b[3]), "+{v[208:211]}"(b[4]), "+{v[212:215]}"(b[5]), "+{v[216:219]}"(b[6]), "+{v[220:223]}"(b[7]));
  } else {
    asm volatile("" : "+{v[224:227]}"(b[0]), "+{v[228:231]}"(b[1]), "+{v[232:235]}"(b[2]), "+{v[236:239]}"(b[3]), "+{v[240:243]}"(b[4]), "+{v[244:247]}"(b[5]), "+{v[248:251]}"(b[6]), "+{v[252:255]}"(b[7]));
  }
}
template <int I> struct Tag { static constexpr int value = I; };

template <int NBAR>
__global__ __launch_bounds__(512, 1) void mimic_untracked(const float* __restrict__ p, int W, int tiles_total, int work, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int per = (tiles_total + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * per, t1 = min(tiles_total, t0 + per);
  f32x4 bufA[8], bufB[8], bufC[8];
  f32x4 acc = {0.f, 1.f, 2.f, 3.f};
  const unsigned voff0 = (unsigned)((l15 * W + wave * 128 + 4 * kq) * 4);
  auto chunk = [&](const f32x4& v) {
    f32x4 a = acc + v;
    for (int i = 0; i < work; ++i) a = a * 1.0001f + 0.5f;
    acc = a;
  };
  auto tile = [&](int t, f32x4(&cur)[8], f32x4(&nxt)[8], auto cb, auto nb) {
    const bool more = t + 2 < t1;
    if (more) issue8<decltype(nb)::value>(p + (size_t)(t + 2) * 16 * W, voff0, nxt);
    wait8<decltype(cb)::value>(more ? 2 : (t + 1 < t1 ? 1 : 0), cur);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      chunk(cur[i]);
      if (NBAR >= 2 && i == 3) lds_barrier();
    }
    if (NBAR >= 1) lds_barrier();
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (t0 < t1) issue8<0>(p + (size_t)t0 * 16 * W, voff0, bufA);
  if (t0 + 1 < t1) issue8<1>(p + (size_t)(t0 + 1) * 16 * W, voff0, bufB);
  for (int t = t0; t < t1; t += 3) {
    tile(t, bufA, bufC, Tag<0>{}, Tag<2>{});
    if (t + 1 < t1) tile(t + 1, bufB, bufA, Tag<1>{}, Tag<0>{});
    if (t + 2 < t1) tile(t + 2, bufC, bufB, Tag<2>{}, Tag<1>{});
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <class F>
static float time_ms(F f, int n = 3) {
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
  const size_t bytes = (size_t)1536 << 20;
  float *p, *out;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(p, 0, bytes));
  const int W = 1024, tiles = (int)(bytes / 4 / W / 16);
  printf("tile = 64 KiB; columns: work per eighth -> TB/s (and shader cycles per tile at 2.1 GHz, 96 tiles per CU)\n");
  for (int work : {0, 10, 20, 30, 40, 80}) {
    float a = time_ms([&] { hipLaunchKernelGGL((mimic<0, 0>), dim3(256), dim3(512), 0, 0, p, W, tiles, work, out); });
    float b = time_ms([&] { hipLaunchKernelGGL((mimic<0, 1>), dim3(256), dim3(512), 0, 0, p, W, tiles, work, out); });
    float c = time_ms([&] { hipLaunchKernelGGL((mimic<1, 0>), dim3(256), dim3(512), 0, 0, p, W, tiles, work, out); });
    float d = time_ms([&] { hipLaunchKernelGGL((mimic<1, 1>), dim3(256), dim3(512), 0, 0, p, W, tiles, work, out); });
    float e = time_ms([&] { hipLaunchKernelGGL((mimic<2, 0>), dim3(256), dim3(512), 0, 0, p, W, tiles, work, out); });
    float f = time_ms([&] { hipLaunchKernelGGL((mimic<2, 1>), dim3(256), dim3(512), 0, 0, p, W, tiles, work, out); });
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.1e9 / (tiles / 256.0); };
    float u0 = time_ms([&] { hipLaunchKernelGGL((mimic_untracked<0>), dim3(256), dim3(512), 0, 0, p, W, tiles, work, out); });
    float u2 = time_ms([&] { hipLaunchKernelGGL((mimic_untracked<2>), dim3(256), dim3(512), 0, 0, p, W, tiles, work, out); });
    printf("work %3d | UNTRACKED loads: no barrier %.2f (%5.0f) | 2 barriers %.2f (%5.0f)\n", work, bytes / u0 / 1e9, cyc(u0), bytes / u2 / 1e9, cyc(u2));
    printf("work %3d | no barrier: burst %.2f (%5.0f) spread %.2f (%5.0f) | 1 barrier: burst %.2f (%5.0f) spread %.2f (%5.0f) | 2 barriers: burst %.2f (%5.0f) spread %.2f (%5.0f)\n",
           work, bytes / a / 1e9, cyc(a), bytes / b / 1e9, cyc(b), bytes / c / 1e9, cyc(c), bytes / d / 1e9, cyc(d), bytes / e / 1e9, cyc(e),
           bytes / f / 1e9, cyc(f));
  }
  return 0;
}
