// CU-masked streams on MI355X: (1) which (XCC, SE, CU) a stream created with hipExtStreamCreateWithCUMask runs on, for a few
// masks -- i.e. the bit layout of the mask; (2) whether a kernel that fills 208 CUs (one 512-thread, full-LDS workgroup each:
// the shape of lstm_bwd_fused_bf16's grid at the metric shape) keeps its time while a second kernel occupies the other 48
// CUs through a masked stream.     hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ long long wall() { return (long long)__builtin_readcyclecounter(); }
__device__ __forceinline__ long long wclk() { return wall_clock64(); }

// each workgroup: claims the whole LDS of its CU, records where it runs, then waits `ticks` (100 MHz)
__global__ void occupy(long long ticks, unsigned* where) {
  extern __shared__ char lds[];
  lds[threadIdx.x] = 1;
  if (threadIdx.x == 0) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);   // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);    // HW_REG_XCC_ID
    where[blockIdx.x] = (xcc << 16) | (hwid & 0xffff);
  }
  const long long t0 = wclk();
  while (wclk() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

static void layout(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask: %s\n", name, hipGetErrorString(e)); return; }
  unsigned* where;
  const int n = 1024;
  CK(hipMalloc(&where, n * 4));
  CK(hipMemset(where, 0xff, n * 4));
  hipLaunchKernelGGL(occupy, dim3(n), dim3(64), 150 * 1024, s, 2000ll, where);
  CK(hipStreamSynchronize(s));
  std::vector<unsigned> h(n);
  CK(hipMemcpy(h.data(), where, n * 4, hipMemcpyDeviceToHost));
  int cnt[8][8][16] = {};
  for (unsigned v : h) cnt[(v >> 16) & 7][(v >> 13) & 7][(v >> 8) & 15]++;
  printf("mask %s:", name);
  int cus = 0;
  for (int x = 0; x < 8; ++x) {
    int cx = 0;
    for (int se = 0; se < 8; ++se) for (int cu = 0; cu < 16; ++cu) if (cnt[x][se][cu]) ++cx;
    printf(" xcc%d:%d(", x, cx);
    for (int se = 0; se < 8; ++se) { int c = 0; for (int cu = 0; cu < 16; ++cu) if (cnt[x][se][cu]) ++c; if (c) printf("se%d:%d ", se, c); }
    printf(")");
    cus += cx;
  }
  printf("  -> %d distinct CUs\n", cus);
  CK(hipFree(where));
  CK(hipStreamDestroy(s));
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("%s CUs %d\n", p.gcnArchName, p.multiProcessorCount);
  CK(hipFuncSetAttribute((const void*)occupy, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  const int words = (p.multiProcessorCount + 31) / 32;
  std::vector<uint32_t> m(words, 0);
  auto setbits = [&](int lo, int hi, int stride = 1) { for (int i = lo; i < hi; i += stride) m[i / 32] |= 1u << (i % 32); };
  std::fill(m.begin(), m.end(), 0xffffffffu); layout("all", m);
  std::fill(m.begin(), m.end(), 0u); setbits(0, 48); layout("bits 0..47", m);
  std::fill(m.begin(), m.end(), 0u); setbits(0, 32); layout("bits 0..31", m);
  std::fill(m.begin(), m.end(), 0u); setbits(0, 256, 8); layout("every 8th", m);
  std::fill(m.begin(), m.end(), 0u); setbits(208, 256); layout("bits 208..255", m);
  std::fill(m.begin(), m.end(), 0u); setbits(0, 8); layout("bits 0..7", m);

  // ---- (2) 208 full-CU workgroups for 100 us, alone and beside masked ones: how often does a launch take two rounds?
  struct V { const char* name; int lo, hi; } vs[] = {{"alone", 0, 0}, {"bits 0..31", 0, 32}, {"bits 0..39", 0, 40}, {"bits 0..47", 0, 48}, {"bits 224..255", 224, 256}};
  for (const V& v : vs) {
    std::fill(m.begin(), m.end(), 0u);
    setbits(v.lo, v.hi);
    hipStream_t side = nullptr, mainS;
    CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
    if (v.hi) CK(hipExtStreamCreateWithCUMask(&side, (uint32_t)m.size(), m.data()));
    unsigned *w1, *w2;
    CK(hipMalloc(&w1, 4096 * 4));
    CK(hipMalloc(&w2, 4096 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int slow = 0;
    float worst = 0, best = 1e9;
    for (int rep = 0; rep < 30; ++rep) {
      if (side) hipLaunchKernelGGL(occupy, dim3(v.hi - v.lo), dim3(512), 150 * 1024, side, 200000ll, w2);  // 2 ms on the masked CUs
      hipLaunchKernelGGL(occupy, dim3(1), dim3(64), 1024, mainS, 20000ll, w1);
      CK(hipEventRecord(e0, mainS));
      for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(occupy, dim3(208), dim3(512), 150 * 1024, mainS, 10000ll, w1);  // 100 us each
      CK(hipEventRecord(e1, mainS));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms > 1.5f) ++slow;
      worst = ms > worst ? ms : worst;
      best = ms < best ? ms : best;
    }
    printf("%-14s 10 x (208 workgroups, 100 us): best %.3f worst %.3f ms, %d of 30 reps above 1.5 ms\n", v.name, best, worst, slow);
  }
  return 0;
}
