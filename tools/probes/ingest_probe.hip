// L2 -> LDS ingest rate of one CU by the SHAPE of a DMA wave-instruction (buffer_load_dwordx4 ... lds, 64 lanes x 16 B):
// `seg` consecutive lanes read seg x 16 contiguous bytes of one row, the next `seg` lanes the next row (row pitch 4 KiB,
// the dz row of the LSTM GEMMs).  seg = 4 is the tiled kernels' shape (a 32-wide k-tile of a bf16 row image: 64 B per
// row); 8 / 16 / 64 are what a 64- / 128-wide k-tile or a fully contiguous KiB would give.  One workgroup per CU, `waves`
// waves each issuing n instructions back to back over a source small enough to stay in L2; prints bytes per clock and CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o ingest_probe ingest_probe.hip ; run: ./ingest_probe [waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SEG>
__global__ void probe(const char* buf, unsigned rows_per_wg, int n, unsigned long long* out, int shared_window) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // this workgroup's window: rows_per_wg rows of 4 KiB
  const char* base = buf + (shared_window ? 0 : (size_t)blockIdx.x * rows_per_wg * 4096);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, rows_per_wg * 4096u, 0x00020000);
  constexpr int RPI = 64 / SEG;  // rows per instruction
  const unsigned row_l = lane / SEG, chunk = lane % SEG;
  char* dst = lds + wave * 4096;  // 4 instructions' worth per wave, rotating
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned row = wave * RPI, kofs = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned voff = (row + row_l) * 4096u + kofs + chunk * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (i & 3) * 1024), 16, voff, 0, 0, 0);
    row += nw * RPI;
    if (row + RPI > rows_per_wg) {
      row = wave * RPI;
      kofs = (kofs + SEG * 16u) & 4095u;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int SEG>
static void run(const char* buf, int waves, unsigned long long* d_out, int cus, int shared_window) {
  const int n = 4096;
  const unsigned rows_per_wg = 256;  // 1 MiB window per workgroup (x 256 workgroups = 256 MiB: L2 + Infinity Cache; the
                                     // k offset walks the window so that a pass re-reads lines its predecessors fetched)
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<SEG>, dim3(cus), dim3(64 * waves), 64 * 1024, 0, buf, rows_per_wg, n, d_out, shared_window);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(cus);
  hipMemcpy(h.data(), d_out, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : h) avg += (double)v;
  avg /= cus;
  printf("%s seg %2d lanes (%4d B per row): %8.0f cycles for %d instructions x %d waves -> %.1f cycles per instruction and CU, %.1f B/clk/CU\n",
         shared_window ? "one 1 MiB window (L2 hits)  " : "1 MiB per workgroup (misses)", SEG, SEG * 16, avg, n, waves, avg / ((double)n * waves), (double)n * waves * 1024.0 / avg);
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 8;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  char* buf;
  hipMalloc(&buf, (size_t)cus * 256 * 4096);
  hipMemset(buf, 1, (size_t)cus * 256 * 4096);
  unsigned long long* d_out;
  hipMalloc(&d_out, cus * sizeof(unsigned long long));
  hipFuncSetAttribute((const void*)probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipFuncSetAttribute((const void*)probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipFuncSetAttribute((const void*)probe<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipFuncSetAttribute((const void*)probe<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int sh = 1; sh >= 0; --sh) {
    run<4>(buf, waves, d_out, cus, sh);
    run<8>(buf, waves, d_out, cus, sh);
    run<16>(buf, waves, d_out, cus, sh);
    run<64>(buf, waves, d_out, cus, sh);
  }
  return 0;
}
