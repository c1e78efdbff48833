// What does the LAYOUT of a row-major bf16 operand cost a tiled k-loop on MI355X?  The backward step and dx read dz
// [rows][2048] (4-KB rows) as 256-row x 64-column k-tiles: 256 pieces of 128 B, 4 KB apart, per k-tile.  This probe streams
// the same 1.6 GB through LDS-DMA with the k-loops' structure (one 512-thread workgroup per CU, `depth` k-tiles in flight,
// no compute) in three address patterns:
//   mode 0  row-major rows, 128 B of each of 256 rows per k-tile (what the kernels do)
//   mode 1  tile-blocked: the k-tile's 256 x 128 B are one contiguous 32 KB
//   mode 2  row-major rows, 256 B of each row per k-tile (128-deep stages)
// Measured (MI355X): row-major 4.8-4.9 TB/s, with each workgroup starting at its own k-tile (rotation) 6.2-6.4 TB/s = the
// tile-blocked layout's rate: the layout costs nothing once workgroups do not walk the same 128-byte column together.  One
// CU draws 55 GB/s of this HBM-sourced stream when few CUs run (16 workgroups), 25 GB/s when all 256 do (the chip's ~6.4
// TB/s); an equal L2-resident second operand (dx's weights) rides along for free here (0.263 vs 0.255 ms; alone 0.054 ms).
// Build: hipcc --offload-arch=gfx950 -O3 -o rowslab_probe rowslab_probe.hip ; run: ./rowslab_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MODE, int DEPTH, int WB = 0>
__device__ __forceinline__ void slab_body(const char* buf, int ntiles, int rot_on, const char* wbuf = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ROWB = 4096;                      // bytes per row
  constexpr int PIECE = MODE == 2 ? 256 : 128;    // bytes of a row per k-tile
  constexpr int NKT = ROWB / PIECE;
  constexpr int PER = 256 * PIECE / 1024 / 8;     // wave-instructions per wave and k-tile (1 KB each)
  constexpr int STAGE = 256 * PIECE;              // bytes
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const char* base = buf + (size_t)tile * 256 * ROWB;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 256u * ROWB, 0x00020000);
    unsigned voff[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int U = (wave * PER + j) * 64 + lane;          // 16-byte unit of the stage
      const int row = U / (PIECE / 16), c = U % (PIECE / 16);
      voff[j] = MODE == 1 ? (unsigned)U * 16u : (unsigned)row * ROWB + 16u * c;
    }
    const int rot = rot_on ? (blockIdx.x >> 3) % NKT : 0;
    // WB: a second operand of the same size per k-tile from a 1 MB (L2-resident) buffer -- dx's weights
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbuf), 0, WB ? 256u * ROWB : 0u, 0x00020000);
    auto issue = [&](int t) {
      int kt = t + rot;
      kt -= kt >= NKT ? NKT : 0;
      const unsigned soff = MODE == 1 ? (unsigned)kt * STAGE : (unsigned)kt * PIECE;
      char* st = smem + (t % DEPTH) * STAGE * (WB ? 2 : 1);
      char* sw = st + STAGE;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        if (WB != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(st + (wave * PER + j) * 1024), 16, voff[j], soff, 0, 0);
        if (WB == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void_ptr)(sw + (wave * PER + j) * 1024), 16, voff[j], soff, 0, 0);
      }
      if (WB == 1 || WB == 3) {
#pragma unroll
        for (int j = 0; j < PER; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void_ptr)(sw + (wave * PER + j) * 1024), 16, voff[j], soff, 0, 0);
      }
    };
#pragma unroll
    for (int p = 0; p < DEPTH - 1; ++p) issue(p);
    for (int t = 0; t < NKT; ++t) {
      if (t + DEPTH - 1 < NKT) {
        issue(t + DEPTH - 1);
        wait_vmcnt<(DEPTH - 1) * PER * ((WB == 1 || WB == 2) ? 2 : 1)>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
    }
  }
}

#define DEF(M, D) __global__ __launch_bounds__(512, 1) void k_##M##_##D(const char* b, int n, int r) { slab_body<M, D>(b, n, r); }
DEF(0, 2) DEF(0, 4) DEF(1, 2) DEF(1, 4) DEF(2, 2)
#define DEFW(D, W) __global__ __launch_bounds__(512, 1) void kw_##D##_##W(const char* b, int n, int r, const char* w) { slab_body<0, D, W>(b, n, r, w); }
DEFW(2, 1) DEFW(2, 2) DEFW(2, 3)
typedef void (*kernw_t)(const char*, int, int, const char*);
static void runw(kernw_t k, int DEPTH, const char* buf, const char* wbuf, int ntiles, const char* what) {
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int lds_req = DEPTH * 2 * 256 * 128 > 96 * 1024 ? DEPTH * 2 * 256 * 128 : 96 * 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(ntiles), dim3(512), lds_req, 0, buf, ntiles, 1, wbuf);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("%-70s depth %d: %.3f ms\n", what, DEPTH, best);
}

typedef void (*kern_t)(const char*, int, int);
static void run(kern_t k, int MODE, int DEPTH, const char* buf, int ntiles, int rot, const char* what, int grid = 0) {
  if (!grid) grid = ntiles;
  const int lds = DEPTH * 256 * (MODE == 2 ? 256 : 128);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int lds_req = lds > 96 * 1024 ? lds : 96 * 1024;   // one workgroup per CU, as the kernels
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds_req, 0, buf, ntiles, rot);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("%-58s depth %d rot %d grid %4d: %.3f ms  %.2f TB/s  %.1f GB/s per workgroup\n", what, DEPTH, rot, grid, best,
         (double)ntiles * 256 * 4096 / best / 1e9, (double)ntiles * 256 * 4096 / best / 1e6 / (grid < 256 ? grid : 256));
}

int main() {
  const int ntiles = 1507;
  char* buf;
  hipMalloc(&buf, (size_t)ntiles * 256 * 4096);
  hipMemset(buf, 1, (size_t)ntiles * 256 * 4096);
  run(k_0_2, 0, 2, buf, ntiles, 0, "row-major, 128 B x 256 rows per k-tile");
  run(k_0_2, 0, 2, buf, ntiles, 1, "row-major, 128 B x 256 rows per k-tile");
  run(k_0_4, 0, 4, buf, ntiles, 0, "row-major, 128 B x 256 rows per k-tile");
  run(k_0_4, 0, 4, buf, ntiles, 1, "row-major, 128 B x 256 rows per k-tile");
  run(k_1_2, 1, 2, buf, ntiles, 0, "tile-blocked, 32 KB contiguous per k-tile");
  run(k_1_4, 1, 4, buf, ntiles, 0, "tile-blocked, 32 KB contiguous per k-tile");
  run(k_2_2, 2, 2, buf, ntiles, 0, "row-major, 256 B x 256 rows per k-tile");
  run(k_2_2, 2, 2, buf, ntiles, 1, "row-major, 256 B x 256 rows per k-tile");
  char* wbuf;
  hipMalloc(&wbuf, 256 * 4096);
  hipMemset(wbuf, 1, 256 * 4096);
  // dx's operand streams: A = the rotated row-major stream above (HBM), B = as many bytes per k-tile from 1 MB (L2)
  runw(kw_2_3, 2, buf, wbuf, ntiles, "B only (1 MB buffer, L2)");
  runw(kw_2_1, 2, buf, wbuf, ntiles, "A then B per k-tile");
  runw(kw_2_2, 2, buf, wbuf, ntiles, "A and B interleaved per instruction");
  // is the per-CU rate of an HBM-sourced stream a CU limit or the chip's?  Fewer workgroups (one per CU, persistent), same pattern
  for (int grid : {16, 32, 64, 128, 256}) run(k_0_4, 0, 4, buf, ntiles / 4, 1, "row-major, rotated, persistent workgroups", grid);
  for (int grid : {16, 64, 256}) run(k_1_4, 1, 4, buf, ntiles / 4, 0, "tile-blocked, persistent workgroups", grid);
  return 0;
}
