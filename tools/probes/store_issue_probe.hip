// How long does a wave stand at a 16-byte-per-lane global store (and at other memory instructions) on gfx950?
// One workgroup per CU; `waves` waves per workgroup each issue `n` stores back to back (or separated by `gap` MFMAs) into
// (Round 4 rerun with 8 / 256 workgroups, 1 / 4 / 8 waves, 32 KB / 2 MB per wave: 116 cycles per buffer store and 128 per
//  LDS-DMA load PER WAVE whether 8 or 256 CUs run and whether the span sits in L2 -- an issue cost, not back-pressure; eight
//  waves reach the CU's 64 B/clk; the ADD_TID descriptor form (no address VGPR) is no faster alone and far slower loaded.)
// their own 1 KiB-per-instruction stream over a footprint of `span` bytes per wave; wave 0 of workgroup 0 prints
// shader cycles per store.  Build: hipcc --offload-arch=gfx950 -O3 -o store_issue_probe store_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

template <int MODE>  // 6: buffer loads to LDS through an ADD_TID descriptor (no address VGPR); 0: stores (64-bit per-lane addresses), 1: loads (results summed), 2: 8-byte stores, 3: buffer stores (32-bit offsets), 4: global stores, scalar base + 32-bit offset, 5: buffer loads to LDS
__global__ void probe(float* buf, size_t span_f, int n, int gap, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wid = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
  float* base = (float*)__builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)(buf + wid * span_f)));
  base = (float*)(((uintptr_t)__builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)(buf + wid * span_f) >> 32)) << 32) | (uintptr_t)(unsigned)(uintptr_t)base);
  __shared__ float lds[4 * 256];
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (unsigned)(span_f * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rtid = __builtin_amdgcn_make_buffer_rsrc(base, 16, (unsigned)(span_f * 4), 0x00820000);  // ADD_TID_ENABLE, stride 16
  const size_t per = MODE == 2 ? 128 : 256;  // floats per instruction
  f32x16 acc = {};
  bf16x8_t a = {}, b = {};
  f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  f32x4 sum = {};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    if (MODE == 0)
      *reinterpret_cast<f32x4*>(base + off + lane * 4) = v;
    else if (MODE == 3)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsrc, (unsigned)(off * 4 + lane * 16), 0, 0);
    else if (MODE == 4) {
      const unsigned o32 = (unsigned)(off * 4 + lane * 16);
      asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(o32), "v"(v), "s"(base) : "memory");
    } else if (MODE == 5)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, (unsigned)(off * 4 + lane * 16), 0, 0, 0);
    else if (MODE == 6)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rtid, (__attribute__((address_space(3))) void*)lds, 16, 0, (unsigned)(off * 4), 0, 0);
    else if (MODE == 2) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<f32x2*>(base + off + lane * 2) = f32x2{v[0], v[3]};
    } else
      sum += *reinterpret_cast<const f32x4*>(base + off + lane * 4);
    off += per;
    if (off + per > span_f) off = 0;
    for (int g = 0; g < gap; ++g) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_readcyclecounter();
  if (lane == 0 && wave == 0) {
    out[blockIdx.x * 4 + 0] = t1 - t0;
    out[blockIdx.x * 4 + 1] = t2 - t0;
  }
  if (sum[0] + acc[0] == 12345.f) buf[0] = sum[1];
}

int main(int argc, char** argv) {
  const int n = 2000;
  float* buf;
  unsigned long long* out;
  const size_t total = (size_t)3 << 30;
  hipMalloc(&buf, total);
  hipMalloc(&out, 256 * 4 * 8);
  hipMemset(buf, 0, total);
  std::vector<unsigned long long> h(1024);
  for (int mode : {3, 5, 6})
    for (int wgs : {8, 256})
      for (int waves : {1, 4, 8})
        for (size_t span_kb : {(size_t)32, (size_t)2048})
          for (int gap : {0, 4}) {
            const size_t span_f = span_kb * 256;
            if ((size_t)wgs * waves * span_f * 4 > total) continue;
            for (int rep = 0; rep < 2; ++rep) {
              if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(wgs), dim3(64 * waves), 0, 0, buf, span_f, n, gap, out);
              if (mode == 5) hipLaunchKernelGGL(probe<5>, dim3(wgs), dim3(64 * waves), 0, 0, buf, span_f, n, gap, out);
              if (mode == 6) hipLaunchKernelGGL(probe<6>, dim3(wgs), dim3(64 * waves), 0, 0, buf, span_f, n, gap, out);
              hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), out, 256 * 4 * 8, hipMemcpyDeviceToHost);
            printf("%-22s wgs %3d waves/CU %d span/wave %5zu KB gap %d MFMA: %6.1f cycles per instr issued (%.1f incl. drain); MFMA-only would be %d\n",
                   mode == 3 ? "buffer_store16" : mode == 5 ? "buffer_load_lds16" : "buffer_load_lds16 add_tid", wgs, waves, span_kb, gap, (double)h[0] / n, (double)h[1] / n, gap * 32);
          }
  return 0;
}
