// Shared by the focal-attention forward kernels (attn_fwd.hip, attn_fwd_wide.hip): the launch arguments, the fp16 split of
// a row value (hi = rtz_f16(x), lo = f16((x - hi) * 2^11)), the LDS-only barrier and the fused DPP row sum.
#pragma once
#include "attn_common.h"

namespace fvta {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
// x[0..7] = (a0, a1) -> hi, lo fp16 pieces (see above); register pairs are concatenated, never re-packed
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_f16x2(float x0, float x1, half2v& hi, half2v& lo) {
  hi = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  // lo = fp16((x - hi) * 2048) as ONE mixed-precision FMA per element, 2048 x - 2048 hi with the fp16 hi read in place
  // (v_fma_mixlo/mixhi_f16 write one half of the destination and keep the other): 4 instructions per pair instead of 6
  // (2 back-conversions, packed subtract, packed scale, pack).  Exact up to the final rounding, as before (x - hi is exact).
  const f32x2 xs = f32x2{x0, x1} * 2048.f;
  const float m2048 = -2048.f;
  unsigned hw = __builtin_bit_cast(unsigned, hi), lw;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=&v"(lw) : "v"(hw), "s"(m2048), "v"(xs[0]));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lw) : "v"(hw), "s"(m2048), "v"(xs[1]));
  lo = __builtin_bit_cast(half2v, lw);
}
__device__ __forceinline__ half8 cat_h2(half2v a, half2v b, half2v c, half2v d) {
  const half4v ab = __builtin_shufflevector(a, b, 0, 1, 2, 3), cd = __builtin_shufflevector(c, d, 0, 1, 2, 3);
  return __builtin_shufflevector(ab, cd, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void split_f16x8(const f32x4 a0, const f32x4 a1, half8& hi, half8& lo) {
  half2v h[4], l[4];
  split_f16x2(a0[0], a0[1], h[0], l[0]);
  split_f16x2(a0[2], a0[3], h[1], l[1]);
  split_f16x2(a1[0], a1[1], h[2], l[2]);
  split_f16x2(a1[2], a1[3], h[3], l[3]);
  hi = cat_h2(h[0], h[1], h[2], h[3]);
  lo = cat_h2(l[0], l[1], l[2], l[3]);
}


__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Sum of each of four values over the 16 lanes of its DPP row, every lane ending with the result: the butterfly of
// row16_sum as FUSED v_add_f32_dpp (the compiler emits v_mov_b32_dpp + a packed add + s_nop per step: 31 issue slots for
// what are 17 here).  The four values are interleaved, so a step's DPP read of a register comes three instructions after
// the previous step wrote it (the hazard wants two wait states; the assembler does not check inside inline asm, hence
// also the leading s_nop against whatever VALU instruction produced the inputs).
__device__ __forceinline__ void row16_sum4(f32x4& v) {
  float a = v[0], b = v[1], c = v[2], d = v[3];
#define FVTA_DPP4(CTRL)                                                   \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"      \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"      \
  "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"      \
  "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
  asm("s_nop 1\n\t" FVTA_DPP4("quad_perm:[1,0,3,2]") FVTA_DPP4("quad_perm:[2,3,0,1]") FVTA_DPP4("row_half_mirror")
          FVTA_DPP4("row_mirror")
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef FVTA_DPP4
  v = f32x4{a, b, c, d};
}


// ---- launch arguments of the forward main kernels
struct AttnFwdArgs {
  AttnShape s;
  AttnSaved sv;
  const float* hinfo;
  float* a_logits;  // may be null
  float* part;      // [N*K][nsplit][w+4] : m, l, mu, -, u[w]   (m: max of the softmax logits z, mu: max of amax)
  const float* tscale;  // [N,T] or null: z[n,k,t] = amax[n,k,t] * tscale[n,t] (time_warp_att)
  int ipw;          // 16-row kernel: items per workgroup
  int dbg;          // FVTA_ATTN_DBG experiment bits (diagnostics only)
  size_t hstride;   // elements between the row blocks of consecutive (n,k): T*w, or fvta_attn_desc.hinfo_stride (K == 1)
  const uint32_t* wgtab;  // pair kernel: workgroup -> n | g << 16 | G_n << 24 (attn_balance_kernel), null: G workgroups for every n
  int* fault;             // host-mapped status word (attn_fault_word): a kernel that has to trap says why first
};

// Fault codes a forward kernel leaves in AttnFwdArgs::fault before it traps: the trap aborts the queue -- a process that
// survives it (or the next call, whichever comes first) reads the word and reports instead of "unspecified launch failure".
enum { ATTN_FAULT_PAIR_WAIT = 1 };
int* attn_fault_word();   // attn_fwd.hip: the word (device-visible address of host memory), null if it cannot be had

}  // namespace fvta
