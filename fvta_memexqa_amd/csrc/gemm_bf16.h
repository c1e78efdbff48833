// bf16 tile engine on v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate).
//
// Block tile BM x BN, BK = 32 (two MFMA k-steps), one wave per SIMD.
//  * "row" images (NN / NT products): As[BM][LDK], Bs[BN][LDK] with k contiguous
//    (LDK = 40 elements = 80 B rows: conflict-free ds_read_b128 for the operand
//    map lane l -> row l&31, k = 8(l>>5)+j).
//  * "k-major" images (TN product, weight gradient): As[BK][LDM], Bs[BK][LDN]
//    exactly as the sources lie in memory (m / n contiguous), consumed through
//    ds_read_b64_tr_b16, the hardware transposing read; row stride BM+32
//    elements (= 16 banks mod 64) keeps the four k-rows of a read apart.
// C/D layout equals the fp32 engine's (dtype independent on gfx950).
#pragma once
#include "fvta_common.h"

namespace fvta {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

union Pack8 {
  bf16x8 s;  // 8 x 16-bit
  bf16x8_t b;
  f32x4 f;   // raw 16 bytes
};

__device__ __forceinline__ bf16x8 cvt8(const f32x4& lo, const f32x4& hi) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[i] = (short)f2bf(lo[i]);
    r[i + 4] = (short)f2bf(hi[i]);
  }
  return r;
}

template <int WAVES_M_, int WAVES_N_, int TM_, int TN_>
struct MmaBf16 {
  static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, TM = TM_, TN = TN_;
  static constexpr int BM = WAVES_M * TM * 32;
  static constexpr int BN = WAVES_N * TN * 32;
  static constexpr int BK = 32;
  static constexpr int NT = WAVES_M * WAVES_N * 64;
  static constexpr int LDK = BK + 8;   // row images
  static constexpr int LDM = BM + 32;  // k-major images
  static constexpr int LDN = BN + 32;
  static constexpr int A_ELEMS = (BM * LDK > BK * LDM) ? BM * LDK : BK * LDM;
  static constexpr int B_ELEMS = (BN * LDK > BK * LDN) ? BN * LDK : BK * LDN;
  static constexpr int LDS_BYTES = 2 * (A_ELEMS + B_ELEMS) * 2;

  f32x16 acc[TM][TN];
  int wm, wn, l31, hf, lane;

  __device__ __forceinline__ void init(int tid) {
    const int wave = tid >> 6;
    lane = tid & 63;
    wm = wave / WAVES_N;
    wn = wave % WAVES_N;
    l31 = lane & 31;
    hf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  // row images: As[BM][LDK], Bs[BN][LDK]
  __device__ __forceinline__ void compute_rows(const bf16_t* __restrict__ As, const bf16_t* __restrict__ Bs) {
    const bf16_t* ap = As + (wm * (TM * 32) + l31) * LDK + 8 * hf;
    const bf16_t* bp = Bs + (wn * (TN * 32) + l31) * LDK + 8 * hf;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      Pack8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i].f = *reinterpret_cast<const f32x4*>(ap + i * 32 * LDK + ks * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j].f = *reinterpret_cast<const f32x4*>(bp + j * 32 * LDK + ks * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].b, b[j].b, acc[i][j], 0, 0, 0);
    }
  }

  // one operand fragment from a k-major image via the transposing read:
  // group g = lane>>4 reads the 4(k) x 16(m) blocks at k0 = kbase + 8(g>>1) (+4), m0 = mbase + 16(g&1);
  // lane 4q+p of the group supplies row k0+q, columns m0+4p..+3 and receives column m0+(lane&15).
  __device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* img, int ld, int kbase, int mbase) const {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const bf16_t* p0 = img + (kbase + 8 * (g >> 1) + q) * ld + mbase + 16 * (g & 1) + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0 + 4 * ld));
    Pack8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r.s[e] = lo[e];
      r.s[e + 4] = hi[e];
    }
    return r.b;
  }

  // k-major images: As[BK][LDM] (m contiguous), Bs[BK][LDN] (n contiguous)
  __device__ __forceinline__ void compute_kmajor(const bf16_t* __restrict__ As, const bf16_t* __restrict__ Bs) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_t a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = tr_frag(As, LDM, ks * 16, wm * (TM * 32) + i * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = tr_frag(Bs, LDN, ks * 16, wn * (TN * 32) + j * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  __device__ __forceinline__ int row_of(int i, int r) const {
    return wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
  }
  __device__ __forceinline__ int col_of(int j) const { return wn * (TN * 32) + j * 32 + l31; }
};

// ---- register-staged tiles (8 bf16 = 16 B units) ----------------------------
// Row image from a source that yields 8 consecutive k of row r as two fp32 quads (converted here)
// or as 8 ready bf16.  fetch(r, k) -> bf16x8.
template <int ROWS, int BK, int NT, int LD>
struct StageRows {
  static constexpr int UNITS = ROWS * BK / 8;
  static constexpr int PER = (UNITS + NT - 1) / NT;
  bf16x8 v[PER];
  template <class F>
  __device__ __forceinline__ void fetch(F&& f, int k0, int tid) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int u = tid + p * NT;
      if (UNITS % NT == 0 || u < UNITS) v[p] = f(u / (BK / 8), k0 + (u % (BK / 8)) * 8);
    }
  }
  __device__ __forceinline__ void store(bf16_t* lds, int tid) const {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int u = tid + p * NT;
      if (UNITS % NT == 0 || u < UNITS)
        *reinterpret_cast<bf16x8*>(lds + (u / (BK / 8)) * LD + (u % (BK / 8)) * 8) = v[p];
    }
  }
};

// k-major image: fetch(k, c) -> 8 consecutive columns c..c+7 of source row k as bf16x8.
template <int COLS, int BK, int NT, int LD>
struct StageKMajor {
  static constexpr int UNITS = COLS * BK / 8;
  static constexpr int PER = (UNITS + NT - 1) / NT;
  bf16x8 v[PER];
  template <class F>
  __device__ __forceinline__ void fetch(F&& f, int k0, int tid) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int u = tid + p * NT;
      if (UNITS % NT == 0 || u < UNITS) v[p] = f(k0 + u / (COLS / 8), (u % (COLS / 8)) * 8);
    }
  }
  __device__ __forceinline__ void store(bf16_t* lds, int tid) const {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int u = tid + p * NT;
      if (UNITS % NT == 0 || u < UNITS)
        *reinterpret_cast<bf16x8*>(lds + (u / (COLS / 8)) * LD + (u % (COLS / 8)) * 8) = v[p];
    }
  }
};

// Double-buffered main loop (same schedule as the fp32 engine).  KMAJOR picks the image kind.
template <bool KMAJOR, class Mma, class SA, class SB, class FA, class FB>
__device__ __forceinline__ void gemm_mainloop_bf16(Mma& mma, SA& sa, SB& sb, FA&& fa, FB&& fb, int k_begin, int k_end,
                                                   bf16_t* smem, int tid) {
  bf16_t* As[2] = {smem, smem + Mma::A_ELEMS};
  bf16_t* Bs[2] = {smem + 2 * Mma::A_ELEMS, smem + 2 * Mma::A_ELEMS + Mma::B_ELEMS};
  if (k_begin >= k_end) return;
  sa.fetch(fa, k_begin, tid);
  sb.fetch(fb, k_begin, tid);
  sa.store(As[0], tid);
  sb.store(Bs[0], tid);
  __syncthreads();
  int cur = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += Mma::BK) {
    const bool more = (k0 + Mma::BK) < k_end;
    if (more) {
      sa.fetch(fa, k0 + Mma::BK, tid);
      sb.fetch(fb, k0 + Mma::BK, tid);
    }
    if (KMAJOR)
      mma.compute_kmajor(As[cur], Bs[cur]);
    else
      mma.compute_rows(As[cur], Bs[cur]);
    if (more) {
      sa.store(As[cur ^ 1], tid);
      sb.store(Bs[cur ^ 1], tid);
    }
    __syncthreads();
    cur ^= 1;
  }
}

__device__ __forceinline__ bf16x8 ld8h(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 zero8h() { return bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }

}  // namespace fvta
