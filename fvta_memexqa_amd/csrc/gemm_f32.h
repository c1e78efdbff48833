// fp32 tile engine on v_mfma_f32_32x32x2_f32 (exact fp32: each MFMA is a
// k-ordered fmaf chain, MI355X guide "FP32-input MFMA").
//
// Block tile BM x BN = (WAVES_M*TM*32) x (WAVES_N*TN*32), BK = 16, one wave per
// SIMD (256 threads for 4 waves).  LDS images are K-MAJOR: As[k][m], Bs[k][n]
// with a 4-float row pad, so an MFMA operand (lane l: A[i=l&31][k=l>>5],
// B[k=l>>5][j=l&31]) is one conflict-free ds_read_b32 per lane.  Sources whose
// rows are k-contiguous are transposed on the way into LDS; sources whose rows
// are m/n-contiguous are stored as they are.
#pragma once
#include "fvta_common.h"

namespace fvta {

template <int WAVES_M_, int WAVES_N_, int TM_, int TN_>
struct MmaF32 {
  static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, TM = TM_, TN = TN_;
  static constexpr int BM = WAVES_M * TM * 32;
  static constexpr int BN = WAVES_N * TN * 32;
  static constexpr int BK = 16;
  static constexpr int NT = WAVES_M * WAVES_N * 64;
  static constexpr int LDA = BM + 4;
  static constexpr int LDB = BN + 4;
  static constexpr int A_FLOATS = BK * LDA;
  static constexpr int B_FLOATS = BK * LDB;
  static constexpr int LDS_FLOATS = 2 * (A_FLOATS + B_FLOATS);

  f32x16 acc[TM][TN];
  int wm, wn, l31, hf;

  __device__ __forceinline__ void init(int tid) {
    const int wave = tid >> 6, lane = tid & 63;
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

  // one BK-deep tile from LDS images As[BK][LDA], Bs[BK][LDB].  (LDS-typed pointers: picked out of an array of generic
  // `float*` by the buffer index, the images were read with FLAT loads -- through the address-space check, and counted on
  // vmcnt as well as lgkmcnt)
  typedef float __attribute__((address_space(3))) lds_f;
  __device__ __forceinline__ void compute(const lds_f* __restrict__ As, const lds_f* __restrict__ Bs) {
    const lds_f* ap = As + hf * LDA + wm * (TM * 32) + l31;
    const lds_f* bp = Bs + hf * LDB + wn * (TN * 32) + l31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = ap[kk * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bp[kk * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // C/D map (dtype independent on gfx950): reg r of lane -> row, col inside a 32x32 tile
  __device__ __forceinline__ int row_of(int i, int r) const {
    return wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
  }
  __device__ __forceinline__ int col_of(int j) const { return wn * (TN * 32) + j * 32 + l31; }
};

// Fetch protocol (both stage kinds): f(i, j, ok) returns an ALWAYS DEREFERENCEABLE pointer to
// 4 floats and sets ok; the stage loads unconditionally and zeroes the value afterwards.  A load
// under a data-dependent branch makes hipcc wait vmcnt(0) per load (serialised L2 round trips).
//
// Register-staged tile whose SOURCE rows are k-contiguous: f(r, k, ok) -> &src[r][k].  Stored transposed.
template <int ROWS, int BK, int NT, int LD>
struct StageKContig {
  static constexpr int UNITS = ROWS * BK / 4;
  static constexpr int PER = (UNITS + NT - 1) / NT;
  f32x4 v[PER];
  template <class F>
  __device__ __forceinline__ void fetch(F&& f, int k0, int tid) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int u = tid + p * NT;
      if (UNITS % NT == 0 || u < UNITS) {
        bool ok;
        const float* src = f(u / (BK / 4), k0 + (u % (BK / 4)) * 4, ok);
        // (every staging source is global memory; said explicitly: the address-provider lambdas return generic pointers, and
        //  through those the loads were FLAT instructions -- which count on lgkmcnt too, so every wait for an LDS operand of
        //  the k-loop also waited for the next tile's global loads)
        const f32x4 t = *(const f32x4 __attribute__((address_space(1)))*)(uintptr_t)src;
        v[p] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  __device__ __forceinline__ void store(float __attribute__((address_space(3)))* lds, int tid) const {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int u = tid + p * NT;
      if (UNITS % NT == 0 || u < UNITS) {
        const int r = u / (BK / 4), kc = (u % (BK / 4)) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds[(kc + i) * LD + r] = v[p][i];
      }
    }
  }
};

// Register-staged tile whose SOURCE rows are m/n-contiguous: f(k, c, ok) -> &src[k][c].  Stored as is.
template <int COLS, int BK, int NT, int LD>
struct StageMNContig {
  static constexpr int UNITS = COLS * BK / 4;
  static constexpr int PER = (UNITS + NT - 1) / NT;
  f32x4 v[PER];
  template <class F>
  __device__ __forceinline__ void fetch(F&& f, int k0, int tid) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int u = tid + p * NT;
      if (UNITS % NT == 0 || u < UNITS) {
        bool ok;
        const float* src = f(k0 + u / (COLS / 4), (u % (COLS / 4)) * 4, ok);
        // (every staging source is global memory; said explicitly: the address-provider lambdas return generic pointers, and
        //  through those the loads were FLAT instructions -- which count on lgkmcnt too, so every wait for an LDS operand of
        //  the k-loop also waited for the next tile's global loads)
        const f32x4 t = *(const f32x4 __attribute__((address_space(1)))*)(uintptr_t)src;
        v[p] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  __device__ __forceinline__ void store(float __attribute__((address_space(3)))* lds, int tid) const {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int u = tid + p * NT;
      if (UNITS % NT == 0 || u < UNITS)
        *reinterpret_cast<f32x4 __attribute__((address_space(3)))*>(&lds[(u / (COLS / 4)) * LD + (u % (COLS / 4)) * 4]) = v[p];
    }
  }
};

// Double-buffered main loop: one barrier per k-tile; global loads of tile k+1
// are issued before the MFMAs of tile k and written to the other LDS buffer
// after them.
template <class Mma, class SA, class SB, class FA, class FB>
__device__ __forceinline__ void gemm_mainloop(Mma& mma, SA& sa, SB& sb, FA&& fa, FB&& fb, int k_begin,
                                              int k_end, float* smem, int tid) {
  typedef float __attribute__((address_space(3))) lds_f;
  lds_f* const base = (lds_f*)smem;  // (the two images of A, then the two of B: addressed by the buffer index, typed as LDS)
  auto As = [&](int b) { return base + b * Mma::A_FLOATS; };
  auto Bs = [&](int b) { return base + 2 * Mma::A_FLOATS + b * Mma::B_FLOATS; };
  if (k_begin >= k_end) return;
  sa.fetch(fa, k_begin, tid);
  sb.fetch(fb, k_begin, tid);
  sa.store(As(0), tid);
  sb.store(Bs(0), tid);
  __syncthreads();
  int cur = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += Mma::BK) {
    const bool more = (k0 + Mma::BK) < k_end;
    if (more) {
      sa.fetch(fa, k0 + Mma::BK, tid);
      sb.fetch(fb, k0 + Mma::BK, tid);
    }
    mma.compute(As(cur), Bs(cur));
    if (more) {
      sa.store(As(cur ^ 1), tid);
      sb.store(Bs(cur ^ 1), tid);
    }
    __syncthreads();
    cur ^= 1;
  }
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

}  // namespace fvta
