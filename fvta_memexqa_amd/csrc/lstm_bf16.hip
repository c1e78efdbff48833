// bi-LSTM step kernels on the bf16 MFMA engine (BASELINE.json configs[2]: bf16
// compute, fp32 accumulate).  Same decomposition and epilogues as lstm.hip; the
// operands are rounded to bf16 on their way into LDS, everything the cell state
// touches (gates, c, h, dz, accumulators, weight-gradient slabs) stays fp32
// except the stored gate-gradient rows dz, which are bf16 (they are only ever
// consumed as MFMA operands).
#include "gemm_bf16.h"
#include "gemm_f32.h"
#include "lstm_common.h"

namespace fvta {

// ---- weight shadows: kernel [K][N4] fp32 -> wb [K][N4] bf16 and wt [N4][Kp] bf16 (zero padded)
__global__ void cvt_weights_kernel(const float* __restrict__ W, bf16_t* __restrict__ wt, bf16_t* __restrict__ wb,
                                   int K, int Kp, int N4) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    const float v = (k < K && n < N4) ? W[(size_t)k * N4 + n] : 0.f;
    tile[r][tx] = v;
    if (k < K && n < N4) wb[(size_t)k * N4 + n] = f2bf(v);
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, k = k0 + tx;
    if (n < N4 && k < Kp) wt[(size_t)n * Kp + k] = f2bf(tile[tx][r]);
  }
}

void launch_cvt_weights_bf16(const float* W, bf16_t* wt, bf16_t* wb, int K, int Kp, int N4, hipStream_t s) {
  hipLaunchKernelGGL(cvt_weights_kernel, dim3((N4 + 31) / 32, (Kp + 31) / 32), dim3(256), 0, s, W, wt, wb, K, Kp, N4);
}

// ------------------------------------------------------------ forward step --
using MmaStepB = MmaBf16<4, 1, 1, 4>;

__global__ __launch_bounds__(256) void lstm_step_fwd_bf16(StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  __shared__ int64_t s_xo[MmaStepB::BM];
  __shared__ int64_t s_ho[MmaStepB::BM];
  const int tid = threadIdx.x;
  const int dir = blockIdx.z;
  const int m0 = blockIdx.x * MmaStepB::BM;
  const int nact = a.plan.nactive[a.t];
  if (m0 >= nact) return;
  const int u0 = blockIdx.y * 32;
  const int d = a.d, in = a.in, t = a.t;
  const int64_t out_ld = a.plan.hdr->out_ld;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  if (tid < MmaStepB::BM) {
    const int i = m0 + tid;
    int64_t xo = -1, oo = -1;
    if (i < nact) {
      xo = a.plan.xo[trow + i];
      oo = a.plan.oo[trow + i];
    }
    s_xo[tid] = xo;
    s_ho[tid] = (oo < 0 || t == 0) ? -1 : (dir ? oo + out_ld : oo - out_ld);
  }
  __syncthreads();
  const bf16_t* __restrict__ Wt = a.Wt[dir];
  const float* __restrict__ x = a.x;
  const float* __restrict__ hsrc = a.out;
  const int K = (t == 0) ? in : in + d;
  const int Kp = a.Kp;

  MmaStepB mma;
  mma.init(tid);
  StageRows<MmaStepB::BM, MmaStepB::BK, MmaStepB::NT, MmaStepB::LDK> sa;
  StageRows<MmaStepB::BN, MmaStepB::BK, MmaStepB::NT, MmaStepB::LDK> sb;
  auto quad = [&](int r, int k) -> f32x4 {  // 4 consecutive k of row r of [x | h_prev]; in % 4 == 0
    if (k >= K) return zero4();
    if (k < in) {
      const int64_t o = s_xo[r];
      return o < 0 ? zero4() : ld4(x + o + k);
    }
    const int64_t o = s_ho[r];
    return o < 0 ? zero4() : ld4(hsrc + o + (k - in));
  };
  auto fa = [&](int r, int k) -> bf16x8 { return cvt8(quad(r, k), quad(r, k + 4)); };
  auto fb = [&](int c, int k) -> bf16x8 {  // virtual column c -> gate strip row of kernel^T
    if (k >= K) return zero8h();
    const int g = c >> 5, u = c & 31;
    return ld8h(Wt + (size_t)(g * d + u0 + u) * Kp + k);
  };
  gemm_mainloop_bf16<false>(mma, sa, sb, fa, fb, 0, (K + MmaStepB::BK - 1) / MmaStepB::BK * MmaStepB::BK, smem_h, tid);
  lstm_gate_epilogue(mma, a, dir, m0, u0, nact, trow);
}

void launch_step_fwd_bf16(const StepArgs& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL(lstm_step_fwd_bf16, grid, dim3(256), MmaStepB::LDS_BYTES, s, a);
}

// ----------------------------------------------------------- backward step --
using MmaSqB = MmaBf16<2, 2, 2, 2>;

__global__ __launch_bounds__(256) void lstm_step_bwd_bf16(StepBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int tid = threadIdx.x, dir = blockIdx.z;
  const int m0 = blockIdx.x * MmaSqB::BM, n0 = blockIdx.y * MmaSqB::BN;
  const int nact = a.plan.nactive[a.t];
  if (m0 >= nact) return;
  const int d = a.d, in = a.in, t = a.t;
  const int NN = in + d, K = 4 * d;
  if (t == 0 && n0 >= in) return;
  if (a.dx == nullptr && n0 + MmaSqB::BN <= in) return;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  const bf16_t* __restrict__ dz = a.dzb + trow * (size_t)K;
  const bf16_t* __restrict__ Wb = a.Wb[dir];
  MmaSqB mma;
  mma.init(tid);
  StageRows<MmaSqB::BM, MmaSqB::BK, MmaSqB::NT, MmaSqB::LDK> sa;
  StageRows<MmaSqB::BN, MmaSqB::BK, MmaSqB::NT, MmaSqB::LDK> sb;
  auto fa = [&](int r, int k) -> bf16x8 {
    const int i = m0 + r;
    return i < nact ? ld8h(dz + (size_t)i * K + k) : zero8h();
  };
  auto fb = [&](int r, int k) -> bf16x8 {
    const int n = n0 + r;
    return n < NN ? ld8h(Wb + (size_t)n * K + k) : zero8h();
  };
  gemm_mainloop_bf16<false>(mma, sa, sb, fa, fb, 0, K, smem_h, tid);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = m0 + mma.row_of(ti, r);
      if (i >= nact) continue;
      const int64_t xo = a.plan.xo[trow + i];
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const int n = n0 + mma.col_of(tj);
        const float v = mma.acc[ti][tj][r];
        if (n < in) {
          if (a.dx) atomicAdd(a.dx + xo + n, v);
        } else if (n < NN && t > 0) {
          a.dh_rec[((size_t)dir * a.B + i) * d + (n - in)] = v;
        }
      }
    }
}

void launch_step_bwd_bf16(const StepBwdArgs& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL(lstm_step_bwd_bf16, grid, dim3(256), MmaSqB::LDS_BYTES, s, a);
}

// -------------------------------------------------------- weight gradient --
__global__ __launch_bounds__(256) void lstm_dw_bf16(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * MmaSqB::BM, n0 = blockIdx.y * MmaSqB::BN;
  const int split = blockIdx.z % a.nsplit, dir = blockIdx.z / a.nsplit;
  const int d = a.d, in = a.in, MM = in + d + 1, N4 = 4 * d;
  const int64_t out_ld = a.plan.hdr->out_ld;
  MmaSqB mma;
  mma.init(tid);
  StageKMajor<MmaSqB::BM, MmaSqB::BK, MmaSqB::NT, MmaSqB::LDM> sa;
  StageKMajor<MmaSqB::BN, MmaSqB::BK, MmaSqB::NT, MmaSqB::LDN> sb;
  const int t_begin = split * a.tgroup, t_end = min(a.J, t_begin + a.tgroup);
  for (int t = t_begin; t < t_end; ++t) {
    const int nact = a.plan.nactive[t];
    if (nact == 0) break;
    const size_t trow = ((size_t)dir * a.J + t) * a.B;
    const bf16_t* __restrict__ dz = a.dzb + trow * (size_t)N4;
    const int64_t* __restrict__ xo = a.plan.xo + trow;
    const int64_t* __restrict__ oo = a.plan.oo + trow;
    auto quad = [&](int k, int m) -> f32x4 {  // A[k = sorted row][m..m+3] of [x | h_prev | 1]
      if (m < in) return ld4(a.x + xo[k] + m);
      if (m < in + d) {
        if (t == 0) return zero4();
        const int64_t ho = dir ? oo[k] + out_ld : oo[k] - out_ld;
        return ld4(a.out + ho + (m - in));
      }
      return m == in + d ? f32x4{1.f, 0.f, 0.f, 0.f} : zero4();
    };
    auto fa = [&](int k, int c) -> bf16x8 {
      if (k >= nact) return zero8h();
      return cvt8(quad(k, m0 + c), quad(k, m0 + c + 4));
    };
    auto fb = [&](int k, int c) -> bf16x8 { return k < nact ? ld8h(dz + (size_t)k * N4 + n0 + c) : zero8h(); };
    gemm_mainloop_bf16<true>(mma, sa, sb, fa, fb, 0, (nact + MmaSqB::BK - 1) / MmaSqB::BK * MmaSqB::BK, smem_h, tid);
  }
  float* slab = a.slabs + (size_t)blockIdx.z * MM * N4;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mma.row_of(ti, r);
      if (m >= MM) continue;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) slab[(size_t)m * N4 + n0 + mma.col_of(tj)] = mma.acc[ti][tj][r];
    }
}

void launch_dw_bf16(const DwArgs& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL(lstm_dw_bf16, grid, dim3(256), MmaSqB::LDS_BYTES, s, a);
}

// ------------------------------------------------------------------ test gemm
// layout 1: C = A[M,K] * B[N,K]^T (row images);  layout 2: C = A[K,M]^T * B[K,N] (k-major images, tr reads)
template <int LAYOUT>
__global__ __launch_bounds__(256) void test_gemm_bf16_kernel(int M, int N, int K, const float* __restrict__ A,
                                                             const float* __restrict__ B, float* __restrict__ C) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * MmaSqB::BM, n0 = blockIdx.y * MmaSqB::BN;
  MmaSqB mma;
  mma.init(tid);
  const int Kp = (K + MmaSqB::BK - 1) / MmaSqB::BK * MmaSqB::BK;
  auto q = [&](const float* p, bool ok) -> f32x4 { return ok ? ld4(p) : zero4(); };
  if (LAYOUT == 1) {
    StageRows<MmaSqB::BM, MmaSqB::BK, MmaSqB::NT, MmaSqB::LDK> sa;
    StageRows<MmaSqB::BN, MmaSqB::BK, MmaSqB::NT, MmaSqB::LDK> sb;
    auto fa = [&](int r, int k) -> bf16x8 {
      const float* p = A + (size_t)(m0 + r) * K + k;
      return cvt8(q(p, m0 + r < M && k < K), q(p + 4, m0 + r < M && k + 4 < K));
    };
    auto fb = [&](int r, int k) -> bf16x8 {
      const float* p = B + (size_t)(n0 + r) * K + k;
      return cvt8(q(p, n0 + r < N && k < K), q(p + 4, n0 + r < N && k + 4 < K));
    };
    gemm_mainloop_bf16<false>(mma, sa, sb, fa, fb, 0, Kp, smem_h, tid);
  } else {
    StageKMajor<MmaSqB::BM, MmaSqB::BK, MmaSqB::NT, MmaSqB::LDM> sa;
    StageKMajor<MmaSqB::BN, MmaSqB::BK, MmaSqB::NT, MmaSqB::LDN> sb;
    auto fa = [&](int k, int c) -> bf16x8 {
      const float* p = A + (size_t)k * M + m0 + c;
      return cvt8(q(p, k < K && m0 + c < M), q(p + 4, k < K && m0 + c + 4 < M));
    };
    auto fb = [&](int k, int c) -> bf16x8 {
      const float* p = B + (size_t)k * N + n0 + c;
      return cvt8(q(p, k < K && n0 + c < N), q(p + 4, k < K && n0 + c + 4 < N));
    };
    gemm_mainloop_bf16<true>(mma, sa, sb, fa, fb, 0, Kp, smem_h, tid);
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mma.row_of(ti, r);
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const int n = n0 + mma.col_of(tj);
        if (m < M && n < N) C[(size_t)m * N + n] = mma.acc[ti][tj][r];
      }
    }
}

int test_gemm_bf16(int layout, int M, int N, int K, const float* A, const float* B, float* C, hipStream_t s) {
  const dim3 grid((M + MmaSqB::BM - 1) / MmaSqB::BM, (N + MmaSqB::BN - 1) / MmaSqB::BN);
  if (layout == 1)
    hipLaunchKernelGGL(test_gemm_bf16_kernel<1>, grid, dim3(256), MmaSqB::LDS_BYTES, s, M, N, K, A, B, C);
  else if (layout == 2)
    hipLaunchKernelGGL(test_gemm_bf16_kernel<2>, grid, dim3(256), MmaSqB::LDS_BYTES, s, M, N, K, A, B, C);
  else
    return FVTA_ERR_UNSUPPORTED;
  return FVTA_OK;
}

}  // namespace fvta
