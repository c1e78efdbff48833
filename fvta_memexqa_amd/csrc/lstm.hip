// bi-LSTM modality encoders for gfx950.
//
// Replaces model_v2.py:652-661 (BasicLSTMCell cells), 694-823 (eight
// bidirectional_dynamic_rnn calls) and 863-914 (tf.pad/tf.stack into `hall`):
// every sequence that shares a cell runs in one call, length-sorted so that the
// active rows of step t are a prefix; each step is one GEMM launch
// [x_t | h_{t-1}] * kernel with the gate math fused into the MFMA epilogue, the
// forward and the reversed direction side by side (blockIdx.z), and h_t stored
// straight into the padded context-tensor rows.
//
// Semantics restated from TF 1.4: z=[x,h]*kernel+bias, split
// i,j,f,o, c'=c*sig(f+1)+sig(i)*tanh(j), h'=tanh(c')*sig(o); rows past their
// length emit zeros and keep their state; the reversed direction runs on
// reverse_sequence(x, len) and its outputs are reversed back.
#include "gemm_f32.h"
#include "fvta_prof.h"
#include <vector>

#include "lstm_common.h"

namespace fvta {

__device__ const float k_ones4[4] = {1.f, 0.f, 0.f, 0.f};  // the [x | h | 1] ones column (dbias)

// ------------------------------------------------------------------ plan ----

// Stable counting sort by length, descending.  One 1024-thread workgroup; each
// of the 16 waves owns a contiguous chunk of the sequences, so the order is
// deterministic (no atomics decide a position).
__global__ __launch_bounds__(1024) void plan_sort_kernel(PlanView v, const int32_t* __restrict__ len_in,
                                                         const int32_t* __restrict__ seq_J_in,
                                                         const int64_t* __restrict__ x_off_in,
                                                         const int64_t* __restrict__ out_off_in,
                                                         int64_t out_ld, int B, int J, int in, int d,
                                                         int64_t x_bw_delta) {
  extern __shared__ int32_t sh[];  // [16][J+1] per-wave histogram, then running bases
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int NW = 16, H = J + 1;
  for (int i = tid; i < NW * H; i += 1024) sh[i] = 0;
  // (memory that has not been a plan of this shape before knows nothing about the caller's output buffer)
  const bool fresh = v.hdr->magic != PLAN_MAGIC || v.hdr->B != B || v.hdr->J != J || v.hdr->d != d || v.hdr->out_ld != out_ld;
  __syncthreads();
  if (tid == 0) {
    v.hdr->out_ld = out_ld;
    v.hdr->B = B;
    v.hdr->J = J;
    v.hdr->in = in;
    v.hdr->d = d;
    v.hdr->x_bw_delta = x_bw_delta;
    v.hdr->magic = PLAN_MAGIC;
  }
  // the plan's copies of the caller's arrays, four sequences per thread and round (sixteen loads in flight, then the stores)
  for (int bb = tid; bb < B; bb += 4096) {
    int Lc[4], Jc[4];
    int64_t xo_[4], oo_[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = min(bb + 1024 * r, B - 1);
      Lc[r] = len_in[b];
      Jc[r] = seq_J_in[b];
      xo_[r] = x_off_in[b];
      oo_[r] = out_off_in[b];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = bb + 1024 * r;
      if (b < B) {
        if (fresh || v.out_off[b] != oo_[r] || v.seq_J[b] != Jc[r]) v.dirty_out[b] = 0;  // another layout: nothing known
        v.len[b] = Lc[r] < 0 ? 0 : (Lc[r] > J ? J : Lc[r]);
        v.seq_J[b] = Jc[r];
        v.x_off[b] = xo_[r];
        v.out_off[b] = oo_[r];
      }
    }
  }
  const int chunk = (B + NW - 1) / NW;
  const int b0 = wave * chunk, b1 = min(B, b0 + chunk);
  // J + 1 <= 64 (every shape of the model): lane v counts the sequences of length v of its wave in a register -- one ballot
  // per possible length, no LDS update between them (the leader-by-leader form below pays an LDS round trip per distinct
  // length of every 64 sequences).  Chunks of <= 1024 sequences per wave (B <= 16384) keep their lengths in registers for
  // both passes, read straight from the caller's array (all loads in flight at once); the exclusive prefix over (length,
  // wave) is one wave's work (lane = length), and the scatter's running bases move by v_readlane.  (91 -> ~25 us at
  // B = 12,864: the old form walked its chunk with one dependent global load per 64 sequences and pass, and thread 0 alone
  // ran the 16 x (J + 1) prefix through LDS.)
  const bool lanes_hold = H <= 64;
  constexpr int MAXIT = 16;
  const int iters = b1 > b0 ? (b1 - b0 + 63) / 64 : 0;
  const bool in_regs = lanes_hold && chunk <= 64 * MAXIT;
  int Lr[MAXIT];
  if (in_regs) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int b = b0 + 64 * it + lane;
      int L = -1;
      if (it < iters && b < b1) {
        L = len_in[b];
        L = L < 0 ? 0 : (L > J ? J : L);
      }
      Lr[it] = L;
    }
  }
  __syncthreads();  // sh is zero; v.len is written (the paths below that read it back)
  // pass 1: per-wave histogram
  int hist = 0;
  if (in_regs) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
      if (it < iters)
        for (int val = 0; val < H; ++val) {
          const unsigned long long same = __ballot(Lr[it] == val);
          if (lane == val) hist += __popcll(same);
        }
  } else
  for (int base = b0; base < b1; base += 64) {
    const int b = base + lane;
    const bool ok = b < b1;
    const int L = ok ? v.len[b] : -1;
    if (lanes_hold) {
      for (int val = 0; val < H; ++val) {
        const unsigned long long same = __ballot(L == val);
        if (lane == val) hist += __popcll(same);
      }
      continue;
    }
    unsigned long long todo = __ballot(ok);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int Lv = __shfl(L, leader, 64);
      const unsigned long long same = __ballot(ok && L == Lv);
      if (lane == leader) sh[wave * H + Lv] += __popcll(same);
      todo &= ~same;
    }
  }
  if (lanes_hold && lane < H) sh[wave * H + lane] = hist;
  __syncthreads();
  // exclusive prefix over (L descending, wave ascending)
  if (lanes_hold) {
    if (wave == 0) {  // lane = length
      int cnt[NW], total = 0;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) {
        cnt[wv] = lane < H ? sh[wv * H + lane] : 0;
        total += cnt[wv];
      }
      int run = 0, mine = 0;  // run: sequences longer than L (uniform)
      for (int L = J; L >= 0; --L) {
        if (lane == L) mine = run;
        run += __builtin_amdgcn_readlane(total, L);
      }
      if (lane < J) v.nactive[lane] = mine;  // sequences with len > lane
      if (lane == J) v.nactive[J] = 0;
      if (lane < H) {
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) {
          sh[wv * H + lane] = mine;
          mine += cnt[wv];
        }
      }
    }
  } else if (tid == 0) {
    int run = 0;
    for (int L = J; L >= 0; --L) {
      if (L < J) v.nactive[L] = run;  // sequences with len > L
      for (int wv = 0; wv < NW; ++wv) {
        const int c = sh[wv * H + L];
        sh[wv * H + L] = run;
        run += c;
      }
    }
    v.nactive[J] = 0;
  }
  __syncthreads();
  // pass 2: scatter, each wave walking its chunk in order (lane v carries the running base of length v)
  int basev = (lanes_hold && lane < H) ? sh[wave * H + lane] : 0;
  if (in_regs) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
      if (it < iters) {
        const int b = b0 + 64 * it + lane;
        const int L = Lr[it];
        for (int val = 0; val < H; ++val) {
          const unsigned long long same = __ballot(L == val);
          if (!same) continue;
          const int bp = __builtin_amdgcn_readlane(basev, val);
          if (L == val) v.order[bp + __popcll(same & ((1ull << lane) - 1ull))] = b;
          if (lane == val) basev += __popcll(same);
        }
      }
    return;
  }
  for (int base = b0; base < b1; base += 64) {
    const int b = base + lane;
    const bool ok = b < b1;
    const int L = ok ? v.len[b] : -1;
    if (lanes_hold) {
      for (int val = 0; val < H; ++val) {
        const unsigned long long same = __ballot(L == val);
        const int bp = __shfl(basev, val, 64);
        if (L == val) v.order[bp + __popcll(same & ((1ull << lane) - 1ull))] = b;
        if (lane == val) basev += __popcll(same);
      }
      continue;
    }
    unsigned long long todo = __ballot(ok);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int Lv = __shfl(L, leader, 64);
      const unsigned long long same = __ballot(ok && L == Lv);
      const int basepos = sh[wave * H + Lv];
      if (ok && L == Lv) {
        const unsigned long long lower = same & ((1ull << lane) - 1ull);
        v.order[basepos + __popcll(lower)] = b;
      }
      // make sure every lane has read basepos before the leader bumps it
      __builtin_amdgcn_wave_barrier();
      if (lane == leader) sh[wave * H + Lv] = basepos + __popcll(same);
      __builtin_amdgcn_wave_barrier();
      todo &= ~same;
    }
  }
}

__global__ void plan_fill_kernel(PlanView v, int B, int J, int in, int d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y, dir = blockIdx.z;
  if (i >= B) return;
  const size_t idx = ((size_t)dir * J + t) * B + i;
  if (i >= v.nactive[t]) {
    v.xo[idx] = -1;
    v.oo[idx] = -1;
    return;
  }
  const int b = v.order[i];
  const int L = v.len[b];
  const int pos = dir ? (L - 1 - t) : t;
  v.xo[idx] = v.x_off[b] + (int64_t)pos * in + (dir ? v.hdr->x_bw_delta : 0);
  v.oo[idx] = v.out_off[b] + (int64_t)pos * v.hdr->out_ld + (int64_t)dir * d;
}

// rows t in [len, seq_J) of both halves are zero (dynamic_rnn zero_output).  grid B
// persist (desc.out_pads_persist: nobody but this op writes the output buffer between forward calls): only the rows the
// LAST forward on this plan memory wrote and this one will not -- [len, dirty) -- are zeroed; the plan remembers per sequence
// how far it has written into which buffer.  (A ragged metric-shape batch: 0.79 GB of zeros per step otherwise, 161 us.)
__global__ __launch_bounds__(256) void pad_zero_kernel(PlanView v, float* __restrict__ out, int d, int persist, int B, int64_t out_skip) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // a wave per sequence
  if (b >= B) return;
  if (v.out_off[b] < out_skip) return;  // (desc.out_skip: this sequence's output rows are not stored at all)
  const int L = v.len[b], Jb = v.seq_J[b];
  const int64_t ld = v.hdr->out_ld;
  float* first = out + v.out_off[b];
  const bool known = persist && v.dirty_out[b] == (int64_t)reinterpret_cast<uintptr_t>(first);
  const int hi = known ? min(v.dirty[b], Jb) : Jb;
  // (the update is behind the reads of the sequence's state: same wave, program order, and its predicate needs their data)
  if (lane == 0) {
    v.dirty[b] = L;
    v.dirty_out[b] = (int64_t)reinterpret_cast<uintptr_t>(first);
  }
  const int w4 = (2 * d) / 4;
  for (int t = L; t < hi; ++t) {
    f32x4* row = reinterpret_cast<f32x4*>(first + (int64_t)t * ld);
    for (int c = lane; c < w4; c += 64) row[c] = zero4();
  }
}

// desc.dx_overwrite: rows [t0, seq_J) of every sequence's dx are zeroed -- t0 = len (the padded positions: the dx kernel
// writes the rest) or 0 (all rows: the dx kernels of this engine / shape add to dx).  Both copies under a separate input per
// direction.  grid (B, 4)
__global__ __launch_bounds__(256) void dx_zero_rows_kernel(PlanView v, float* __restrict__ dx, int in, int pads_only, int B) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // a wave per sequence
  if (b >= B) return;
  const int L = pads_only ? v.len[b] : 0, Jb = v.seq_J[b];
  const int64_t delta = v.hdr->x_bw_delta;
  const int w4 = in / 4;
  for (int t = L; t < Jb; ++t) {
    float* row = dx + v.x_off[b] + (int64_t)t * in;
    if ((reinterpret_cast<uintptr_t>(row) & 15) == 0 && (delta & 3) == 0) {
      for (int c = lane; c < w4; c += 64) {
        reinterpret_cast<f32x4*>(row)[c] = zero4();
        if (delta) reinterpret_cast<f32x4*>(row + delta)[c] = zero4();
      }
    } else {
      for (int c = lane; c < in; c += 64) {
        row[c] = 0.f;
        if (delta) row[delta + c] = 0.f;
      }
    }
  }
}

// ------------------------------------------------------------- saved state --

// --------------------------------------------------------- forward step -----
// grid (ceil(B/128), d/32, 2).  Block tile: 128 sorted sequences x (4 gates x 32
// units); wave wv owns rows [32wv, 32wv+32) and all four gate tiles, so a lane
// ends up with i,j,f,o of the same (row, unit) in its accumulators.
using MmaStep = MmaF32<4, 1, 1, 4>;


__global__ __launch_bounds__(256) void lstm_step_fwd_f32(StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int64_t s_xo[MmaStep::BM];
  __shared__ int64_t s_ho[MmaStep::BM];
  __shared__ int64_t s_oo[MmaStep::BM];
  const int tid = threadIdx.x;
  const int dir = blockIdx.z;
  const int m0 = blockIdx.x * MmaStep::BM;
  const int nact = a.plan.nactive[a.t];
  if (m0 >= nact) return;
  const int u0 = blockIdx.y * 32;
  const int d = a.d, in = a.in, t = a.t;
  const int64_t out_ld = a.plan.hdr->out_ld;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  if (tid < MmaStep::BM) {
    const int i = m0 + tid;
    int64_t xo = -1, oo = -1;
    if (i < nact) {
      xo = a.plan.xo[trow + i];
      oo = a.plan.oo[trow + i];
    }
    s_xo[tid] = xo;
    s_oo[tid] = oo;
    // h_{t-1} sits one output row before (fw) / after (bw) this step's row
    s_ho[tid] = (oo < 0 || t == 0) ? -1 : (dir ? oo + out_ld : oo - out_ld);
  }
  __syncthreads();
  const float* __restrict__ W = a.W[dir];
  const float* __restrict__ x = a.x;
  const float* __restrict__ hsrc = a.out;
  const int K = (t == 0) ? in : in + d;

  MmaStep mma;
  mma.init(tid);
  StageKContig<MmaStep::BM, MmaStep::BK, MmaStep::NT, MmaStep::LDA> sa;
  StageMNContig<MmaStep::BN, MmaStep::BK, MmaStep::NT, MmaStep::LDB> sb;
  auto fa = [&](int r, int k, bool& ok) -> const float* {  // 4 consecutive k of row r of [x | h_prev]
    const bool isx = k < in;
    const int64_t o = isx ? s_xo[r] : s_ho[r];
    ok = (k < K) && (o >= 0);
    const float* base = isx ? x + k : hsrc + (k - in);
    return ok ? base + o : x;  // x itself is always dereferenceable
  };
  auto fb = [&](int k, int c, bool& ok) -> const float* {
    ok = k < K;
    const int g = c >> 5, u = c & 31;  // virtual column -> gate strip
    return W + (size_t)(ok ? k : 0) * (4 * d) + g * d + u0 + u;
  };
  gemm_mainloop(mma, sa, sb, fa, fb, 0, K, smem, tid);

  lstm_gate_epilogue(mma, a, dir, m0, u0, nact, trow, s_oo);
}

// ------------------------------------------------------- backward pieces ----
// (a) elementwise: dz_t from dh_t, dc_t and the saved activations; dz overwrites
// the saved gates in place.  grid (ceil(nact_max*d/256)...): one thread per (i,u).
__global__ void lstm_gate_bwd(GateBwdArgs a) {
  const int dir = blockIdx.z;
  const int d = a.d, t = a.t;
  const int nact = a.plan.nactive[t];
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)(idx / d), u = (int)(idx % d);
  if (i >= nact) return;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  const int64_t oo = a.plan.oo[trow + i];
  const size_t su = ((size_t)dir * a.B + i) * d + u;
  // rows whose last step is t have never been written by a later step: dh_rec, dc are still 0
  const float dh = a.d_out[oo + u] + a.dh_rec[su];
  float* g = a.gates + (trow + i) * (size_t)(4 * d) + u;
  float ig, jg, fg, og;
  if (a.gatesb) {
    const bf16x4 pk = *reinterpret_cast<const bf16x4*>(a.gatesb + (trow + i) * (size_t)(4 * d) + 4 * u);
    ig = bf2f((bf16_t)pk[0]), jg = bf2f((bf16_t)pk[1]), fg = bf2f((bf16_t)pk[2]), og = bf2f((bf16_t)pk[3]);
  } else {
    ig = g[0], jg = g[d], fg = g[2 * d], og = g[3 * d];
  }
  const float c = a.cs[(trow + i) * d + u];
  const float cprev = t > 0 ? a.cs[(trow - a.B + i) * d + u] : 0.f;
  const float tc = fvta_tanh(c);
  const float dc = a.dc[su] + dh * og * (1.f - tc * tc);
  const float dzi = dc * jg * ig * (1.f - ig), dzj = dc * ig * (1.f - jg * jg);
  const float dzf = dc * cprev * fg * (1.f - fg), dzo = dh * tc * og * (1.f - og);
  if (a.dzb) {  // bf16 engine: dz rows are only ever MFMA operands
    bf16x4 pk;  // unit-major [row][u][4]: the k order of wb matches
    pk[0] = (short)f2bf(dzi);
    pk[1] = (short)f2bf(dzj);
    pk[2] = (short)f2bf(dzf);
    pk[3] = (short)f2bf(dzo);
    *reinterpret_cast<bf16x4*>(a.dzb + (trow + i) * (size_t)(4 * d) + 4 * u) = pk;
  } else {
    g[0] = dzi;
    g[d] = dzj;
    g[2 * d] = dzf;
    g[3 * d] = dzo;
  }
  a.dc[su] = dc * fg;
}

// (b) [dx_t | dh_{t-1}] = dz_t * kernel^T.  grid (ceil(B/128), ceil((in+d)/128), 2)
using MmaSq = MmaF32<2, 2, 2, 2>;
__global__ __launch_bounds__(256) void lstm_step_bwd_f32(StepBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, dir = blockIdx.z;
  const int m0 = blockIdx.x * MmaSq::BM, n0 = blockIdx.y * MmaSq::BN;
  const int nact = a.plan.nactive[a.t];
  if (m0 >= nact) return;
  const int d = a.d, in = a.in, t = a.t;
  const int NN = in + d, K = 4 * d;
  if (t == 0 && n0 >= in) return;  // dh_{-1} is not needed
  if (a.dx == nullptr && n0 + MmaSq::BN <= in) return;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  const float* __restrict__ dz = a.dz + trow * (size_t)K;
  const float* __restrict__ W = a.W[dir];
  MmaSq mma;
  mma.init(tid);
  StageKContig<MmaSq::BM, MmaSq::BK, MmaSq::NT, MmaSq::LDA> sa;
  StageKContig<MmaSq::BN, MmaSq::BK, MmaSq::NT, MmaSq::LDB> sb;
  auto fa = [&](int r, int k, bool& ok) -> const float* {
    const int i = m0 + r;
    ok = i < nact;
    return dz + (size_t)(ok ? i : 0) * K + k;
  };
  auto fb = [&](int r, int k, bool& ok) -> const float* {
    const int n = n0 + r;
    ok = n < NN;
    return W + (size_t)(ok ? n : 0) * K + k;
  };
  gemm_mainloop(mma, sa, sb, fa, fb, 0, K, smem, tid);
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
          // both directions add into the same dx row (at most two addends: order-free)
          if (a.dx) atomicAdd(a.dx + xo + n, v);
        } else if (n < NN && t > 0) {
          a.dh_rec[((size_t)dir * a.B + i) * d + (n - in)] = v;
        }
      }
    }
}

// (c) dkernel = [x | h_prev | 1]^T * dz over all steps: split-K slabs.
// grid (ceil((in+d+1)/128), 4d/128, nsplit*ndirslab)
__global__ __launch_bounds__(256) void lstm_dw_f32(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * MmaSq::BM, n0 = blockIdx.y * MmaSq::BN;
  const int split = blockIdx.z % a.nsplit, dir = blockIdx.z / a.nsplit;
  const int d = a.d, in = a.in, MM = in + d + 1, N4 = 4 * d;
  const int64_t out_ld = a.plan.hdr->out_ld;
  MmaSq mma;
  mma.init(tid);
  StageMNContig<MmaSq::BM, MmaSq::BK, MmaSq::NT, MmaSq::LDA> sa;
  StageMNContig<MmaSq::BN, MmaSq::BK, MmaSq::NT, MmaSq::LDB> sb;
  const int t_begin = split * a.tgroup, t_end = min(a.J, t_begin + a.tgroup);
  for (int t = t_begin; t < t_end; ++t) {
    const int nact = a.plan.nactive[t];
    if (nact == 0) break;
    const size_t trow = ((size_t)dir * a.J + t) * a.B;
    const float* __restrict__ dz = a.dz + trow * (size_t)N4;
    const int64_t* __restrict__ xo = a.plan.xo + trow;
    const int64_t* __restrict__ oo = a.plan.oo + trow;
    auto fa = [&](int k, int c, bool& ok) -> const float* {  // A[k = sorted row][m] of [x | h_prev | 1]
      const int m = m0 + c;
      const int kk = k < nact ? k : 0;
      const bool isx = m < in, ish = !isx && m < in + d;
      const int64_t o = isx ? xo[kk] : (dir ? oo[kk] + out_ld : oo[kk] - out_ld);
      ok = (k < nact) && (isx || (ish && t > 0) || m == in + d);
      const float* src = isx ? a.x + o + m : a.out + o + (m - in);
      return (ok && (isx || ish)) ? src : k_ones4;  // ones column -> dbias
    };
    auto fb = [&](int k, int c, bool& ok) -> const float* {
      ok = k < nact;
      return dz + (size_t)(ok ? k : 0) * N4 + n0 + c;
    };
    gemm_mainloop(mma, sa, sb, fa, fb, 0, (nact + MmaSq::BK - 1) / MmaSq::BK * MmaSq::BK, smem, tid);
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

// sum the slabs in a fixed order into dkernel [in+d,4d] and dbias [4d] (accumulate)
__global__ void lstm_dw_reduce(const float* __restrict__ slabs, int nslab, size_t slab_elems, int rowsW,
                               int N4, float* __restrict__ dW, float* __restrict__ dbias) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= slab_elems) return;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += slabs[(size_t)k * slab_elems + idx];
  const size_t row = idx / N4;
  if (row < (size_t)rowsW)
    dW[idx] += s;
  else
    dbias[idx - (size_t)rowsW * N4] += s;
}

// ------------------------------------------------------------ last states ---
__global__ void last_state_kernel(PlanView v, const float* __restrict__ out, int s0, int count, int d,
                                  float* __restrict__ dst, const float* __restrict__ d_dst,
                                  float* __restrict__ d_out) {
  const int s = blockIdx.x;
  if (s >= count) return;
  const int b = s0 + s;
  const int L = v.len[b];
  const int64_t ld = v.hdr->out_ld;
  const int64_t base = v.out_off[b];
  for (int c = threadIdx.x; c < 2 * d; c += blockDim.x) {
    const int64_t o = base + (c < d ? (int64_t)(L - 1) * ld : 0) + c;
    if (dst) dst[(size_t)s * 2 * d + c] = L > 0 ? out[o] : 0.f;
    if (d_out && L > 0) d_out[o] += d_dst[(size_t)s * 2 * d + c];
  }
}

// ---------------------------------------------------------------- test gemm -
template <int LAYOUT>
__global__ __launch_bounds__(256) void test_gemm_f32(int M, int N, int K, const float* __restrict__ A,
                                                     const float* __restrict__ B, float* __restrict__ C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * MmaSq::BM, n0 = blockIdx.y * MmaSq::BN;
  MmaSq mma;
  mma.init(tid);
  const int Kp = (K + MmaSq::BK - 1) / MmaSq::BK * MmaSq::BK;
  if (LAYOUT == 0) {  // A[M,K] k-contig, B[K,N] n-contig
    StageKContig<MmaSq::BM, MmaSq::BK, MmaSq::NT, MmaSq::LDA> sa;
    StageMNContig<MmaSq::BN, MmaSq::BK, MmaSq::NT, MmaSq::LDB> sb;
    auto fa = [&](int r, int k, bool& ok) -> const float* { ok = m0 + r < M && k < K; return ok ? A + (size_t)(m0 + r) * K + k : A; };
    auto fb = [&](int k, int c, bool& ok) -> const float* { ok = k < K && n0 + c < N; return ok ? B + (size_t)k * N + n0 + c : B; };
    gemm_mainloop(mma, sa, sb, fa, fb, 0, Kp, smem, tid);
  } else if (LAYOUT == 1) {  // A[M,K], B[N,K] both k-contig
    StageKContig<MmaSq::BM, MmaSq::BK, MmaSq::NT, MmaSq::LDA> sa;
    StageKContig<MmaSq::BN, MmaSq::BK, MmaSq::NT, MmaSq::LDB> sb;
    auto fa = [&](int r, int k, bool& ok) -> const float* { ok = m0 + r < M && k < K; return ok ? A + (size_t)(m0 + r) * K + k : A; };
    auto fb = [&](int r, int k, bool& ok) -> const float* { ok = n0 + r < N && k < K; return ok ? B + (size_t)(n0 + r) * K + k : B; };
    gemm_mainloop(mma, sa, sb, fa, fb, 0, Kp, smem, tid);
  } else {  // A[K,M], B[K,N] both k-major
    StageMNContig<MmaSq::BM, MmaSq::BK, MmaSq::NT, MmaSq::LDA> sa;
    StageMNContig<MmaSq::BN, MmaSq::BK, MmaSq::NT, MmaSq::LDB> sb;
    auto fa = [&](int k, int c, bool& ok) -> const float* { ok = k < K && m0 + c < M; return ok ? A + (size_t)k * M + m0 + c : A; };
    auto fb = [&](int k, int c, bool& ok) -> const float* { ok = k < K && n0 + c < N; return ok ? B + (size_t)k * N + n0 + c : B; };
    gemm_mainloop(mma, sa, sb, fa, fb, 0, Kp, smem, tid);
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

}  // namespace fvta

using namespace fvta;

// ================================================================== C ABI ===
static int check_lstm_desc(const fvta_lstm_desc* d) {
  FVTA_CHECK_ARG(d != nullptr, "lstm: null descriptor");
  FVTA_CHECK_ARG(d->B > 0 && d->J > 0 && d->J <= 1024, "lstm: need B>0 and 0<J<=1024 (B=%d J=%d)", d->B, d->J);
  FVTA_CHECK_ARG(d->in > 0 && d->in % 4 == 0, "lstm: input width must be a positive multiple of 4 (in=%d)", d->in);
  FVTA_CHECK_ARG(d->d > 0 && d->d % 32 == 0, "lstm: hidden size must be a positive multiple of 32 (d=%d)", d->d);
  FVTA_CHECK_ARG(d->precision == FVTA_F32 || d->precision == FVTA_BF16 || d->precision == FVTA_BF16X3,
                 "lstm: unknown precision %d", d->precision);
  return FVTA_OK;
}

extern "C" size_t fvta_lstm_plan_bytes(const fvta_lstm_desc* d) { return plan_view(d, nullptr).bytes; }
extern "C" size_t fvta_lstm_saved_bytes(const fvta_lstm_desc* d) { return saved_view(d, nullptr).bytes; }
extern "C" size_t fvta_lstm_workspace_bytes(const fvta_lstm_desc* d) { return work_view(d, nullptr).bytes; }

extern "C" int fvta_lstm_plan(const fvta_lstm_desc* d, const int32_t* len, const int32_t* seq_J,
                              const int64_t* x_off, const int64_t* out_off, int64_t out_ld, void* plan,
                              fvta_stream_t stream_) {
  return fvta_lstm_plan_xdir(d, len, seq_J, x_off, out_off, out_ld, 0, plan, stream_);
}

extern "C" int fvta_lstm_plan_xdir(const fvta_lstm_desc* d, const int32_t* len, const int32_t* seq_J,
                                   const int64_t* x_off, const int64_t* out_off, int64_t out_ld, int64_t x_bw_delta,
                                   void* plan, fvta_stream_t stream_) {
  if (int e = check_lstm_desc(d)) return e;
  FVTA_CHECK_ARG(len && seq_J && x_off && out_off && plan, "lstm_plan: null pointer");
  FVTA_CHECK_ARG(x_bw_delta >= 0 && x_bw_delta % 4 == 0, "lstm_plan: x_bw_delta=%lld must be >= 0 and a multiple of 4",
                 (long long)x_bw_delta);
  FVTA_CHECK_ARG(out_ld >= 2 * d->d && out_ld % 4 == 0, "lstm_plan: out_ld=%lld must be >= 2d and a multiple of 4",
                 (long long)out_ld);
  hipStream_t stream = (hipStream_t)stream_;
  PlanView v = plan_view(d, plan);
  const size_t sh = (size_t)16 * (d->J + 1) * sizeof(int32_t);
  hipLaunchKernelGGL(plan_sort_kernel, dim3(1), dim3(1024), sh, stream, v, len, seq_J, x_off, out_off, out_ld,
                     d->B, d->J, d->in, d->d, x_bw_delta);
  FVTA_CHECK_LAUNCH("plan_sort");
  hipLaunchKernelGGL(plan_fill_kernel, dim3((d->B + 255) / 256, d->J, 2), dim3(256), 0, stream, v, d->B, d->J,
                     d->in, d->d);
  FVTA_CHECK_LAUNCH("plan_fill");
  return FVTA_OK;
}

extern "C" int fvta_bilstm_fwd(const fvta_lstm_desc* d, const void* plan, const float* x, float* out,
                               const float* kernel_fw, const float* bias_fw, const float* kernel_bw,
                               const float* bias_bw, void* saved, void* workspace, fvta_stream_t stream_) {
  if (int e = check_lstm_desc(d)) return e;
  FVTA_CHECK_ARG(plan && x && out && kernel_fw && bias_fw && workspace, "bilstm_fwd: null pointer");
  FVTA_CHECK_ARG(d->share_fw_bw || (kernel_bw && bias_bw), "bilstm_fwd: kernel_bw/bias_bw required");
  FVTA_CHECK_ARG(saved, "bilstm_fwd: saved buffer required (size it with fvta_lstm_saved_bytes)");
  hipStream_t stream = (hipStream_t)stream_;
  PlanView pv = plan_view(d, const_cast<void*>(plan));
  WorkView wv = work_view(d, workspace);
  StepArgs a;
  a.plan = pv;
  a.x = x;
  a.out = out;
  a.W[0] = kernel_fw;
  a.bias[0] = bias_fw;
  a.W[1] = d->share_fw_bw ? kernel_fw : kernel_bw;
  a.bias[1] = d->share_fw_bw ? bias_fw : bias_bw;
  SavedView sv = saved_view(d, saved);
  a.gates = sv.gates;  // null when not training
  a.gatesb = sv.gatesb;
  a.cs = sv.cs;
  a.xs = sv.xs;        // null for the fp32 engine
  a.hs = sv.hs;
  a.cstate = wv.cstate;
  a.B = d->B;
  a.J = d->J;
  a.in = d->in;
  a.d = d->d;
  FVTA_CHECK_ARG(d->out_skip == 0 || (d->precision == FVTA_BF16 && d->out_skip > 0),
                 "bilstm_fwd: out_skip needs the bf16 engine (the rows' readers take its bf16 shadow rows)");
  a.out_skip = d->out_skip;
  hipLaunchKernelGGL(pad_zero_kernel, dim3((d->B + 3) / 4), dim3(256), 0, stream, pv, out, d->d, d->out_pads_persist, d->B,
                     (int64_t)d->out_skip);
  FVTA_CHECK_LAUNCH("pad_zero");
  const dim3 grid((d->B + MmaStep::BM - 1) / MmaStep::BM, d->d / 32, 2);
  const size_t sh = MmaStep::LDS_FLOATS * sizeof(float);
  fvta_prof_begin(FVTA_PROF_LSTM_STEP_FWD + 16 * d->reserved, stream);
  const bool bf = lstm_is_bf(d);
  a.Kp = kpad8(d);
  a.xm = lstm_xm(d);
  a.dbg = fvta_diag_env("FVTA_DEBUG_SKIP", 0);  // -DFVTA_DIAG builds only
  a.nt = 0;
  a.Wt[0] = a.Wt[1] = nullptr;
  a.Wf[0] = a.Wf[1] = nullptr;
  if (bf) {  // refresh the bf16 weight shadows (the optimiser has just changed the fp32 masters)
    const int ndir = d->share_fw_bw ? 1 : 2;
    for (int i = 0; i < ndir; ++i)
      launch_cvt_weights_bf16(a.W[i], wv.wt[i], wv.wb[i], d->in, in_internal(d), d->d, a.xm, stream);
    a.Wt[0] = wv.wt[0];
    a.Wt[1] = d->share_fw_bw ? wv.wt[0] : wv.wt[1];
    if (a.xm == 1 ? wreg_nct(in_internal(d), d->d) != 0 : wreg_x3_built(in_internal(d), d->d)) {
      for (int i = 0; i < ndir; ++i)
        launch_cvt_weights_frag(a.W[i], a.bias[i], wv.wf[i], d->in, in_internal(d), d->d, a.xm, stream);
      a.Wf[0] = wv.wf[0];
      a.Wf[1] = d->share_fw_bw ? wv.wf[0] : wv.wf[1];
    }
    launch_cvt_x_bf16(pv, x, sv.xs, d->B, d->J, d->in, in_internal(d), a.xm, stream);
  }
  const int launches = d->J;
  for (int t = 0; t < d->J; ++t) {
    a.t = t;
    if (bf) {  // weights-in-registers kernel where it is built for the shape, the tiled one otherwise
      if (!launch_step_fwd_wreg(a, stream)) launch_step_fwd_bf16(a, stream);
    } else
      hipLaunchKernelGGL(lstm_step_fwd_f32, grid, dim3(256), sh, stream, a);
  }
  fvta_prof_end(FVTA_PROF_LSTM_STEP_FWD + 16 * d->reserved, launches, stream);
  FVTA_CHECK_LAUNCH("lstm_step_fwd");
  return FVTA_OK;
}

extern "C" int fvta_bilstm_bwd(const fvta_lstm_desc* d, const void* plan, const float* x, const float* out,
                               const float* d_out, const float* kernel_fw, const float* kernel_bw, void* saved,
                               float* dx, float* dkernel_fw, float* dbias_fw, float* dkernel_bw,
                               float* dbias_bw, void* workspace, fvta_stream_t stream_) {
  return fvta_bilstm_bwd_hint(d, plan, x, out, d_out, kernel_fw, kernel_bw, saved, dx, dkernel_fw, dbias_fw, dkernel_bw,
                              dbias_bw, workspace, stream_, nullptr, nullptr);
}

extern "C" int fvta_bilstm_bwd_overlap(const fvta_lstm_desc* d, const void* plan, const float* x, const float* out,
                                       const float* d_out, const float* kernel_fw, const float* kernel_bw, void* saved,
                                       float* dx, float* dkernel_fw, float* dbias_fw, float* dkernel_bw,
                                       float* dbias_bw, void* workspace, fvta_stream_t stream_,
                                       fvta_stream_t side_stream_) {
  return fvta_bilstm_bwd_hint(d, plan, x, out, d_out, kernel_fw, kernel_bw, saved, dx, dkernel_fw, dbias_fw, dkernel_bw,
                              dbias_bw, workspace, stream_, side_stream_, nullptr);
}

extern "C" int fvta_bilstm_bwd_hint(const fvta_lstm_desc* d, const void* plan, const float* x, const float* out,
                                    const float* d_out, const float* kernel_fw, const float* kernel_bw, void* saved,
                                    float* dx, float* dkernel_fw, float* dbias_fw, float* dkernel_bw,
                                    float* dbias_bw, void* workspace, fvta_stream_t stream_,
                                    fvta_stream_t side_stream_, const int32_t* nactive_host) {
  if (int e = check_lstm_desc(d)) return e;
  FVTA_CHECK_ARG(d->training, "bilstm_bwd: forward was not run with training=1");
  FVTA_CHECK_ARG(plan && x && out && d_out && kernel_fw && saved && dkernel_fw && dbias_fw && workspace,
                 "bilstm_bwd: null pointer");
  FVTA_CHECK_ARG(d->share_fw_bw || (kernel_bw && dkernel_bw && dbias_bw), "bilstm_bwd: *_bw pointers required");
  hipStream_t stream = (hipStream_t)stream_;
  PlanView pv = plan_view(d, const_cast<void*>(plan));
  SavedView sv = saved_view(d, saved);
  WorkView wv = work_view(d, workspace);
  const int B = d->B, J = d->J, in = d->in, dd = d->d;
  // dc and dh_rec start at zero (fp32 engine).  The bf16 engines' step kernels read dc of rows that were not active at
  // step t + 1 as zero themselves and have no dh_rec round trip: no 105 MB fill per call at the metric shape
  if (!lstm_is_bf(d)) FVTA_CHECK_HIP(hipMemsetAsync(wv.cstate, 0, (size_t)2 * B * dd * sizeof(float) * 2, stream));
  GateBwdArgs g;
  g.plan = pv;
  g.d_out = d_out;
  g.gates = sv.gates;
  g.gatesb = sv.gatesb;
  g.cs = sv.cs;
  g.dc = wv.cstate;
  g.dh_rec = wv.dh_rec;
  const bool bf = lstm_is_bf(d);
  g.dzb = bf ? wv.dzb : nullptr;
  g.B = B;
  g.J = J;
  g.d = dd;
  StepBwdArgs s;
  s.plan = pv;
  s.dz = sv.gates;
  s.W[0] = kernel_fw;
  s.W[1] = d->share_fw_bw ? kernel_fw : kernel_bw;
  s.dx = dx;
  s.dh_rec = wv.dh_rec;
  s.dzb = wv.dzb;
  s.in_i = in_internal(d);
  const int dbg = fvta_diag_env("FVTA_DEBUG_SKIP", 0);  // -DFVTA_DIAG builds only
  s.dbg = dbg;
  s.Wb[0] = wv.wb[0];
  s.Wb[1] = d->share_fw_bw ? wv.wb[0] : wv.wb[1];
  s.B = B;
  s.J = J;
  s.in = in;
  s.d = dd;
  const dim3 ggrid((unsigned)(((size_t)B * dd + 255) / 256), 1, 2);
  const dim3 sgrid((B + MmaSq::BM - 1) / MmaSq::BM, (in + dd + MmaSq::BN - 1) / MmaSq::BN, 2);
  const size_t sh = MmaSq::LDS_FLOATS * sizeof(float);
  // the weight-gradient launch arguments (both engines)
  DwArgs w;
  w.plan = pv;
  w.x = x;
  w.out = out;
  w.dz = sv.gates;
  w.slabs = wv.slabs;
  w.B = B;
  w.J = J;
  w.in = in;
  w.d = dd;
  w.tgroup = dw_tgroup(d);
  w.nsplit = dw_nsplit(d);
  w.dzb = wv.dzb;
  w.in_i = in_internal(d);
  w.xs = sv.xs;
  w.hs = sv.hs;
  w.xm = lstm_xm(d);
  // (side_stream_: accepted for ABI stability, unused -- running dx / the weight gradient beside the recurrence was
  //  measured slower, DESIGN.md appendix.  nactive_host, when given, picks each backward step's block tile.)
  (void)side_stream_;
  fvta_prof_begin(FVTA_PROF_LSTM_STEP_BWD + 16 * d->reserved, stream);
  if (bf) {
    // bf16 engine: ONE launch per step -- dh_{t} = dz_{t+1} * wb_h^T in the k-loop, the gate gradient
    // (dz_t, dc) as its epilogue -- and dx = dz * wb_x^T batched over all steps
    FusedBwdArgs f;
    f.plan = pv;
    f.Wb[0] = wv.wb[0];
    f.Wb[1] = d->share_fw_bw ? wv.wb[0] : wv.wb[1];
    f.gatesb = sv.gatesb;
    f.cs = sv.cs;
    f.d_out = d_out;
    f.dzb = wv.dzb;
    f.dc = wv.cstate;
    f.dx = dx;
    f.B = B;
    f.J = J;
    f.in = in;
    f.d = dd;
    f.in_i = in_internal(d);
    f.gates32 = sv.gates;  // split engine: fp32 gates (sv.gatesb is null there)
    f.xm = lstm_xm(d);
    f.dx_accumulate = 1;  // the ABI's contract: dx is accumulated into
    f.dx_both = 0;
    if (dx && d->dx_overwrite) {  // ... or written: the padded rows zeroed here, the rest by the dx kernel where it can
      const bool direct = dx_writes_whole_rows(f);
      hipLaunchKernelGGL(dx_zero_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, pv, dx, in, direct ? 1 : 0, B);
      f.dx_accumulate = direct ? 0 : 1;
    }
    for (int t = J - 1; t >= 0; --t) {
      f.t = t;
      f.nact_hint = nactive_host ? nactive_host[t] : -1;
      launch_bwd_fused_bf16(f, stream);
    }
    fvta_prof_end(FVTA_PROF_LSTM_STEP_BWD + 16 * d->reserved, J, stream);
    if (dx) {
      fvta_prof_begin(FVTA_PROF_LSTM_DX + 16 * d->reserved, stream);
      launch_dx_bf16(f, stream);
      fvta_prof_end(FVTA_PROF_LSTM_DX + 16 * d->reserved, 1, stream);
    }
  } else {
  if (dx && d->dx_overwrite) hipLaunchKernelGGL(dx_zero_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, pv, dx, in, 0, B);
  for (int t = J - 1; t >= 0; --t) {
    g.t = t;
    s.t = t;
    if (!(dbg & 512)) hipLaunchKernelGGL(lstm_gate_bwd, ggrid, dim3(256), 0, stream, g);
    if (t > 0 || dx) hipLaunchKernelGGL(lstm_step_bwd_f32, sgrid, dim3(256), sh, stream, s);
  }
  }
  if (!bf) fvta_prof_end(FVTA_PROF_LSTM_STEP_BWD + 16 * d->reserved, 2 * J, stream);
  FVTA_CHECK_LAUNCH("lstm_step_bwd");
  const int MM = in + dd + 1, N4 = 4 * dd;
  const dim3 wgrid((MM + MmaSq::BM - 1) / MmaSq::BM, N4 / MmaSq::BN, 2 * w.nsplit);
  fvta_prof_begin(FVTA_PROF_LSTM_DW + 16 * d->reserved, stream);
  if (dbg & 1024) {
  } else if (bf)
    launch_dw_bf16(w, stream);
  else
    hipLaunchKernelGGL(lstm_dw_f32, wgrid, dim3(256), sh, stream, w);
  FVTA_CHECK_LAUNCH("lstm_dw");
  const size_t slab_elems = bf ? (size_t)kpad8(d) * N4 : (size_t)MM * N4;
  const unsigned rgrid = (unsigned)((slab_elems + 255) / 256);
  if (bf) {  // slabs are in the engine's internal row order: x rows, ones row (dbias), zero pad, h rows
    const int ndir = d->share_fw_bw ? 1 : 2;
    for (int i = 0; i < ndir; ++i)
      launch_dw_reduce_bf16(wv.slabs + (size_t)i * w.nsplit * slab_elems, d->share_fw_bw ? 2 * w.nsplit : w.nsplit,
                            in, in_internal(d), dd, i == 0 ? dkernel_fw : dkernel_bw, i == 0 ? dbias_fw : dbias_bw,
                            stream);
  } else if (d->share_fw_bw) {
    hipLaunchKernelGGL(lstm_dw_reduce, dim3(rgrid), dim3(256), 0, stream, wv.slabs, 2 * w.nsplit, slab_elems,
                       in + dd, N4, dkernel_fw, dbias_fw);
  } else {
    hipLaunchKernelGGL(lstm_dw_reduce, dim3(rgrid), dim3(256), 0, stream, wv.slabs, w.nsplit, slab_elems, in + dd,
                       N4, dkernel_fw, dbias_fw);
    hipLaunchKernelGGL(lstm_dw_reduce, dim3(rgrid), dim3(256), 0, stream, wv.slabs + (size_t)w.nsplit * slab_elems,
                       w.nsplit, slab_elems, in + dd, N4, dkernel_bw, dbias_bw);
  }
  fvta_prof_end(FVTA_PROF_LSTM_DW + 16 * d->reserved, 1, stream);
  FVTA_CHECK_LAUNCH("lstm_dw_reduce");
  return FVTA_OK;
}

// ---- the bf16 shadow rows of the output (fvta_lstm_shadow_rows / fvta_rows_from_shadow) --------------------------------
// table[dir][row] = address of the d bf16 values hs[dir][t][i] that ARE the output half-row (dir) of arena row `row`, for
// every active (dir, t, sorted i) whose output row lies below nrows; other entries are left as the caller initialised them
__global__ __launch_bounds__(256) void shadow_rows_kernel(PlanView v, const bf16_t* __restrict__ hs, int B, int J, int d,
                                                          int64_t nrows, unsigned long long* __restrict__ table) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (dir, t, i)
  if (idx >= (size_t)2 * J * B) return;
  const int64_t oo = v.oo[idx];
  if (oo < 0) return;
  const int dir = (int)(idx / ((size_t)J * B));
  const int64_t row = oo / v.hdr->out_ld;
  if (row >= nrows) return;
  table[(size_t)dir * nrows + row] = (unsigned long long)reinterpret_cast<uintptr_t>(hs + idx * (size_t)d);
}
extern "C" int fvta_lstm_shadow_rows(const fvta_lstm_desc* d, const void* plan, const void* saved, int64_t nrows,
                                     uint64_t* table, fvta_stream_t stream) {
  if (int e = check_lstm_desc(d)) return e;
  FVTA_CHECK_ARG(plan && saved && table && nrows > 0, "lstm_shadow_rows: null pointer / no rows");
  FVTA_CHECK_ARG(d->precision == FVTA_BF16, "lstm_shadow_rows: the bf16 engine's shadow rows only");
  PlanView pv = plan_view(d, const_cast<void*>(plan));
  SavedView sv = saved_view(d, const_cast<void*>(saved));
  const size_t n = (size_t)2 * d->J * d->B;
  hipLaunchKernelGGL(shadow_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pv, sv.hs, d->B,
                     d->J, d->d, nrows, reinterpret_cast<unsigned long long*>(table));
  FVTA_CHECK_LAUNCH("lstm_shadow_rows");
  return FVTA_OK;
}
// out[row][dir * d + c] = float(table[dir][row][c]) -- the fp32 rows an inspection output wants (vis tensors, tests)
__global__ __launch_bounds__(256) void rows_from_shadow_kernel(const unsigned long long* __restrict__ table, int64_t nrows, int d,
                                                               int64_t out_ld, float* __restrict__ out) {
  const int64_t row = blockIdx.x;
  for (int c = threadIdx.x; c < 2 * d; c += blockDim.x) {
    const int dir = c / d;
    const bf16_t __attribute__((address_space(1)))* p = (const bf16_t __attribute__((address_space(1)))*)table[(size_t)dir * nrows + row];  // (global, not flat)
    out[row * out_ld + c] = bf2f(p[c - dir * d]);
  }
}
extern "C" int fvta_rows_from_shadow(const uint64_t* table, int64_t nrows, int32_t d, int64_t out_ld, float* out,
                                     fvta_stream_t stream) {
  FVTA_CHECK_ARG(table && out && nrows > 0 && d > 0 && out_ld >= 2 * d, "rows_from_shadow: bad arguments");
  hipLaunchKernelGGL(rows_from_shadow_kernel, dim3((unsigned)nrows), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const unsigned long long*>(table), nrows, d, out_ld, out);
  FVTA_CHECK_LAUNCH("rows_from_shadow");
  return FVTA_OK;
}

extern "C" int fvta_lstm_last_state(const fvta_lstm_desc* d, const void* plan, const float* out, int32_t s0,
                                    int32_t count, float* dst, fvta_stream_t stream) {
  if (int e = check_lstm_desc(d)) return e;
  FVTA_CHECK_ARG(plan && out && dst && s0 >= 0 && count > 0 && s0 + count <= d->B, "lstm_last_state: bad range");
  PlanView pv = plan_view(d, const_cast<void*>(plan));
  hipLaunchKernelGGL(last_state_kernel, dim3(count), dim3(256), 0, (hipStream_t)stream, pv, out, s0, count, d->d,
                     dst, (const float*)nullptr, (float*)nullptr);
  FVTA_CHECK_LAUNCH("last_state");
  return FVTA_OK;
}

extern "C" int fvta_lstm_last_state_bwd(const fvta_lstm_desc* d, const void* plan, const float* d_dst, int32_t s0,
                                        int32_t count, float* d_out, fvta_stream_t stream) {
  if (int e = check_lstm_desc(d)) return e;
  FVTA_CHECK_ARG(plan && d_dst && d_out && s0 >= 0 && count > 0 && s0 + count <= d->B,
                 "lstm_last_state_bwd: bad range");
  PlanView pv = plan_view(d, const_cast<void*>(plan));
  hipLaunchKernelGGL(last_state_kernel, dim3(count), dim3(256), 0, (hipStream_t)stream, pv, (const float*)nullptr,
                     s0, count, d->d, (float*)nullptr, d_dst, d_out);
  FVTA_CHECK_LAUNCH("last_state_bwd");
  return FVTA_OK;
}

extern "C" int fvta_test_gemm(int32_t precision, int32_t layout, int32_t M, int32_t N, int32_t K, const float* A,
                              const float* B, float* C, fvta_stream_t stream_) {
  if (precision == FVTA_BF16) {
    FVTA_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 8 == 0 && M % 8 == 0,
                   "test_gemm(bf16): M,N,K must be positive multiples of 8");
    const int e = test_gemm_bf16(layout, M, N, K, A, B, C, (hipStream_t)stream_);
    if (e) {
      fvta_set_error("test_gemm(bf16): layout %d not built", layout);
      return e;
    }
    FVTA_CHECK_LAUNCH("test_gemm_bf16");
    return FVTA_OK;
  }
  FVTA_CHECK_ARG(precision == FVTA_F32, "test_gemm: precision %d not built", precision);
  FVTA_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 4 == 0 && N % 4 == 0 && M % 4 == 0,
                 "test_gemm: M,N,K must be positive multiples of 4");
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid((M + MmaSq::BM - 1) / MmaSq::BM, (N + MmaSq::BN - 1) / MmaSq::BN);
  const size_t sh = MmaSq::LDS_FLOATS * sizeof(float);
  if (layout == 0)
    hipLaunchKernelGGL(test_gemm_f32<0>, grid, dim3(256), sh, stream, M, N, K, A, B, C);
  else if (layout == 1)
    hipLaunchKernelGGL(test_gemm_f32<1>, grid, dim3(256), sh, stream, M, N, K, A, B, C);
  else if (layout == 2)
    hipLaunchKernelGGL(test_gemm_f32<2>, grid, dim3(256), sh, stream, M, N, K, A, B, C);
  else
    FVTA_CHECK_ARG(false, "test_gemm: layout %d", layout);
  FVTA_CHECK_LAUNCH("test_gemm");
  return FVTA_OK;
}
