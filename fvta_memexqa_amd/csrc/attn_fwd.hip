// Focal attention forward for gfx950: model_v2.py:210-298 (attention_3d) and
// 125-201 (attention, K = 1).
//
// The reference materialises tile(h) / tile(q) / concat as [N,K,T,JQ,2w] and
// runs a 1-column matmul over it.  Here the logits are the bilinear form of
// SURVEY.md 3.5, so one workgroup streams 32 context rows at a time ONCE from
// HBM into registers, stages them through LDS as the A operand of an exact-fp32
// MFMA (v_mfma_f32_32x32x2_f32) against the pre-scaled question (B operand
// straight from L2), takes max_j, and folds the rows into an online softmax /
// weighted-sum accumulator from the same registers.  Masked rows are never read.
#include "attn_common.h"
#include "attn_fwd_shared.h"
#include "fvta_prof.h"

namespace fvta {

__device__ __forceinline__ f32x4 ld4g(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
#ifndef FVTA_ATTN_WAVE16_DEFAULT
#define FVTA_ATTN_WAVE16_DEFAULT 3
#endif
// s_waitcnt vmcnt(n) for a loop-unrolled n (the switch folds after unrolling)
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// ---- per-channel vectors from att_logits/W (model_v2.py:242-248, model.py:146-151)
__global__ void attn_vecs_kernel(const float* __restrict__ W, int w, int simi, int feat_order,
                                 float* __restrict__ vecs) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= w) return;
  float U = 0, Rh = 0, R2 = 0, Cq = 0, C2 = 0;
  if (simi == 1) {  // [h, q, h*q]
    Rh = W[c]; Cq = W[w + c]; U = W[2 * w + c];
  } else if (simi == 2) {  // v2: [h*q, (h-q)^2]; model.py: [(h-q)^2, h*q]
    const float W1 = feat_order == 0 ? W[c] : W[w + c];
    const float W2 = feat_order == 0 ? W[w + c] : W[c];
    U = W1 - 2.f * W2; R2 = W2; C2 = W2;
  } else if (simi == 3) {  // [h, q, (h-q)^2, h*q]
    Rh = W[c]; Cq = W[w + c];
    const float W2 = W[2 * w + c];
    U = W[3 * w + c] - 2.f * W2; R2 = W2; C2 = W2;
  } else {  // cosine: row "term" slot carries sum h^2 for the norm
    U = 1.f; R2 = 1.f;
  }
  vecs[VEC_U * w + c] = U;
  vecs[VEC_RH * w + c] = Rh;
  vecs[VEC_R2 * w + c] = R2;
  vecs[VEC_CQ * w + c] = Cq;
  vecs[VEC_C2 * w + c] = C2;
}

// ---- question side: Qs [N][W4][JP][4], ct [N][JP], valid-j bits.  grid N, 256 threads
// qh_wide: the fp16 split in the fragment order of attn_fwd_wide (attn_fwd_wide.hip) instead of [2][W4][JP][4]:
// [wave v = c / (w/8)][stage = (j / 32) (w/256) + k-step][fragment = 2 ((j / 16) & 1) + {hi, lo}][lane = j % 16 + 16 kq][8],
// the lane's 8 channels being 32 ks + 4 kq + (0..3) and 32 ks + 16 + 4 kq + (0..3) of the wave's w/8
__global__ __launch_bounds__(256) void attn_prep_q_kernel(AttnShape s, AttnSaved sv, const float* __restrict__ hq,
                                                          const uint8_t* __restrict__ qmask,
                                                          const float* __restrict__ bptr, int qh_wide) {
  __shared__ float s_rq[64];
  const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int w = s.w;
  const float* q = hq + (size_t)n * s.JQ * w;
  const float* Uv = sv.vecs + VEC_U * w;
  const float* Cq = sv.vecs + VEC_CQ * w;
  const float* C2 = sv.vecs + VEC_C2 * w;
  const float bias = (s.simi == 4 || bptr == nullptr) ? 0.f : bptr[0];
  // grid (N, slices): the question norms are needed by every slice (cosine only: every slice walks every j and slice 0
  // writes ct); otherwise the slices share the positions (ct of position j by slice (j / 4) % slices: one position per wave
  // and round -- slice 0 alone walked all JP positions, 8 per wave, while seven slices waited at the barrier: 45 us)
  const bool every = s.simi == 4;
  for (int j = (every ? 0 : 4 * (int)blockIdx.y) + wave; j < s.JP; j += every ? 4 : 4 * (int)gridDim.y) {
    float s1 = 0.f, s2 = 0.f;
    if (j < s.JQ)
      for (int c = lane; c < w; c += 64) {
        const float v = q[(size_t)j * w + c];
        s1 += Cq[c] * v + C2[c] * v * v;
        s2 += v * v;
      }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
      if (blockIdx.y == 0 || !every) sv.ct[(size_t)n * s.JP + j] = (j < s.JQ && s.simi != 4) ? s1 + bias : 0.f;
      s_rq[j] = rsqrtf(fmaxf(s2, 1e-12f));  // tf.nn.l2_normalize eps (read back under the cosine similarity only)
    }
  }
  if (tid == 0 && blockIdx.y == 0) {
    uint64_t bits = 0;
    for (int j = 0; j < s.JQ; ++j)
      if (!s.use_mask || qmask[(size_t)n * s.JQ + j]) bits |= (1ull << j);
    sv.qvalid[(size_t)n * 2] = bits;
  }
  __syncthreads();
  float* Qs = sv.Qs + (size_t)n * s.W4 * s.JP * 4;
  for (int u = blockIdx.y * 256 + tid; u < s.W4 * s.JP; u += 256 * gridDim.y) {
    const int c4 = u / s.JP, j = u % s.JP;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (j < s.JQ) {
      const f32x4 qv = ld4g(q + (size_t)j * w + 4 * c4);
      if (s.simi == 4) {
        o = qv * s_rq[j];
      } else {
        const f32x4 uv = ld4g(Uv + 4 * c4);
        o = qv * uv;
      }
    }
    *reinterpret_cast<f32x4*>(Qs + (size_t)u * 4) = o;
    // fp16 split of the same values for the 16-row forward kernel (see attn_fwd_rows16)
    uint16_t* Qh = sv.Qh + (size_t)n * 2 * s.W4 * s.JP * 4;
    half2v h0, l0, h1, l1;
    split_f16x2(o[0], o[1], h0, l0);
    split_f16x2(o[2], o[3], h1, l1);
    if (qh_wide) {
      const int cw = w / 8, c = 4 * c4, v = c / cw, cin = c % cw, ks = cin / 32, r0 = cin % 32;
      const int kq = (r0 % 16) / 4, e0 = r0 < 16 ? 0 : 4, jt = j / 16, nst = (cw / 32) * (s.JP / 32);
      const size_t frag = (((size_t)v * nst + (size_t)(jt / 2) * (cw / 32) + ks) * 4 + 2 * (jt & 1)) * 512 + (size_t)(j % 16 + 16 * kq) * 8 + e0;
      *reinterpret_cast<half4v*>(Qh + frag) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
      *reinterpret_cast<half4v*>(Qh + frag + 512) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
      continue;
    }
    *reinterpret_cast<half4v*>(Qh + (size_t)u * 4) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
    *reinterpret_cast<half4v*>(Qh + ((size_t)s.W4 * s.JP + u) * 4) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
  }
}

// ---- valid-row lists.  grid N*K, 256 threads.  Also resets amax/jmax of every t.
__global__ __launch_bounds__(256) void attn_compact_kernel(AttnShape s, AttnSaved sv,
                                                           const uint8_t* __restrict__ hmask) {
  __shared__ int s_wcnt[4];
  __shared__ int s_base;
  const int nk = blockIdx.x, n = nk / s.K, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int T = s.T;
  int32_t* idx = sv.idx + (size_t)nk * T;
  float* amax = sv.amax + (size_t)nk * T;
  uint8_t* jmax = sv.jmax + (size_t)nk * T;
  const bool qany = sv.qvalid[(size_t)n * 2] != 0;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + tid;
    bool valid = false;
    if (t < T) {
      valid = s.use_mask ? (hmask[(size_t)nk * T + t] != 0) : true;
      amax[t] = FVTA_NEG;
      jmax[t] = 0;
    }
    const unsigned long long b = __ballot(valid);
    if (lane == 0) s_wcnt[wave] = __popcll(b);
    __syncthreads();
    int off = s_base;
    for (int v = 0; v < wave; ++v) off += s_wcnt[v];
    if (valid) idx[off + __popcll(b & ((1ull << lane) - 1ull))] = t;
    __syncthreads();
    if (tid == 0) s_base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
  const int cnt = s_base;
  const bool allm = s.use_mask && (cnt == 0 || !qany);
  if (allm)
    for (int t = tid; t < T; t += 256) idx[t] = t;  // softmax goes uniform over ALL T
  if (tid == 0) {
    sv.cnt[nk] = allm ? T : cnt;
    sv.allmasked[nk] = allm ? 1 : 0;
  }
}

__global__ void fill_kernel(float* __restrict__ p, size_t n, float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- workgroups per n in proportion to the n's valid 16-row tiles (pair kernel, N <= 64).  One wave: lane n holds n's
// tile count; every n gets floor(share) >= 1 workgroups (at most maxg), the remainder goes one at a time to the n with
// the most tiles per workgroup.  tab[wg] = n | g << 16 | G_n << 24 in n order (an n's workgroups stay XCD-contiguous),
// 0xffffffff for workgroups left over.
// run_cap > 0 (attn_fwd_wide, whose workgroup lists its run's 32-row tiles in LDS): if some album's share would make a
// workgroup's run longer than run_cap tiles (a skewed batch near the limit: an album may get fewer workgroups than nwg / N),
// the table is the plain deal instead -- n = wg / G, every album G = nwg / N workgroups -- which the host's bound
// (wide_covers) covers; the kernel never sees a run it cannot hold.
__global__ __launch_bounds__(64) void attn_balance_kernel(AttnShape s, AttnSaved sv, int nwg, int maxg, uint32_t* __restrict__ tab,
                                                          int run_cap) {
  const int lane = threadIdx.x;
  int tiles = 0, tiles32 = 0;
  if (lane < s.N)
    for (int k = 0; k < s.K; ++k) {
      tiles += (sv.cnt[lane * s.K + k] + 15) >> 4;
      tiles32 += (sv.cnt[lane * s.K + k] + 31) >> 5;
    }
  if (lane < s.N && tiles < 1) tiles = 1;
  int total = tiles;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
  int G = 0;
  if (lane < s.N) {
    G = (int)((long long)tiles * nwg / total);
    G = G < 1 ? 1 : (G > maxg ? maxg : G);
  }
  int used = G;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) used += __shfl_xor(used, o, 64);
  for (int it = 0; it < 256 && used != nwg; ++it) {
    // load per workgroup of each n (scaled); give to the most loaded (used < nwg) or take from the least (used > nwg)
    const bool give = used < nwg;
    const bool can = lane < s.N && (give ? G < maxg : G > 1);
    long long key = can ? ((long long)tiles << 20) / (give ? G : G - 1) : (give ? -1 : (1ll << 62));
    long long best = key;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const long long other = __shfl_xor(best, o, 64);
      best = give ? (other > best ? other : best) : (other < best ? other : best);
    }
    if (give ? best < 0 : best == (1ll << 62)) break;  // nobody can take / give
    const unsigned long long who = __ballot(can && key == best);
    if (lane == __ffsll((long long)who) - 1) G += give ? 1 : -1;
    used += give ? 1 : -1;
  }
  if (run_cap > 0) {  // (a run crosses at most K stream boundaries: one ragged tile each)
    const bool over = lane < s.N && (tiles32 + G - 1) / G + s.K + 1 > run_cap;
    if (__ballot(over) != 0ull) {
      const int Gall = nwg / s.N;
      for (int i = lane; i < nwg; i += 64) tab[i] = (uint32_t)(i / Gall) | ((uint32_t)(i % Gall) << 16) | ((uint32_t)Gall << 24);
      return;
    }
  }
  int first = G;  // exclusive prefix sum over n
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(first, o, 64);
    if (lane >= o) first += v;
  }
  first -= G;
  if (lane < s.N)
    for (int g = 0; g < G; ++g) tab[first + g] = (uint32_t)lane | ((uint32_t)g << 16) | ((uint32_t)G << 24);
  const int tot = __shfl(first + G, 63, 64);
  for (int i = tot + lane; i < nwg; i += 64) tab[i] = 0xffffffffu;
}

// ---- main kernel -----------------------------------------------------------

// w = 4 * SCW * NSC * NW * NSLAB.  A workgroup is NW waves; a wave owns NSC
// sub-chunks of SCW float4 columns per slab; lane = (c4l = lane % SCW,
// rg = lane / SCW) holds rows rg + RGN*p.  The 32 x (w / NSLAB) row slab lives in
// registers (NSC * P float4 per lane) between the score pass and the weighted
// sum, so NSC * P is kept <= 16: wide rows use 8 waves, not more sub-chunks.
// TW: time_warp_att (the softmax over t runs on amax * tscale[n,t]) -- a compile-time flag: the widest shapes sit at
// the 256-VGPR limit and must not pay for it when it is off.
template <int SCW, int NSC, int NSLAB, int JT, int NW, bool TW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void attn_fwd_main(AttnFwdArgs a) {
  constexpr int NT = NW * 64;
  constexpr int RGN = 64 / SCW;
  constexpr int P = 32 / RGN;
  constexpr int LDW = 4 * SCW + 4;
  constexpr int SLAB4 = NW * NSC * SCW;
  static_assert(NSC * P <= 16, "row slab must fit the register file without scratch");
  constexpr bool COUNTED = !(NW == 8 && JT == 2);
  __shared__ __attribute__((aligned(16))) float s_stage[NW][32 * LDW];
  // per wave: the pre-scaled question fragments of the current sub-chunk, DMA'd from L2 ([s4][jt][lane][4],
  // the exact order the MFMA B operands are read in); the same space later carries the wave's partial scores
  constexpr int BQ = ((SCW / 2) * JT * 256 > JT * 1024) ? (SCW / 2) * JT * 256 : JT * 1024;
  __shared__ __attribute__((aligned(16))) float s_bq[NW][BQ];
  // row-term vectors Rh, R2 of the bilinear form: LDS copies, so that no ordinary global load sits between
  // the fragment DMAs and their counted waits (vmcnt retires in order)
  __shared__ __attribute__((aligned(16))) float s_vec[2][4 * SLAB4 * NSLAB];
  __shared__ float s_rt[NW][32];
  __shared__ int s_t[32];
  __shared__ float s_amax[32];  // the inner softmax's logits z of the tile's rows
  __shared__ float s_amu[32];   // amax itself (differs from z under time_warp_att)
  __shared__ float s_mu_run;    // running max of amax over the workgroup's rows
  __shared__ float s_p[32];

  const AttnShape& s = a.s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c4l = lane % SCW, rg = lane / SCW;
  const int l31 = lane & 31, hf = lane >> 5;
  // XCD-aware work mapping: blocks are dealt round-robin over the 8 XCDs, so block b and b+8 share an L2.
  // Give each XCD a contiguous range of (n,k,split) items: the pre-scaled question of one n (w*32*4 B) is then
  // re-read from ONE L2 by all K*nsplit workgroups of that n instead of thrashing all eight.
  const int nitems = s.N * s.K * s.nsplit;
  const int bid = blockIdx.x;
  const int per = (nitems + 7) / 8;
  const int item = (bid & 7) * per + (bid >> 3);
  if (item >= nitems || (bid >> 3) >= per) return;
  const int nk = item / s.nsplit, n = nk / s.K, split = item % s.nsplit;
  const int T = s.T, w = s.w, JP = s.JP;
  const int cnt = a.sv.cnt[nk];
  const bool allm = a.sv.allmasked[nk] != 0;
  const int tiles_total = (cnt + 31) >> 5;
  const int tiles_per = (tiles_total + s.nsplit - 1) / s.nsplit;
  const int tile0 = split * tiles_per, tile1 = min(tiles_total, tile0 + tiles_per);
  float* part = a.part + ((size_t)nk * s.nsplit + split) * (w + 4);
  if (tile0 >= tile1) {
    if (tid == 0) {
      part[0] = -INFINITY;
      part[1] = 0.f;
      part[2] = -INFINITY;
    }
    return;
  }
  const float* __restrict__ hbase = a.hinfo + (size_t)nk * a.hstride;
  const int32_t* __restrict__ idx = a.sv.idx + (size_t)nk * T;
  const float* __restrict__ Qs = a.sv.Qs + (size_t)n * s.W4 * JP * 4;
  const float* __restrict__ ct = a.sv.ct + (size_t)n * JP;
  const float* __restrict__ vRh = a.sv.vecs + VEC_RH * w;
  const float* __restrict__ vR2 = a.sv.vecs + VEC_R2 * w;
  const uint64_t qvalid = a.sv.qvalid[(size_t)n * 2];
  const bool cosine = s.simi == 4;

  f32x4 hreg[NSC][P];
  f32x4 uacc[NSLAB][NSC];
#pragma unroll
  for (int sl = 0; sl < NSLAB; ++sl)
#pragma unroll
    for (int sc = 0; sc < NSC; ++sc) uacc[sl][sc] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  if (TW && tid == 0) s_mu_run = -INFINITY;  // (thread 0 alone keeps it: no register per lane)
  float* stage = s_stage[wave];
  float* bq = s_bq[wave];
  const __amdgpu_buffer_rsrc_t rq =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Qs), 0, (unsigned)(s.W4 * JP * 16), 0x00020000);
  // direct-to-LDS copy of the question fragments of sub-chunk c4base (asynchronous; no VGPRs, no L2 latency
  // inside the MFMA chain)
  auto issue_b1 = [&](int c4base, int s4) {  // the JT blocks of MFMA group s4
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void*)(bq + (s4 * JT + jt) * 256), 16,
                                               (unsigned)(((c4base + 2 * s4 + hf) * JP + jt * 32 + l31) * 16), 0, 0, 0);
  };
  auto issue_b = [&](int c4base) {
#pragma unroll
    for (int s4 = 0; s4 < SCW / 2; ++s4) issue_b1(c4base, s4);
  };
  for (int c = tid; c < w; c += NT) {
    s_vec[0][c] = vRh[c];
    s_vec[1][c] = vR2[c];
  }

  auto load_slab = [&](int sl) {
#pragma unroll
    for (int sc = 0; sc < NSC; ++sc)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int t = s_t[rg + RGN * p];
        const int c4 = sl * SLAB4 + wave * (NSC * SCW) + sc * SCW + c4l;
        // unconditional load from a clamped row, zeroed afterwards: a load under a per-lane branch would be
        // waited for one at a time
        const f32x4 v = ld4g(hbase + (size_t)max(t, 0) * w + 4 * c4);
        hreg[sc][p] = t >= 0 ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };

  for (int tile = tile0; tile < tile1; ++tile) {
    __syncthreads();  // previous tile's readers of s_t / s_p are done
    if (tid < 32) {
      const int r = tile * 32 + tid;
      s_t[tid] = r < cnt ? idx[r] : -1;
    }
    __syncthreads();
    if (!allm) {
      f32x16 acc[JT];
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
      float rtp[P];
#pragma unroll
      for (int p = 0; p < P; ++p) rtp[p] = 0.f;
      issue_b(wave * (NSC * SCW));
#pragma unroll 1
      for (int sl = 0; sl < NSLAB; ++sl) {
        load_slab(sl);
#pragma unroll
        for (int sc = 0; sc < NSC; ++sc) {
          const int c4base = sl * SLAB4 + wave * (NSC * SCW) + sc * SCW;
          const f32x4 rh4 = *reinterpret_cast<const f32x4*>(&s_vec[0][4 * (c4base + c4l)]);
          const f32x4 r24 = *reinterpret_cast<const f32x4*>(&s_vec[1][4 * (c4base + c4l)]);
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const f32x4 h = hreg[sc][p];
            const f32x4 tmp = h * (rh4 + r24 * h);
            rtp[p] += (tmp[0] + tmp[1]) + (tmp[2] + tmp[3]);
            *reinterpret_cast<f32x4*>(&stage[(rg + RGN * p) * LDW + 4 * c4l]) = h;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          // sub-chunk 0 of a slab: its fragments were requested before the slab's row loads, which the
          // row-term math above has already waited for (in-order retirement) -- nothing left in flight.
          // later sub-chunks: their fragments were requested block by block during the previous MFMA
          // chain; wait per block with a counted vmcnt.
          if (sc == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          const bool has_next = (sc + 1 < NSC) || (sl + 1 < NSLAB);
          const int next_base = (sc + 1 < NSC) ? c4base + SCW : (sl + 1) * SLAB4 + wave * (NSC * SCW);
#pragma unroll
          for (int s4 = 0; s4 < SCW / 2; ++s4) {
            // younger than group s4's blocks: the rest of this sub-chunk's (7 - s4 groups) plus, when refilling,
            // the s4 groups already re-requested for the next sub-chunk
            // (the counted form assumes the DMAs are the only VMEM traffic of the chain, i.e. a kernel
            // without scratch; the one shape at the 256-VGPR limit drains everything at group 0 instead)
            if (sc != 0) {
              if (COUNTED) wait_vmcnt_upto((has_next ? SCW / 2 - 1 : SCW / 2 - 1 - s4) * JT);
              else if (s4 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const f32x4 av = *reinterpret_cast<const f32x4*>(&stage[l31 * LDW + 8 * s4 + 4 * hf]);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(&bq[(s4 * JT + jt) * 256 + lane * 4]);
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[jt], 0, 0, 0);
            }
            // refill group s4's blocks with the next sub-chunk's, so the copy runs under the rest of this
            // chain.  The explicit lgkmcnt(0) is load-bearing: the MFMAs above only need the fragments at
            // execution, so without it the compiler issues the DMA right behind the ds_reads, and under LDS
            // queueing the DMA write can overtake them (write-after-read on the block).
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (has_next) issue_b1(next_base, s4);
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      // partial scores of this wave's K-slice -> LDS
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) bq[jt * 1024 + r * 64 + lane] = acc[jt][r];
      // row terms: reduce over the SCW lanes that share a row
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float v = rtp[p];
#pragma unroll
        for (int o = 1; o < SCW; o <<= 1) v += __shfl_xor(v, o, 64);
        if (c4l == 0) s_rt[wave][rg + RGN * p] = v;
      }
      __syncthreads();
      if (tid < 256) {
        const int row = tid >> 3, jg = tid & 7;
        const int t = s_t[row];
        float rt = (s_rt[0][row] + s_rt[1][row]) + (s_rt[2][row] + s_rt[3][row]);
        if (NW == 8) rt += (s_rt[4][row] + s_rt[5][row]) + (s_rt[6][row] + s_rt[7][row]);
        const float rs = cosine ? rsqrtf(fmaxf(rt, 1e-12f)) : 1.f;
        const int preg = (row & 3) + 4 * (row >> 3), phf = (row >> 2) & 1;
        float best = -INFINITY;
        int bestj = 0;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int jl = jg * 4 + e, j = jt * 32 + jl;
            const int pi = preg * 64 + phf * 32 + jl;
            const int pj = jt * 1024 + pi;
            float x = (s_bq[0][pj] + s_bq[1][pj]) + (s_bq[2][pj] + s_bq[3][pj]);
            if (NW == 8) x += (s_bq[4][pj] + s_bq[5][pj]) + (s_bq[6][pj] + s_bq[7][pj]);
            x = cosine ? x * rs : x + rt + ct[j];
            const bool valid = (qvalid >> j) & 1ull;
            if (a.a_logits && t >= 0 && j < s.JQ)
              a.a_logits[((size_t)nk * T + t) * s.JQ + j] = valid ? (s.add_tanh ? tanhf(x) : x) : FVTA_NEG;
            if (valid && x > best) {
              best = x;
              bestj = j;
            }
          }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
          const float ob = __shfl_xor(best, o, 64);
          const int oj = __shfl_xor(bestj, o, 64);
          if (ob > best || (ob == best && oj < bestj)) {
            best = ob;
            bestj = oj;
          }
        }
        if (jg == 0) {
          if (t >= 0) {
            const float av = s.add_tanh ? tanhf(best) : best;
            s_amax[row] = TW ? tw_logit(av, a.tscale[(size_t)n * T + t]) : av;
            if (TW) s_amu[row] = av;
            a.sv.amax[(size_t)nk * T + t] = av;
            a.sv.jmax[(size_t)nk * T + t] = (uint8_t)bestj;
          } else {
            s_amax[row] = -INFINITY;
            if (TW) s_amu[row] = -INFINITY;
          }
        }
      }
    } else {
      if (tid < 32) {
        const int t = s_t[tid];
        s_amax[tid] = t >= 0 ? (TW ? tw_logit(FVTA_NEG, a.tscale[(size_t)n * T + t]) : FVTA_NEG) : -INFINITY;
        if (TW) s_amu[tid] = t >= 0 ? FVTA_NEG : -INFINITY;
      }
      load_slab(NSLAB - 1);
    }
    __syncthreads();
    // online softmax over t (softsel inner, model_v2.py:278)
    float mt = s_amax[0];
#pragma unroll
    for (int r = 1; r < 32; ++r) mt = fmaxf(mt, s_amax[r]);
    if (TW && tid == 0) {
      float mut = s_mu_run;
      for (int r = 0; r < 32; ++r) mut = fmaxf(mut, s_amu[r]);
      s_mu_run = mut;
    }
    const float m_new = fmaxf(m_run, mt);
    const float scale = expf(m_run - m_new);
    if (tid < 32) s_p[tid] = expf(s_amax[tid] - m_new);
    __syncthreads();
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) lsum += s_p[r];
    l_run = l_run * scale + lsum;
    m_run = m_new;
    float pr[P];
#pragma unroll
    for (int p = 0; p < P; ++p) pr[p] = s_p[rg + RGN * p];
#pragma unroll
    for (int k = 0; k < NSLAB; ++k) {
      const int sl = NSLAB - 1 - k;  // the last slab is still in registers
      if (k > 0) load_slab(sl);
#pragma unroll
      for (int sc = 0; sc < NSC; ++sc) {
        f32x4 u = uacc[sl][sc] * scale;
#pragma unroll
        for (int p = 0; p < P; ++p) u += hreg[sc][p] * pr[p];
        uacc[sl][sc] = u;
      }
    }
  }
  // fold the RGN row groups of each wave, store the partial (m, l, u)
#pragma unroll
  for (int sl = 0; sl < NSLAB; ++sl)
#pragma unroll
    for (int sc = 0; sc < NSC; ++sc) {
      f32x4 u = uacc[sl][sc];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = u[e];
#pragma unroll
        for (int o = SCW; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        u[e] = v;
      }
      if (rg == 0) {
        const int c4 = sl * SLAB4 + wave * (NSC * SCW) + sc * SCW + c4l;
        *reinterpret_cast<f32x4*>(part + 4 + 4 * c4) = u;
      }
    }
  if (tid == 0) {
    part[0] = m_run;
    part[1] = l_run;
    part[2] = TW ? s_mu_run : m_run;
  }
}

// ---- main kernel, 16-row tiles (JQ <= 32, 128 <= w <= 1024) -------------------
// The shape the metric runs.  Differences from the kernel above, all aimed at keeping HBM busy:
//  * the pre-scaled question slice of a wave (w/NW channels x 32 j) only changes with n, so it lives in
//    registers as the MFMA B operand -- no per-tile question traffic;
//  * context rows are loaded global -> VGPR directly in the lane order of the MFMA A operand (lane = row l15,
//    k quarter kq: 4 consecutive channels of its row per 16-channel block), so they cross neither LDS nor a
//    transposition; the same registers are the operand of the weighted sum, so a row is read once.  The
//    registers are double buffered (the tile loop is unrolled by two) and the next tile's loads are issued
//    before the current tile is touched -- the compiler's own counted vmcnt does the rest.
//    (A global -> LDS DMA version of this kernel spent its time in the LDS: the copy's LDS write path plus the
//    read-back saturated it at ~40% of the HBM roofline.)
//  * a workgroup owns `ipw` consecutive (n,k,split) items and runs their tiles as ONE stream: the prefetch
//    crosses item boundaries, so the short items of the metric shape (150 rows) do not pay a pipeline fill each;
//  * the fp32 matrix pipe (64 flop/clk/SIMD) would cost as many cycles per tile as HBM takes to deliver it, so the
//    dot products run on the fp16 pipe as a 3-term split: x = hi + lo with hi = rtz_f16(x) (11 bits, the
//    remainder x - hi is exact in fp32) and lo = rtz_f16((x - hi) * 2^11) (the scaling keeps lo out of the fp16
//    subnormals);  h.q = hi.hi + 2^-11 (hi.lo + lo.hi) with fp32 accumulation, every product exact.  What is
//    dropped is lo.lo and the bits below 22: <= 3 * 2^-22 of |h||q| per product, the size of fp32 rounding in
//    the reference's own summation.  Domain: |h|, |U q| < 65504 (encoder outputs are in (-1, 1));
//    FVTA_ATTN_EXACT=1 routes to the fp32-MFMA kernel above;
//  * all in-wave reductions are DPP row operations (no ds_bpermute round trips), and the barriers are LDS-only.
constexpr int R16_CAP = 1600;            // rows of one workgroup's items (LDS row list, 16-bit ids)
constexpr int R16_MAXT = R16_CAP / 16;   // tiles
constexpr int R16_MAXI = 32;             // items

// Row loads the compiler does NOT track.  Its waitcnt pass merges the rarely taken "new n" path (whose loads are
// issued after the prefetch) into the common one and guards every later use of anything loaded there -- and, at the
// loop header, the prefetched rows themselves -- with `s_waitcnt vmcnt(0)`: the prefetch of tile g+2 then has to LAND
// inside tile g, i.e. memory and compute take turns instead of overlapping (measured: the phases add up).  Issued
// through inline asm the loads are invisible to that pass; the kernel waits for exactly the tile it is about to use
// (loads return in order) with counted waits.  Two things the compiler no longer does for us:
//  * it must not MOVE a register that has a load in flight (it happily inserted phi copies of the three buffers at the
//    loop latch): each buffer is pinned to fixed physical registers (A: v160.., B: v192.., C: v224..) both where it is
//    loaded and where it is first used, so there is nothing to shuffle;
//  * its hazard recogniser does not look into inline asm: a VMEM instruction may read an SGPR 5 wait states after a
//    SALU instruction wrote it at the earliest -- and the compiler is free to copy the base pointer into the asm's
//    SGPR operand right in front of it -- so every load carries its own `s_nop 4`.
template <int BUF, int NB, int FROM = 0, int TO = 8>
__device__ __forceinline__ void rows_issue(const float* base, unsigned voff, f32x4 (&d)[NB]) {
  static_assert(NB == 2 || NB == 4 || NB == 8, "pinned register tables below");
  if constexpr (BUF == 0) {
    if (NB > 0 && 0 >= FROM && 0 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:0" : "={v[160:163]}"(d[0 % NB]) : "v"(voff), "s"(base));
    if (NB > 1 && 1 >= FROM && 1 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:64" : "={v[164:167]}"(d[1 % NB]) : "v"(voff), "s"(base));
    if (NB > 2 && 2 >= FROM && 2 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:128" : "={v[168:171]}"(d[2 % NB]) : "v"(voff), "s"(base));
    if (NB > 3 && 3 >= FROM && 3 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:192" : "={v[172:175]}"(d[3 % NB]) : "v"(voff), "s"(base));
    if (NB > 4 && 4 >= FROM && 4 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:256" : "={v[176:179]}"(d[4 % NB]) : "v"(voff), "s"(base));
    if (NB > 5 && 5 >= FROM && 5 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:320" : "={v[180:183]}"(d[5 % NB]) : "v"(voff), "s"(base));
    if (NB > 6 && 6 >= FROM && 6 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:384" : "={v[184:187]}"(d[6 % NB]) : "v"(voff), "s"(base));
    if (NB > 7 && 7 >= FROM && 7 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:448" : "={v[188:191]}"(d[7 % NB]) : "v"(voff), "s"(base));
  } else if constexpr (BUF == 1) {
    if (NB > 0 && 0 >= FROM && 0 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:0" : "={v[192:195]}"(d[0 % NB]) : "v"(voff), "s"(base));
    if (NB > 1 && 1 >= FROM && 1 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:64" : "={v[196:199]}"(d[1 % NB]) : "v"(voff), "s"(base));
    if (NB > 2 && 2 >= FROM && 2 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:128" : "={v[200:203]}"(d[2 % NB]) : "v"(voff), "s"(base));
    if (NB > 3 && 3 >= FROM && 3 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:192" : "={v[204:207]}"(d[3 % NB]) : "v"(voff), "s"(base));
    if (NB > 4 && 4 >= FROM && 4 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:256" : "={v[208:211]}"(d[4 % NB]) : "v"(voff), "s"(base));
    if (NB > 5 && 5 >= FROM && 5 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:320" : "={v[212:215]}"(d[5 % NB]) : "v"(voff), "s"(base));
    if (NB > 6 && 6 >= FROM && 6 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:384" : "={v[216:219]}"(d[6 % NB]) : "v"(voff), "s"(base));
    if (NB > 7 && 7 >= FROM && 7 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:448" : "={v[220:223]}"(d[7 % NB]) : "v"(voff), "s"(base));
  } else {
    if (NB > 0 && 0 >= FROM && 0 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:0" : "={v[224:227]}"(d[0 % NB]) : "v"(voff), "s"(base));
    if (NB > 1 && 1 >= FROM && 1 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:64" : "={v[228:231]}"(d[1 % NB]) : "v"(voff), "s"(base));
    if (NB > 2 && 2 >= FROM && 2 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:128" : "={v[232:235]}"(d[2 % NB]) : "v"(voff), "s"(base));
    if (NB > 3 && 3 >= FROM && 3 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:192" : "={v[236:239]}"(d[3 % NB]) : "v"(voff), "s"(base));
    if (NB > 4 && 4 >= FROM && 4 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:256" : "={v[240:243]}"(d[4 % NB]) : "v"(voff), "s"(base));
    if (NB > 5 && 5 >= FROM && 5 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:320" : "={v[244:247]}"(d[5 % NB]) : "v"(voff), "s"(base));
    if (NB > 6 && 6 >= FROM && 6 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:384" : "={v[248:251]}"(d[6 % NB]) : "v"(voff), "s"(base));
    if (NB > 7 && 7 >= FROM && 7 < TO) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:448" : "={v[252:255]}"(d[7 % NB]) : "v"(voff), "s"(base));
  }
}
// Wait until at most `tiles_younger` * NB loads issued after buffer b's tile are still in flight; every later use of
// b depends on the pinning statement that follows the wait, so nothing can be scheduled above it.
template <int BUF, int NB>
__device__ __forceinline__ void rows_wait(int tiles_younger, f32x4 (&b)[NB]) {
  if (tiles_younger >= 2)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory");
  else if (tiles_younger == 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (BUF == 0) {
    if constexpr (NB == 8) {
      asm volatile("" : "+{v[160:163]}"(b[0]), "+{v[164:167]}"(b[1]), "+{v[168:171]}"(b[2]), "+{v[172:175]}"(b[3]), "+{v[176:179]}"(b[4]), "+{v[180:183]}"(b[5]), "+{v[184:187]}"(b[6]), "+{v[188:191]}"(b[7]));
    }
    else if constexpr (NB == 4) {
      asm volatile("" : "+{v[160:163]}"(b[0]), "+{v[164:167]}"(b[1]), "+{v[168:171]}"(b[2]), "+{v[172:175]}"(b[3]));
    }
    else if constexpr (NB == 2) {
      asm volatile("" : "+{v[160:163]}"(b[0]), "+{v[164:167]}"(b[1]));
    }
  } else if constexpr (BUF == 1) {
    if constexpr (NB == 8) {
      asm volatile("" : "+{v[192:195]}"(b[0]), "+{v[196:199]}"(b[1]), "+{v[200:203]}"(b[2]), "+{v[204:207]}"(b[3]), "+{v[208:211]}"(b[4]), "+{v[212:215]}"(b[5]), "+{v[216:219]}"(b[6]), "+{v[220:223]}"(b[7]));
    }
    else if constexpr (NB == 4) {
      asm volatile("" : "+{v[192:195]}"(b[0]), "+{v[196:199]}"(b[1]), "+{v[200:203]}"(b[2]), "+{v[204:207]}"(b[3]));
    }
    else if constexpr (NB == 2) {
      asm volatile("" : "+{v[192:195]}"(b[0]), "+{v[196:199]}"(b[1]));
    }
  } else {
    if constexpr (NB == 8) {
      asm volatile("" : "+{v[224:227]}"(b[0]), "+{v[228:231]}"(b[1]), "+{v[232:235]}"(b[2]), "+{v[236:239]}"(b[3]), "+{v[240:243]}"(b[4]), "+{v[244:247]}"(b[5]), "+{v[248:251]}"(b[6]), "+{v[252:255]}"(b[7]));
    }
    else if constexpr (NB == 4) {
      asm volatile("" : "+{v[224:227]}"(b[0]), "+{v[228:231]}"(b[1]), "+{v[232:235]}"(b[2]), "+{v[236:239]}"(b[3]));
    }
    else if constexpr (NB == 2) {
      asm volatile("" : "+{v[224:227]}"(b[0]), "+{v[228:231]}"(b[1]));
    }
  }
}
template <int I>
struct BufTag {
  static constexpr int value = I;
};

// Workgroup barrier for LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which on gfx9 is
// s_waitcnt vmcnt(0): it would drain the row loads in flight at each barrier and serialise the prefetch.

// RMODE: which row-term vectors of the bilinear form are non-zero: 1 = Rh (simi 1), 2 = R2 (simi 2 and 4),
// 3 = both (simi 3).
template <int NB, int NW, int RMODE>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 1 : 2) void attn_fwd_rows16(AttnFwdArgs a) {
  static_assert(NB % 2 == 0, "two 16-channel blocks feed one K = 32 MFMA");
  constexpr int NT = NW * 64;
  constexpr int NM = NB / 2;         // MFMAs along K per wave
  constexpr int NPART = NW * 4;      // row-term partials per row
  __shared__ __attribute__((aligned(16))) float s_part[NW][512];  // [jt][r][lane] partial scores
  __shared__ __attribute__((aligned(16))) float s_vec[2][NW * NB * 16];
  __shared__ uint16_t s_idx[R16_CAP];
  __shared__ int s_tnk[R16_MAXT];
  __shared__ uint8_t s_titem[R16_MAXT], s_iallm[R16_MAXI];
  __shared__ int s_ink[R16_MAXI], s_int[R16_MAXI], s_it0[R16_MAXI], s_ifirst[R16_MAXI + 1];
  __shared__ float s_rtp[NPART][16];
  __shared__ float s_ct[32];
  __shared__ float s_cand_v[2][16];  // per j tile: row max of the raw logits and its j
  __shared__ int s_cand_j[2][16];
  // the lo pieces of the question (B operand of the hi.lo term): per wave, in MFMA lane order.  Only the hi
  // pieces fit the register file next to the double-buffered rows; the LDS is otherwise idle in this kernel.
  __shared__ __attribute__((aligned(16))) half8 s_blo[NW][NM * 2][64];
  __shared__ __attribute__((aligned(16))) half8 s_bhi[NW][NM * 2][64];

  const AttnShape& s = a.s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, kq = lane >> 4;
  const int T = s.T, w = s.w, JP = s.JP;
  const int nitems = s.N * s.K * s.nsplit;
  // XCD-contiguous workgroup order (see attn_fwd_main): the workgroups of one n share an L2
  const int nwg = (nitems + a.ipw - 1) / a.ipw;
  const int per = (nwg + 7) / 8;
  const int wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (wg >= nwg || (int)(blockIdx.x >> 3) >= per) return;
  const int item_lo = wg * a.ipw;
  const int nit = min(nitems, item_lo + a.ipw) - item_lo;
  const unsigned long long t_entry = __builtin_readcyclecounter();

  // ---- the workgroup's tile stream: per item (nk, first tile, tiles), per tile its item and nk, per row its t
  if (tid < nit) {
    const int item = item_lo + tid;
    const int nk = item / s.nsplit, split = item % s.nsplit;
    const int cnt = a.sv.cnt[nk];
    const int tiles_total = (cnt + 15) >> 4;
    const int tiles_per = (tiles_total + s.nsplit - 1) / s.nsplit;
    const int t0 = split * tiles_per, t1 = min(tiles_total, t0 + tiles_per);
    s_ink[tid] = nk;
    s_it0[tid] = t0;
    s_int[tid] = max(0, t1 - t0);
    s_iallm[tid] = a.sv.allmasked[nk] != 0;
    if (t1 <= t0) {  // empty split
      float* part = a.part + (size_t)item * (w + 4);
      part[0] = -INFINITY;
      part[1] = 0.f;
      part[2] = -INFINITY;
    }
  }
  for (int c = tid; c < w; c += NT) {
    if (RMODE != 2) s_vec[0][c] = a.sv.vecs[VEC_RH * w + c];
    if (RMODE != 1) s_vec[1][c] = a.sv.vecs[VEC_R2 * w + c];
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < nit; ++i) {
      s_ifirst[i] = acc;
      acc += s_int[i];
    }
    s_ifirst[nit] = acc;
  }
  __syncthreads();
  const int G = s_ifirst[nit];
  if (G == 0) return;
  for (int it = 0; it < nit; ++it) {
    const int nk = s_ink[it], g0 = s_ifirst[it], ntl = s_int[it];
    for (int tl = tid; tl < ntl; tl += NT) {
      s_titem[g0 + tl] = (uint8_t)it;
      s_tnk[g0 + tl] = nk;
    }
  }
  __syncthreads();
  // one flat pass over the stream's rows: every lookup independent (a per-item loop chains ~10 load latencies)
  for (int r = tid; r < G * 16; r += NT) {
    const int g = r >> 4, it = s_titem[g], nk = s_tnk[g];
    const int lr = (s_it0[it] + g - s_ifirst[it]) * 16 + (r & 15);
    s_idx[r] = lr < a.sv.cnt[nk] ? (uint16_t)a.sv.idx[(size_t)nk * T + lr] : (uint16_t)0xFFFF;
  }
  __syncthreads();

  const bool cosine = s.simi == 4;
  f32x4 fragA[NB], fragB[NB], fragC[NB], uacc[NB];
  float m_run = -INFINITY, l_run = 0.f;
  half8 bhi[NM * 2];
#pragma unroll
  for (int i = 0; i < NM * 2; ++i) bhi[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t qvalid = 0;
  int cur_n = -1;

  // lane (row l15, k quarter kq) loads 4 consecutive channels of its row per 16-channel block
  auto tile_base = [&](int g2) {
    const int nk2 = __builtin_amdgcn_readfirstlane(s_tnk[g2]);
    return a.hinfo + (size_t)nk2 * T * w;
  };
  auto tile_voff = [&](int g2) {
    const int t = s_idx[g2 * 16 + l15];
    // (invalid rows of the last tile read row 0: finite data, weight 0)
    return (unsigned)((t == 0xFFFF ? 0 : t) * (w * 4) + (16 * NB * wave + 4 * kq) * 4);
  };
  auto load_tile = [&](int g2, f32x4(&dst)[NB], auto buf) {
    rows_issue<decltype(buf)::value, NB>(tile_base(g2), tile_voff(g2), dst);
  };

  // FVTA_ATTN_DBG & 16: wave `dbg >> 8` of workgroup 0 stamps the shader clock at each phase boundary of its first
  // 64 tiles into the (otherwise unused) tail of the workspace, 32 MiB in -- see tools/attn_phases.py
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.part) + (32u << 20));
  const bool stamp = (a.dbg & 16) && wg == 0 && wave == ((a.dbg >> 8) & 7) && lane == 0;
#define FVTA_STAMP(k) do { if (stamp && g < 64) stamps[g * 16 + (k)] = __builtin_readcyclecounter(); } while (0)

  auto tile = [&](const int g, f32x4 (&frag)[NB], f32x4 (&next)[NB], auto fbuf, auto nbuf) {
    FVTA_STAMP(0);
    const int it = s_titem[g];
    const int nk = s_tnk[g], n = nk / s.K;
    const bool first = g == s_ifirst[it], last = g + 1 == s_ifirst[it + 1];
    // prefetch distance TWO tiles (three register buffers): with one tile in flight the CU holds 64 KB of
    // outstanding reads at best and ~32 KB on average -- half of what HBM latency x the CU's bandwidth share needs
    if (g + 2 < G) load_tile(g + 2, next, nbuf);
    rows_wait<decltype(fbuf)::value, NB>(g + 2 < G ? 2 : (g + 1 < G ? 1 : 0), frag);  // tile g has landed; g+1, g+2 in flight
    FVTA_STAMP(1);
    if (first) {
      m_run = -INFINITY;
      l_run = 0.f;
#pragma unroll
      for (int i = 0; i < NB; ++i) uacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (n != cur_n) {
        // B operand of MFMA m, j tile jt: lane (col l15, k group kq) holds channels 32m + 4kq + (0..3) and
        // 32m + 16 + 4kq + (0..3) of the wave's slice -- the channels the A lanes of the same k group hold.
        // (buffer loads: one lane offset register + scalar offsets)
        cur_n = n;
        constexpr int W4c = NW * NB * 4;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(a.sv.Qh + (size_t)n * 2 * W4c * 32 * 4), 0, (unsigned)(2 * W4c * 32 * 8), 0x00020000);
        const int lane_off = ((4 * NB * wave + kq) * 32 + l15) * 8;
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
          for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
              typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
              const u32x2 x0 = __builtin_amdgcn_raw_buffer_load_b64(rq, lane_off, ((pc * W4c + 8 * m) * 32 + 16 * jt) * 8, 0);
              const u32x2 x1 = __builtin_amdgcn_raw_buffer_load_b64(rq, lane_off, ((pc * W4c + 8 * m + 4) * 32 + 16 * jt) * 8, 0);
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              const u32x4 xx = __builtin_shufflevector(x0, x1, 0, 1, 2, 3);
              const half8 v = __builtin_bit_cast(half8, xx);
              if (pc == 0) s_bhi[wave][m * 2 + jt][lane] = v;
              else s_blo[wave][m * 2 + jt][lane] = v;  // wave-private: written and read by this wave only
            }
        // (a SCALAR load: it waits on lgkmcnt, not on the vmcnt the row prefetch is counted with)
        // the hi pieces feed two of the three MFMAs of a group: they go on into registers (through LDS, so that the
        // registers never have a VMEM load pending -- see rows_issue); the lo pieces are read from LDS per tile
#pragma unroll
        for (int i = 0; i < NM * 2; ++i) bhi[i] = s_bhi[wave][i][lane];
        qvalid = a.sv.qvalid[(size_t)__builtin_amdgcn_readfirstlane(n) * 2];
        if (tid < 32) s_ct[tid] = a.sv.ct[(size_t)n * JP + tid];  // read after the next barrier
      }
    }
    FVTA_STAMP(2);
    const bool allm = s_iallm[it] != 0;
    if (!allm) {
      f32x4 ahh[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 axx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 rt4 = {0.f, 0.f, 0.f, 0.f};  // row term h.(Rh + R2 h), 4-wide so that it stays two packed FMAs per block
#pragma unroll
      for (int m = 0; m < NM; ++m) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c0 = 16 * (NB * wave + 2 * m + q) + 4 * kq;
          const f32x4 h = frag[2 * m + q];
          if (RMODE == 1) {
            rt4 += h * *reinterpret_cast<const f32x4*>(&s_vec[0][c0]);
          } else if (RMODE == 2) {
            rt4 += (h * h) * *reinterpret_cast<const f32x4*>(&s_vec[1][c0]);
          } else {
            rt4 += h * (*reinterpret_cast<const f32x4*>(&s_vec[0][c0]) + *reinterpret_cast<const f32x4*>(&s_vec[1][c0]) * h);
          }
        }
        half8 hi, lo;
        split_f16x8(frag[2 * m], frag[2 * m + 1], hi, lo);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const half8 bh = bhi[m * 2 + jt];
          ahh[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, bh, ahh[jt], 0, 0, 0);
          axx[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, s_blo[wave][m * 2 + jt][lane], axx[jt], 0, 0, 0);
          axx[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, bh, axx[jt], 0, 0, 0);
        }
        // keep the unrolled iterations apart: hoisting every LDS read and split to the top costs ~60 VGPRs
        __builtin_amdgcn_sched_barrier(0);
      }
      const float rtp = (rt4[0] + rt4[1]) + (rt4[2] + rt4[3]);
      // D layout of 16x16xK: lane -> column j = l15, rows 4*kq + r
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_part[wave][(jt * 4 + r) * 64 + lane] = ahh[jt][r] + axx[jt][r] * (1.f / 2048.f);
      s_rtp[wave * 4 + kq][l15] = rtp;
      FVTA_STAMP(3);
      lds_barrier();
      FVTA_STAMP(4);
      if (NT <= 512 || wave < 8) {
        // (row, j) of this thread: wave pw & 3 owns rows {pw, 4+pw, 8+pw, 12+pw} so that its 64 lanes read 64
        // consecutive partial scores; with 8+ waves waves 4..7 take the second j tile
        constexpr int JPT = NT >= 512 ? 1 : 2;
        const int pw = wave & 3, jt0 = NT >= 512 ? wave >> 2 : 0;
        const int row = 4 * kq + pw;
        float rt = 0.f;
#pragma unroll
        for (int v = 0; v < NPART / 16; ++v) rt += s_rtp[l15 * (NPART / 16) + v][row];
        rt = row16_sum(rt);
        const float rs = cosine ? rsqrtf(fmaxf(rt, 1e-12f)) : 1.f;
        float best = -INFINITY;
        int bestj = 0;
#pragma unroll
        for (int q = 0; q < JPT; ++q) {
          const int jt = jt0 + q, j = l15 + 16 * jt;
          const int pi = (jt * 4 + pw) * 64 + lane;
          float x = 0.f;
#pragma unroll
          for (int v = 0; v < NW; ++v) x += s_part[v][pi];
          x = cosine ? x * rs : x + rt + s_ct[j];
          const bool valid = (qvalid >> j) & 1ull;
          if (valid && x > best) {
            best = x;
            bestj = j;
          }
        }
        row16_argmax(best, bestj);
        if (l15 == 0) {
          s_cand_v[jt0][row] = best;
          s_cand_j[jt0][row] = bestj;
          if (NT < 512) s_cand_v[1][row] = -INFINITY;
        }
      }
    }
    FVTA_STAMP(5);
    if (!allm) lds_barrier();
    FVTA_STAMP(6);
    // online softmax over t (softsel inner, model_v2.py:278); lane l holds row l15 of the tile
    float am;
    {
      const int t = s_idx[g * 16 + l15];
      if (allm) {
        am = t != 0xFFFF ? FVTA_NEG : -INFINITY;
      } else {
        const float v0 = s_cand_v[0][l15], v1 = s_cand_v[1][l15];
        const bool second = v1 > v0;  // ties keep the smaller j
        const float best = second ? v1 : v0;
        const int bestj = second ? s_cand_j[1][l15] : s_cand_j[0][l15];
        am = t != 0xFFFF ? (s.add_tanh ? fvta_tanh(best) : best) : -INFINITY;
        if (wave == 0 && kq == 0 && t != 0xFFFF) {
          a.sv.amax[(size_t)nk * T + t] = am;
          a.sv.jmax[(size_t)nk * T + t] = (uint8_t)bestj;
        }
      }
    }
    const float m_new = fmaxf(m_run, row16_max(am));
    const float scale = expf(m_run - m_new);
    const float pr = expf(am - m_new);
    l_run = l_run * scale + row16_sum(pr);
    m_run = m_new;
    if (scale != 1.f) {  // workgroup-uniform; the running max rarely moves after the first tiles of an item
#pragma unroll
      for (int i = 0; i < NB; ++i) uacc[i] *= scale;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) uacc[i] += frag[i] * pr;
    FVTA_STAMP(7);
    if (last) {
      // fold the 16 row lanes, store the item's partial (m, l, u)
      float* part = a.part + (size_t)(item_lo + it) * (w + 4);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        f32x4 u = uacc[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = row16_sum(u[e]);
        if (l15 == 0) *reinterpret_cast<f32x4*>(part + 4 + 16 * (NB * wave + i) + 4 * kq) = u;
      }
      if (tid == 0) {
        part[0] = m_run;
        part[1] = l_run;
        part[2] = m_run;  // no time_warp_att in this kernel: the softmax logits are amax itself
      }
    }
  };

  if (stamp) {
    stamps[15] = t_entry;
    stamps[14] = __builtin_readcyclecounter();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of the prologue is in flight when the counting starts
  load_tile(0, fragA, BufTag<0>{});
  if (G > 1) load_tile(1, fragB, BufTag<1>{});
  for (int g = 0; g < G; g += 3) {  // (tile, buffer of tile g, buffer that takes tile g + 2)
    tile(g, fragA, fragC, BufTag<0>{}, BufTag<2>{});
    if (g + 1 < G) tile(g + 1, fragB, fragA, BufTag<1>{}, BufTag<0>{});
    if (g + 2 < G) tile(g + 2, fragC, fragB, BufTag<2>{}, BufTag<1>{});
  }
#undef FVTA_STAMP
}

// ---- one wave per 16-row tile (JQ <= 32, 256 <= w <= 1024, simi 1-3) -----------------------------------------------
// The kernel above shares a tile among eight waves (each owns w/8 channels), so every tile pays two workgroup barriers and
// a cross-wave reduction of the partial scores: a ~7000-cycle dependent chain per 64 KB.  Here a tile belongs to ONE wave:
// the 16 rows (w/16 float4 registers per lane, lane = row l15, k quarter kq -- up to 256 registers, the kernel runs at one
// wave per SIMD with the AGPRs as the second half of the register file) are scored over all channels on the fp16-split
// pipe with the question operand read from LDS (hi and lo pieces, 2 x w x 32 x 2 B = 128 KB at w = 1024, filled once per
// workgroup: its four waves work on items of ONE n), max / arg-max / online softmax run inside the wave, and the weighted
// sum uses the same registers, folded over the 16 row lanes with DPP row sums -- lane (row r, quarter kq) keeps the
// channel blocks b = r (mod 16), so the accumulator is w/256 float4 registers.  No barrier after the prologue, no
// cross-wave traffic, rows read once; a (n, k, split) item is a wave's own tile stream and ends in the same (m, l, u)
// partial as the kernels above.  grid: G workgroups per n, XCD-contiguous in n.
template <int NBLK, int RMODE>
__global__ __launch_bounds__(256, 1) void attn_fwd_wave16(AttnFwdArgs a, int G) {
  constexpr int NKS = NBLK / 2;  // K = 32 MFMA steps over the channels
  constexpr int NU = NBLK / 16;  // accumulator registers (float4) per lane
  static_assert(NBLK % 16 == 0, "channel blocks are dealt round-robin to the 16 row lanes");
  extern __shared__ __attribute__((aligned(16))) char s_dyn[];
  half8(*s_qhi)[2][64] = reinterpret_cast<half8(*)[2][64]>(s_dyn);                                   // [NKS][2][64]
  half8(*s_qlo)[2][64] = reinterpret_cast<half8(*)[2][64]>(s_dyn + (size_t)NKS * 2 * 64 * sizeof(half8));
  float* s_vec = reinterpret_cast<float*>(s_dyn + (size_t)2 * NKS * 2 * 64 * sizeof(half8));         // [2][w]
  __shared__ float s_ct[32];

  const AttnShape& s = a.s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, kq = lane >> 4;
  const int T = s.T, w = s.w, JP = s.JP;
  const int nwg = s.N * G, per = (nwg + 7) / 8;
  const int wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (wg >= nwg || (int)(blockIdx.x >> 3) >= per) return;
  const int n = wg / G, g0 = wg % G;

  // ---- the question operand of n, in MFMA B lane order: step ks, j tile jt, lane (col l15, k group kq) holds channels
  // 32 ks + 4 kq + (0..3) and 32 ks + 16 + 4 kq + (0..3)
  {
    const int W4c = w / 4;
    const uint16_t* qh = a.sv.Qh + (size_t)n * 2 * W4c * 32 * 4;
    for (int e = tid; e < 2 * NKS * 2 * 64; e += 256) {
      const int ln = e & 63, jt = (e >> 6) & 1, ks = (e >> 7) % NKS, pc = (e >> 7) / NKS;
      const int j = (ln & 15) + 16 * jt, q4 = ln >> 4;
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x2 x0 = *reinterpret_cast<const u32x2*>(qh + (((size_t)pc * W4c + 8 * ks + q4) * 32 + j) * 4);
      const u32x2 x1 = *reinterpret_cast<const u32x2*>(qh + (((size_t)pc * W4c + 8 * ks + 4 + q4) * 32 + j) * 4);
      const u32x4 xx = __builtin_shufflevector(x0, x1, 0, 1, 2, 3);
      (pc == 0 ? s_qhi : s_qlo)[ks][jt][ln] = __builtin_bit_cast(half8, xx);
    }
    for (int c = tid; c < w; c += 256) {
      s_vec[c] = a.sv.vecs[VEC_RH * w + c];
      s_vec[w + c] = a.sv.vecs[VEC_R2 * w + c];
    }
    if (tid < 32) s_ct[tid] = a.sv.ct[(size_t)n * JP + tid];
  }
  const uint64_t qvalid = a.sv.qvalid[(size_t)n * 2];
  __syncthreads();

  const int nitems_n = s.K * s.nsplit;
  for (int il = g0 + G * wave; il < nitems_n; il += 4 * G) {
    const int k = il / s.nsplit, split = il % s.nsplit;
    const int nk = n * s.K + k;
    const int cnt = a.sv.cnt[nk];
    const bool allm = a.sv.allmasked[nk] != 0;
    const int tiles_total = (cnt + 15) >> 4;
    const int tiles_per = (tiles_total + s.nsplit - 1) / s.nsplit;
    const int t0 = split * tiles_per, t1 = min(tiles_total, t0 + tiles_per);
    float* part = a.part + ((size_t)nk * s.nsplit + split) * (w + 4);
    if (t1 <= t0) {  // empty split
      if (lane == 0) {
        part[0] = -INFINITY;
        part[1] = 0.f;
        part[2] = -INFINITY;
      }
      continue;
    }
    const float* hbase = a.hinfo + (size_t)nk * a.hstride;
    const int32_t* idx = a.sv.idx + (size_t)nk * T;
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 u[NU];   // lane (row l15, quarter kq): channels 16 (16 i + l15) + 4 kq + (0..3) of the weighted sum
#pragma unroll
    for (int i = 0; i < NU; ++i) u[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the tile lives in h[] across the loop: the weighted sum hands every block's registers to the NEXT tile's load as
    // soon as it has consumed them, so the next tile streams in under this tile's tail and the following score pass
    f32x4 h[NBLK];
    int t_cur;
    bool v_cur;
    {
      const int lr = t0 * 16 + l15;
      v_cur = lr < cnt;
      t_cur = v_cur ? idx[lr] : 0;  // (invalid rows of the last tile read row 0: finite data, weight 0)
      const float* rowp = hbase + (size_t)t_cur * w + 4 * kq;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) h[b] = *reinterpret_cast<const f32x4*>(rowp + 16 * b);
    }
    for (int tl = t0; tl < t1; ++tl) {
      const bool rvalid = v_cur;
      const int t = t_cur;
      const bool has_next = tl + 1 < t1;
      const int lrn = (tl + 1) * 16 + l15;
      const bool v_next = has_next && lrn < cnt;
      const int t_next = v_next ? idx[lrn] : 0;
      const float* rowp_next = hbase + (size_t)t_next * w + 4 * kq;
      float am = rvalid ? FVTA_NEG : -INFINITY;  // (a fully masked (n,k): every row of the identity list at -1e30)
      if (!allm) {
        f32x4 ahh[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        f32x4 axx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        f32x4 rt4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int c0 = 16 * (2 * ks + q) + 4 * kq;
            const f32x4 hv = h[2 * ks + q];
            if (RMODE == 1)
              rt4 += hv * *reinterpret_cast<const f32x4*>(&s_vec[c0]);
            else if (RMODE == 2)
              rt4 += (hv * hv) * *reinterpret_cast<const f32x4*>(&s_vec[w + c0]);
            else
              rt4 += hv * (*reinterpret_cast<const f32x4*>(&s_vec[c0]) + *reinterpret_cast<const f32x4*>(&s_vec[w + c0]) * hv);
          }
          half8 hi, lo;
          split_f16x8(h[2 * ks], h[2 * ks + 1], hi, lo);
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            const half8 bh = s_qhi[ks][jt][lane];
            ahh[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, bh, ahh[jt], 0, 0, 0);
            axx[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, s_qlo[ks][jt][lane], axx[jt], 0, 0, 0);
            axx[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, bh, axx[jt], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the unrolled steps apart (hoisted LDS reads / splits cost registers)
        }
        // the row term of row l15: fold its four k quarters (lanes l15, l15 + 16, + 32, + 48)
        float rtp = (rt4[0] + rt4[1]) + (rt4[2] + rt4[3]);
        rtp += __shfl_xor(rtp, 16, 64);
        rtp += __shfl_xor(rtp, 32, 64);
        // D layout of 16x16xK: lane -> column j = l15 (+ 16 jt), rows 4 kq + i.  Per row: max / first arg-max over j.
        float amr[4];
        int jmr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float rti = __shfl(rtp, 4 * kq + i, 64);
          float best = -INFINITY;
          int bestj = 0;
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            const int j = l15 + 16 * jt;
            const float x = ahh[jt][i] + axx[jt][i] * (1.f / 2048.f) + rti + s_ct[j];
            if (((qvalid >> j) & 1ull) && x > best) {
              best = x;
              bestj = j;
            }
          }
          row16_argmax(best, bestj);
          amr[i] = best;
          jmr[i] = bestj;
        }
        // row l15's result sits in the lane group l15 >> 2 as its entry l15 & 3
        const int src = ((l15 >> 2) << 4) | l15;
        float bestv = 0.f;
        int bestj = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = __shfl(amr[i], src, 64);
          const int jj = __shfl(jmr[i], src, 64);
          if ((l15 & 3) == i) {
            bestv = v;
            bestj = jj;
          }
        }
        am = rvalid ? (s.add_tanh ? fvta_tanh(bestv) : bestv) : -INFINITY;
        if (kq == 0 && rvalid) {
          a.sv.amax[(size_t)nk * T + t] = am;
          a.sv.jmax[(size_t)nk * T + t] = (uint8_t)bestj;
        }
      }
      // online softmax over t (softsel inner, model_v2.py:278); every lane of a DPP row holds its own row l15
      const float m_new = fmaxf(m_run, row16_max(am));
      const float scale = expf(m_run - m_new);
      const float pr = expf(am - m_new);
      l_run = l_run * scale + row16_sum(pr);
      m_run = m_new;
      // weighted sum u[c] = u[c] * scale + sum_r p_r h[r, c]: every block summed over the 16 row lanes by DPP row sums
      // (an all-reduce), kept by lane b & 15.  Measured alternatives, all slower with one wave per SIMD: a reduce-scatter
      // over the row lanes (60 instead of 256 values per 16 blocks, but two selects per value and one long dependent
      // chain: 0.48 vs 0.45 ms); a transposition through a wave-private LDS scratch (a sixth of the VALU work, but two
      // exposed LDS round trips per 64 channels: 0.49); two blocks per butterfly step to cover the DPP read hazard
      // (fewer s_nop, but the compiler spills: 0.50).
      if (scale != 1.f) {
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] *= scale;
      }
      if (has_next) {
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          f32x4 v = h[b] * pr;
          h[b] = *reinterpret_cast<const f32x4*>(rowp_next + 16 * b);
          row16_sum4(v);
          if ((b & 15) == l15) u[b >> 4] += v;
        }
      } else {
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          f32x4 v = h[b] * pr;
          row16_sum4(v);
          if ((b & 15) == l15) u[b >> 4] += v;
        }
      }
      t_cur = t_next;
      v_cur = v_next;
    }
    // the item's partial (m, l, u): lane (l15, kq) holds channels 16 (16 i + l15) + 4 kq + (0..3)
#pragma unroll
    for (int i = 0; i < NU; ++i) *reinterpret_cast<f32x4*>(part + 4 + 16 * (16 * i + l15) + 4 * kq) = u[i];
    if (lane == 0) {
      part[0] = m_run;
      part[1] = l_run;
      part[2] = m_run;
    }
  }
}

// ---- two waves per 16-row tile (JQ <= 32, 512 <= w <= 1024, simi 1-3) ---------------------------------------------
// attn_fwd_wave16 with the tile split over a PAIR of waves (each holds half the channels: w/32 float4 registers per lane),
// eight waves = four pairs per workgroup, two waves per SIMD: a pair exchanges its partial scores through LDS once per
// tile (one workgroup barrier on each side of the exchange), everything else -- max / arg-max / softmax (computed by both
// waves of a pair), the weighted sum of a wave's own channels, the rolling refill of the tile registers -- is as there.
// Against attn_fwd_rows16 (eight waves per tile): a 2-way instead of an 8-way reduction, four independent tile streams
// per workgroup, and a quarter of the barriers per byte; against attn_fwd_wave16: a second wave per SIMD to cover the
// first one's latencies.  The pairs of a workgroup run the same number of barrier rounds (the longest pair's tile count).
// FLAGS: the two waves of a pair synchronise through LDS flags (published / consumed round numbers, polled with s_sleep)
// instead of workgroup barriers, so the four pairs of a workgroup drift apart freely (the polls are bounded: a broken
// hand-shake gives wrong numbers, never a hung GPU).
template <int NBH, int RMODE, bool FLAGS = false>
__global__ __launch_bounds__(512, 1) void attn_fwd_pair16(AttnFwdArgs a, int G_all) {
  constexpr int NKS = NBH / 2;   // MFMA steps over a wave's half of the channels
  constexpr int NU = NBH / 16;   // accumulator registers (float4) per lane
  static_assert(NBH % 16 == 0, "a wave's channel blocks are dealt round-robin to the 16 row lanes");
  extern __shared__ __attribute__((aligned(16))) char s_dyn[];
  half8(*s_qhi)[2][64] = reinterpret_cast<half8(*)[2][64]>(s_dyn);                                        // [2 NKS][2][64]
  half8(*s_qlo)[2][64] = reinterpret_cast<half8(*)[2][64]>(s_dyn + (size_t)2 * NKS * 2 * 64 * sizeof(half8));
  float* s_vec = reinterpret_cast<float*>(s_dyn + (size_t)2 * 2 * NKS * 2 * 64 * sizeof(half8));           // [2][w]
  __shared__ float s_ct[32];
  __shared__ __attribute__((aligned(16))) float s_x[8][8 * 64];  // per wave: its partial scores [jt * 4 + i][lane]
  __shared__ float s_rt[8][16];                                  // per wave: its partial row terms
  __shared__ int s_tiles[4];
  __shared__ int s_kstart[65], s_kcnt[64], s_kall[64], s_flat;  // the n's streams: first flat tile, valid rows, fully masked
  __shared__ int s_pub[8], s_done[8];  // FLAGS: last round whose partials a wave has published / whose partner data it has consumed
#ifdef FVTA_DIAG
  // FVTA_ATTN_DBG & 16: wave `dbg >> 8` of workgroup 0 stamps the shader clock at the phase boundaries of its first 32
  // rounds (into LDS, dumped 32 MiB into the workspace at the end) -- tools/attn_phases.py
  __shared__ unsigned long long s_stamp[32][8];
#define FVTA_PSTAMP(k) do { if (pstamp && g < 32) s_stamp[g][(k)] = __builtin_readcyclecounter(); } while (0)
#else
#define FVTA_PSTAMP(k) do { } while (0)
#endif

  const AttnShape& s = a.s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pair = wave >> 1, hv = wave & 1;
  const int l15 = lane & 15, kq = lane >> 4;
  const int T = s.T, w = s.w, JP = s.JP;
  const int nwg = s.N * G_all, per = (nwg + 7) / 8;
  const int wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (wg >= nwg || (int)(blockIdx.x >> 3) >= per) return;
  // the n this workgroup serves, its index among that n's workgroups and their number: uniform (G_all each), or dealt in
  // proportion to the n's valid tiles (masked batches: albums differ in rows)
  int n = wg / G_all, g0 = wg % G_all, G = G_all;
  if (a.wgtab) {
    const uint32_t e = a.wgtab[wg];
    if (e == 0xffffffffu) return;
    n = (int)(e & 0xffffu);
    g0 = (int)((e >> 16) & 0xffu);
    G = (int)(e >> 24);
  }
  // -DFVTA_PAIR_ABL=bits: compile-time ablations (timing only, results are wrong; a run-time switch makes the compiler
  // spill): 1 no tile loads after an item's first, 2 no score MFMA loop, 4 no weighted sum (refill only), 8 no pair
  // hand-shake -- tools/r02_v.sh
#ifdef FVTA_PAIR_ABL
  constexpr int abl = FVTA_PAIR_ABL;
#else
  constexpr int abl = 0;
#endif
#ifdef FVTA_DIAG
  const bool pstamp = (a.dbg & 16) && wg == 0 && wave == ((a.dbg >> 8) & 7) && lane == 0;
  const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
  {
    const int W4c = w / 4;
    const uint16_t* qh = a.sv.Qh + (size_t)n * 2 * W4c * 32 * 4;
    // every load of the staging is issued before the first LDS write: as a rolled loop (two loads, wait, write) the
    // 128 KB took 16 dependent round trips per thread, 35-50 k cycles before the first tile was even requested
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int QIT = 2 * 2 * NKS * 2 * 64 / 512;
    u32x2 x0[QIT], x1[QIT];
#pragma unroll
    for (int it = 0; it < QIT; ++it) {
      const int e = tid + 512 * it;
      const int ln = e & 63, jt = (e >> 6) & 1, ks = (e >> 7) % (2 * NKS), pc = (e >> 7) / (2 * NKS);
      const int j = (ln & 15) + 16 * jt, q4 = ln >> 4;
      x0[it] = *reinterpret_cast<const u32x2*>(qh + (((size_t)pc * W4c + 8 * ks + q4) * 32 + j) * 4);
      x1[it] = *reinterpret_cast<const u32x2*>(qh + (((size_t)pc * W4c + 8 * ks + 4 + q4) * 32 + j) * 4);
    }
    float v0[2], v1[2];  // w <= 1024: at most two channels of each row-term vector per thread
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + 512 * it;
      v0[it] = c < w ? a.sv.vecs[VEC_RH * w + c] : 0.f;
      v1[it] = c < w ? a.sv.vecs[VEC_R2 * w + c] : 0.f;
    }
    const float ctv = tid < 32 ? a.sv.ct[(size_t)n * JP + tid] : 0.f;
#pragma unroll
    for (int it = 0; it < QIT; ++it) {
      const int e = tid + 512 * it;
      const int ln = e & 63, jt = (e >> 6) & 1, ks = (e >> 7) % (2 * NKS), pc = (e >> 7) / (2 * NKS);
      const u32x4 xx = __builtin_shufflevector(x0[it], x1[it], 0, 1, 2, 3);
      (pc == 0 ? s_qhi : s_qlo)[ks][jt][ln] = __builtin_bit_cast(half8, xx);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + 512 * it;
      if (c < w) {
        s_vec[c] = v0[it];
        s_vec[w + c] = v1[it];
      }
    }
    if (tid < 32) s_ct[tid] = ctv;
  }
  const uint64_t qvalid = a.sv.qvalid[(size_t)n * 2];
  const int nitems_n = s.K * s.nsplit;
  const int P = 4 * G, pg = 4 * g0 + pair;  // pairs that share this n, this pair's index among them
  // ---- the streams of this n: valid rows, first tile (in the n's flat tile order), fully-masked flag
  if (tid == 64) {
    int acc = 0;
    for (int k = 0; k < s.K; ++k) {
      const int c = a.sv.cnt[n * s.K + k];
      s_kcnt[k] = c;
      s_kall[k] = a.sv.allmasked[n * s.K + k];
      s_kstart[k] = acc;
      acc += (c + 15) >> 4;
    }
    s_kstart[s.K] = acc;
    // FLAT dealing: the n's tiles, in (k, tile) order, are cut into P equal runs, one per pair (a run crosses stream
    // boundaries; pair p's piece of stream k is that stream's partial number p - (first pair that touches k)).  Only if
    // no stream is cut into more pieces than it has partial slots (nsplit); otherwise the (k, split) items are dealt
    // round-robin as in the other kernels (pieces of ceil(tiles / nsplit) tiles: unequal sums per pair).
    int ok = acc > 0;
    for (int k = 0; k < s.K && ok; ++k) {
      const int st = s_kstart[k], en = s_kstart[k + 1];
      if (en > st && ((en * P - 1) / acc) - (((st + 1) * P - 1) / acc) + 1 > s.nsplit) ok = 0;
    }
    s_flat = ok;
  }
  if (tid < 8) {
    s_pub[tid] = 0;
    s_done[tid] = 0;
  }
  __syncthreads();
  const bool flat = s_flat != 0;
  const int tot = s_kstart[s.K];
  const int lo = flat ? tot * pg / P : 0, hi = flat ? tot * (pg + 1) / P : 0;  // this pair's run (flat dealing)
  auto empty_partial = [&](int nk, int split) {
    float* pp = a.part + ((size_t)nk * s.nsplit + split) * (w + 4);
    pp[0] = -INFINITY;
    pp[1] = 0.f;
    pp[2] = -INFINITY;
  };
  if (flat && g0 == 0) {  // the partial slots no pair fills
    for (int e = tid; e < nitems_n; e += 512) {
      const int k = e / s.nsplit, sp = e % s.nsplit;
      const int st = s_kstart[k], en = s_kstart[k + 1];
      bool filled = false;  // slot sp belongs to pair (first pair that touches k) + sp, if that pair's run meets k at all
      if (en > st) {
        const int px = ((st + 1) * P - 1) / tot + sp;
        filled = px < P && max(tot * px / P, st) < min(tot * (px + 1) / P, en);
      }
      if (!filled) empty_partial(n * s.K + k, sp);
    }
  }
  // ---- the pair's pieces ("segments": consecutive tiles [t0, t1) of one stream, summed into one partial)
  struct Seg {
    int nk, t0, t1, slot, cnt, allm;
  };
  auto item_seg = [&](int il, Seg& sg) {  // round-robin dealing: item il = (k, split)
    const int k = il / s.nsplit, split = il % s.nsplit;
    const int c = s_kcnt[k];
    const int tiles_total = (c + 15) >> 4;
    const int tiles_per = (tiles_total + s.nsplit - 1) / s.nsplit;
    sg.nk = n * s.K + k;
    sg.t0 = split * tiles_per;
    sg.t1 = min(tiles_total, sg.t0 + tiles_per);
    sg.slot = split;
    sg.cnt = c;
    sg.allm = s_kall[k];
    return sg.t1 > sg.t0;
  };
  int it_k = 0, it_il = g0 + G * pair - 4 * G;
  auto next_seg = [&](Seg& sg) {  // false: none left
    if (flat) {
      while (it_k < s.K) {
        const int k = it_k++;
        const int st = s_kstart[k], en = s_kstart[k + 1];
        if (st >= hi) break;
        const int x0 = max(lo, st), x1 = min(hi, en);
        if (x0 < x1) {
          sg.nk = n * s.K + k;
          sg.t0 = x0 - st;
          sg.t1 = x1 - st;
          sg.slot = pg - ((st + 1) * P - 1) / tot;
          sg.cnt = s_kcnt[k];
          sg.allm = s_kall[k];
          return true;
        }
      }
      it_k = s.K;
      return false;
    }
    for (;;) {
      it_il += 4 * G;
      if (it_il >= nitems_n) return false;
      if (item_seg(it_il, sg)) return true;
      if (hv == 0 && lane == 0) empty_partial(sg.nk, sg.slot);  // empty split
    }
  };
  int myrounds = hi - lo;
  if (!flat) {
    myrounds = 0;
    Seg sg;
    for (int il = g0 + G * pair; il < nitems_n; il += 4 * G)
      if (item_seg(il, sg)) myrounds += sg.t1 - sg.t0;
  }
  int rounds = myrounds;
  if (!FLAGS) {  // the barrier version runs every pair for the longest pair's number of rounds
    if (hv == 0 && lane == 0) s_tiles[pair] = myrounds;
    __syncthreads();
    rounds = max(max(s_tiles[0], s_tiles[1]), max(s_tiles[2], s_tiles[3]));
  }
  const int pwv = wave ^ 1;
  // (the flags are accessed through LDS-address-space pointers: through a generic pointer the compiler emits FLAT
  //  loads/stores, whose s_waitcnt vmcnt(0) would also wait for every outstanding load of the next tile)
  typedef __attribute__((address_space(3))) int lds_int;
  auto wait_flag = [&](int* flag, int want) {  // bounded poll of an LDS word
    volatile lds_int* f = (volatile lds_int*)flag;
    bool arrived = false;
    // (the partner wave is resident in this very workgroup: it can only be DELAYED -- counter collection serialising waves,
    //  pre-emption, a debugger -- so the bound is generous: 2^28 polls of s_sleep 2, tens of seconds)
    for (int spin = 0; spin < (1 << 28); ++spin) {
      if (*f >= want) {
        arrived = true;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    // a partner that never arrives is a bug (a wedged wave).  The trap aborts the queue -- on ROCm that usually ends the
    // process, it is NOT a recoverable launch error -- which is still better than folding stale partials into amax /
    // jmax / h_a and training on them
    if (!arrived) {
      if (a.fault) {  // (host-mapped: visible to the host once the system-scope fence has drained)
        __hip_atomic_store(a.fault, (int)ATTN_FAULT_PAIR_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
      }
      __builtin_trap();
    }
    // acquire: the partner's published area (plain LDS loads below) is read only after the poll has matched; workgroup
    // scope lowers to s_waitcnt lgkmcnt(0) and, unlike an empty asm, is a compiler fence for __shared__ accesses too
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  auto post_flag = [&](int* flag, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // my partials are written before the flag says so
    *(volatile lds_int*)flag = v;
  };

  // ---- the pair's tile stream: the CURRENT tile is in h[], the NEXT tile's identity and row numbers are known one
  // round ahead (its rows replace the current tile's registers during the weighted sum, also across a stream boundary)
  Seg cs = {0, 0, 0, 0, 0, 0}, ns = {0, 0, 0, 0, 0, 0};
  int ctl = 0, ntl = 0;  // tile numbers within their streams
  int t_cur = 0, t_nxt = 0;
  bool v_cur = false, v_nxt = false;
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 u[NU];
  f32x4 h[NBH];
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int coff = 16 * NBH * hv + 4 * kq;  // this lane's first channel: block b of the wave's half is at coff + 16 b
  // (an invalid row of a stream's last tile reads the stream's last valid row: finite data, weight 0.  No select on
  //  the loaded value: the load's destination is the loop-carried register, nothing waits for it before its use)
  auto rows_of = [&](const Seg& sg, int tl, int& t, bool& v) {
    const int lr = tl * 16 + l15;
    v = lr < sg.cnt;
    t = a.sv.idx[(size_t)sg.nk * T + max(min(lr, sg.cnt - 1), 0)];
  };
  auto tile_after = [&](const Seg& from, int ftl, Seg& to, int& ttl) {
    if (ftl + 1 < from.t1) {
      to = from;
      ttl = ftl + 1;
      return true;
    }
    if (next_seg(to)) {
      ttl = to.t0;
      return true;
    }
    return false;
  };
  bool active = next_seg(cs), has_n = false;
  if (active) {
    ctl = cs.t0;
    rows_of(cs, ctl, t_cur, v_cur);
    const float* rowp = a.hinfo + (size_t)cs.nk * a.hstride + (size_t)t_cur * w + coff;
#pragma unroll
    for (int b = 0; b < NBH; ++b) h[b] = *reinterpret_cast<const f32x4*>(rowp + 16 * b);
    has_n = tile_after(cs, ctl, ns, ntl);
    if (!has_n) {
      ns = cs;
      ntl = ctl;
    }
    rows_of(ns, ntl, t_nxt, v_nxt);  // (unconditional: without a next tile it re-reads the current tile's row numbers)
  }

#pragma unroll 2
  for (int g = 0; g < rounds; ++g) {
    FVTA_PSTAMP(0);
    if (!FLAGS) lds_barrier();  // every wave is done with the previous round's exchange area
    const bool rvalid = v_cur;
    const int t = t_cur;
    const bool allm = cs.allm != 0;
    float xown[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xown[i] = 0.f;
    if (active && !allm && !(abl & 2)) {
      f32x4 ahh[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 axx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 rt4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c0 = coff + 16 * (2 * ks + q);
          const f32x4 hv4 = h[2 * ks + q];
          if (RMODE == 1)
            rt4 += hv4 * *reinterpret_cast<const f32x4*>(&s_vec[c0]);
          else if (RMODE == 2)
            rt4 += (hv4 * hv4) * *reinterpret_cast<const f32x4*>(&s_vec[w + c0]);
          else
            rt4 += hv4 * (*reinterpret_cast<const f32x4*>(&s_vec[c0]) + *reinterpret_cast<const f32x4*>(&s_vec[w + c0]) * hv4);
        }
        half8 hi8, lo8;
        split_f16x8(h[2 * ks], h[2 * ks + 1], hi8, lo8);
        const int kg = NKS * hv + ks;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const half8 bh = s_qhi[kg][jt][lane];
          ahh[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi8, bh, ahh[jt], 0, 0, 0);
          axx[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi8, s_qlo[kg][jt][lane], axx[jt], 0, 0, 0);
          axx[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo8, bh, axx[jt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) FVTA_PSTAMP(1);
      }
      FVTA_PSTAMP(2);
      float rtp = (rt4[0] + rt4[1]) + (rt4[2] + rt4[3]);
      rtp += __shfl_xor(rtp, 16, 64);
      rtp += __shfl_xor(rtp, 32, 64);
      if (FLAGS && !(abl & 8)) wait_flag(&s_done[pwv], g);  // the partner has read my partials of round g - 1 (rounds are numbered from 1)
      if (kq == 0) s_rt[wave][l15] = rtp;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xown[jt * 4 + i] = ahh[jt][i] + axx[jt][i] * (1.f / 2048.f);
          s_x[wave][(jt * 4 + i) * 64 + lane] = xown[jt * 4 + i];
        }
    }
    // (the next tile's row numbers were loaded behind the current tile: they have landed with it.  Using them here keeps
    //  the wait for them from moving behind this round's amax / jmax stores, where it would wait for those too)
    asm volatile("" ::"v"(t_nxt));
    if (FLAGS) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my partials are in LDS (a wave's LDS operations complete in order)
      if (lane == 0) post_flag(&s_pub[wave], g + 1);
      FVTA_PSTAMP(3);
      if (!(abl & 8)) wait_flag(&s_pub[pwv], g + 1);
    } else {
      lds_barrier();  // both halves of every tile are published
    }
    FVTA_PSTAMP(4);
    if (active) {
      float am = rvalid ? FVTA_NEG : -INFINITY;
      if (!allm) {
        const int pw = wave ^ 1;
        float amr[4];
        int jmr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float rti = s_rt[wave][4 * kq + i] + s_rt[pw][4 * kq + i];
          float best = -INFINITY;
          int bestj = 0;
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            const int j = l15 + 16 * jt;
            const int pi = (jt * 4 + i) * 64 + lane;
            // (own + partner: floating-point addition commutes, so both waves of the pair get bit-identical sums)
            const float x = (xown[jt * 4 + i] + s_x[pw][pi]) + rti + s_ct[j];
            if (((qvalid >> j) & 1ull) && x > best) {
              best = x;
              bestj = j;
            }
          }
          row16_argmax(best, bestj);
          amr[i] = best;
          jmr[i] = bestj;
        }
        const int src = ((l15 >> 2) << 4) | l15;
        float bestv = 0.f;
        int bestj = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = __shfl(amr[i], src, 64);
          const int jj = __shfl(jmr[i], src, 64);
          if ((l15 & 3) == i) {
            bestv = v;
            bestj = jj;
          }
        }
        am = rvalid ? (s.add_tanh ? fvta_tanh(bestv) : bestv) : -INFINITY;
        if (hv == 0 && kq == 0 && rvalid) {
          a.sv.amax[(size_t)cs.nk * T + t] = am;
          a.sv.jmax[(size_t)cs.nk * T + t] = (uint8_t)bestj;
        }
      }
      if (FLAGS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the partner's partials are in my registers
        if (lane == 0) post_flag(&s_done[wave], g + 1);
      }
      const float m_new = fmaxf(m_run, row16_max(am));
      const float scale = expf(m_run - m_new);
      const float pr = expf(am - m_new);
      l_run = l_run * scale + row16_sum(pr);
      m_run = m_new;
      if (scale != 1.f) {
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] *= scale;
      }
      FVTA_PSTAMP(5);
      if (has_n) {
        const float* rowp_next = a.hinfo + (size_t)ns.nk * a.hstride + (size_t)t_nxt * w + coff;
#ifdef FVTA_PAIR_ABL
        if (abl & 4) {
#pragma unroll
          for (int b = 0; b < NBH; ++b) h[b] = *reinterpret_cast<const f32x4*>(rowp_next + 16 * b);
        } else if (abl & 1) {
#pragma unroll
          for (int b = 0; b < NBH; ++b) {
            f32x4 v = h[b] * pr;
            row16_sum4(v);
            if ((b & 15) == l15) u[b >> 4] += v;
          }
        } else
#endif
#pragma unroll
        for (int b = 0; b < NBH; ++b) {
          f32x4 v = h[b] * pr;
          h[b] = *reinterpret_cast<const f32x4*>(rowp_next + 16 * b);
          row16_sum4(v);
          if ((b & 15) == l15) u[b >> 4] += v;
        }
      } else {
#pragma unroll
        for (int b = 0; b < NBH; ++b) {
          f32x4 v = h[b] * pr;
          row16_sum4(v);
          if ((b & 15) == l15) u[b >> 4] += v;
        }
      }
      if (ctl + 1 == cs.t1) {
        // the segment's partial (m, l, u): lane (l15, kq) holds channels coff + 16 (16 i + l15) + (0..3)
        float* part = a.part + ((size_t)cs.nk * s.nsplit + cs.slot) * (w + 4);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          *reinterpret_cast<f32x4*>(part + 4 + coff + 16 * (16 * i + l15)) = u[i];
          u[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (hv == 0 && lane == 0) {
          part[0] = m_run;
          part[1] = l_run;
          part[2] = m_run;
        }
        m_run = -INFINITY;
        l_run = 0.f;
      }
      // the next tile becomes the current one; look one tile further ahead
      active = has_n;
      if (has_n) {
        cs = ns;
        ctl = ntl;
        t_cur = t_nxt;
        v_cur = v_nxt;
        has_n = tile_after(cs, ctl, ns, ntl);
        if (!has_n) {
          ns = cs;
          ntl = ctl;
        }
      }
    }
    // (at the top level of the loop body, so that the load's destination IS the loop-carried register: inside the
    //  conditionals above the compiler loads into a temporary and copies it at the loop latch -- behind an
    //  s_waitcnt vmcnt(0) that also waits for the whole next tile)
    rows_of(ns, ntl, t_nxt, v_nxt);
    FVTA_PSTAMP(6);
  }
#ifdef FVTA_DIAG
  if (pstamp) {
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.part) + (32u << 20));
    for (int g = 0; g < 32; ++g)
      for (int k = 0; k < 8; ++k) stamps[g * 16 + k] = s_stamp[g][k];
    stamps[15] = t_entry;
  }
#endif
}

// ---- time_warp_att only: the MASKED rows of a (n,k) that has valid rows.  The reference scales the max-pooled logit
// AFTER exp_mask (model_v2.py:263-275), so a masked row's softmax logit is -1e30 * tscale[n,t]: hugely negative for
// tscale > 0 (weight exactly 0, as exp underflows), but +huge for tscale < 0 -- then the masked rows take the whole
// softmax (one-hot on the smallest tscale, uniform over exact ties; rows that are padding for every modality share one
// c[n,t] and tie).  The main kernels read valid rows only; this kernel adds the masked rows' share as one more partial
// (m, l, u) per (n,k), gathering the few rows that carry weight.  grid N*K, 256 threads.
constexpr float EXP_CUT = 104.f;  // expf(x) == 0 in fp32 below about -103.97
__global__ __launch_bounds__(256) void attn_pad_terms_kernel(AttnShape s, AttnSaved sv, const float* __restrict__ hinfo,
                                                             const uint8_t* __restrict__ hmask,
                                                             const float* __restrict__ tscale,
                                                             const float* __restrict__ part, float* __restrict__ padpart) {
  __shared__ float s_red[4];
  __shared__ int s_list[256];
  __shared__ float s_wt[256];
  __shared__ int s_n;
  const int nk = blockIdx.x, n = nk / s.K, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int T = s.T, w = s.w;
  float* pd = padpart + (size_t)nk * (w + 4);
  const uint8_t* hm = hmask + (size_t)nk * T;
  const float* sc = tscale + (size_t)n * T;
  float Mv = -INFINITY;
  for (int sp = 0; sp < s.nsplit; ++sp) Mv = fmaxf(Mv, part[((size_t)nk * s.nsplit + sp) * (w + 4)]);
  float zp = -INFINITY;
  if (!sv.allmasked[nk])  // (a fully masked (n,k) runs all its T rows through the main kernel already)
    for (int t = tid; t < T; t += 256)
      if (!hm[t]) zp = fmaxf(zp, tw_logit(FVTA_NEG, sc[t]));
  zp = wave_max(zp);
  if (lane == 0) s_red[wave] = zp;
  __syncthreads();
  zp = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  if (!(zp >= Mv - EXP_CUT)) {  // no masked rows, or every one of them underflows to weight 0
    if (tid == 0) {
      pd[0] = -INFINITY;
      pd[1] = 0.f;
      pd[2] = -INFINITY;
    }
    return;
  }
  float lsum = 0.f;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // channels 4 tid .. (w <= 2048)
  for (int t0 = 0; t0 < T; t0 += 256) {
    __syncthreads();
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int t = t0 + tid;
    if (t < T && !hm[t]) {
      const float z = tw_logit(FVTA_NEG, sc[t]);
      if (z >= zp - EXP_CUT) {
        const int slot = atomicAdd(&s_n, 1);  // (order within the chunk is irrelevant only for the list; the sums
        s_list[slot] = t;                     //  below run over the sorted list: fixed order)
        s_wt[slot] = expf(z - zp);
      }
    }
    __syncthreads();
    const int cnt = s_n;
    if (tid == 0) {  // insertion sort by t: a handful of rows in practice
      for (int i = 1; i < cnt; ++i) {
        const int tv = s_list[i];
        const float wv = s_wt[i];
        int j = i - 1;
        while (j >= 0 && s_list[j] > tv) {
          s_list[j + 1] = s_list[j];
          s_wt[j + 1] = s_wt[j];
          --j;
        }
        s_list[j + 1] = tv;
        s_wt[j + 1] = wv;
      }
    }
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
      const float wt = s_wt[i];
      lsum += wt;
      const float* row = hinfo + ((size_t)nk * T + s_list[i]) * w;
      if (4 * tid < w) acc[0] += ld4g(row + 4 * tid) * wt;
      if (4 * tid + 1024 < w) acc[1] += ld4g(row + 4 * tid + 1024) * wt;
    }
  }
  if (4 * tid < w) *reinterpret_cast<f32x4*>(pd + 4 + 4 * tid) = acc[0];
  if (4 * tid + 1024 < w) *reinterpret_cast<f32x4*>(pd + 4 + 4 * tid + 1024) = acc[1];
  if (tid == 0) {
    pd[0] = zp;
    pd[1] = lsum;
    pd[2] = -INFINITY;
  }
}

// ---- merge: splits -> u[n,k], M, L; softmax over K; h_a.  grid (N, w/256), 256 threads: one channel per thread
constexpr int MERGE_MAXP = 2048;  // K * nsplit partials of one n whose weights are cached in LDS
__global__ __launch_bounds__(256) void attn_merge_kernel(AttnShape s, AttnSaved sv, const float* __restrict__ part,
                                                         const float* __restrict__ padpart, float* __restrict__ h_a) {
  __shared__ float s_M[64], s_Mz[64], s_L[64], s_r[64], s_pw[64];
  __shared__ float s_wt[MERGE_MAXP];  // exp(m_split - Mz_k), 0 for empty splits
  const int n = blockIdx.x, tid = threadIdx.x;
  const int w = s.w, K = s.K, ns = s.nsplit;
  const size_t pstride = (size_t)(w + 4);
  for (int k = tid; k < K; k += 256) {
    const float* pp = part + ((size_t)(n * K + k) * ns) * pstride;
    const float* pd = padpart ? padpart + (size_t)(n * K + k) * pstride : nullptr;
    const bool pad = pd && pd[1] > 0.f;
    float Mz = pad ? pd[0] : -INFINITY, Mu = -INFINITY;
    for (int sp = 0; sp < ns; ++sp) {
      Mz = fmaxf(Mz, pp[(size_t)sp * pstride]);
      Mu = fmaxf(Mu, pp[(size_t)sp * pstride + 2]);
    }
    float L = 0.f;
    for (int sp = 0; sp < ns; ++sp) {
      const float l = pp[(size_t)sp * pstride + 1];
      if (l > 0.f) L += l * expf(pp[(size_t)sp * pstride] - Mz);
    }
    const float pw = pad ? expf(pd[0] - Mz) : 0.f;
    if (pad) L += pd[1] * pw;
    s_M[k] = Mu;
    s_Mz[k] = Mz;
    s_L[k] = L;
    s_pw[k] = pw;
    if (blockIdx.y == 0) {
      sv.M[n * K + k] = Mu;
      sv.Mz[n * K + k] = Mz;
      sv.L[n * K + k] = L;
    }
  }
  __syncthreads();
  if (tid == 0) {  // outer softsel over K (model_v2.py:278): logits = max over (t, j) of the masked logits, unscaled
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s_M[k]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) {
      s_r[k] = expf(s_M[k] - mx);
      sum += s_r[k];
    }
    for (int k = 0; k < K; ++k) {
      s_r[k] /= sum;
      if (blockIdx.y == 0) sv.r[n * K + k] = s_r[k];
    }
  }
  const bool cached = K * ns <= MERGE_MAXP;
  if (cached)
    for (int i = tid; i < K * ns; i += 256) {
      const int k = i / ns;
      const float* pp = part + ((size_t)n * K * ns + i) * pstride;
      s_wt[i] = pp[1] > 0.f ? expf(pp[0] - s_Mz[k]) : 0.f;
    }
  __syncthreads();
  const int c = blockIdx.y * 256 + tid;
  if (c >= w) return;
  float ha = 0.f;
  for (int k = 0; k < K; ++k) {
    const float* pp = part + ((size_t)(n * K + k) * ns) * pstride;
    float u = 0.f;
    for (int sp = 0; sp < ns; ++sp) {
      float wt;
      if (cached) {
        wt = s_wt[k * ns + sp];
      } else {
        wt = pp[(size_t)sp * pstride + 1] > 0.f ? expf(pp[(size_t)sp * pstride] - s_Mz[k]) : 0.f;
      }
      if (wt != 0.f) u += pp[(size_t)sp * pstride + 4 + c] * wt;  // (empty splits hold no vector)
    }
    if (s_pw[k] != 0.f) u += padpart[(size_t)(n * K + k) * pstride + 4 + c] * s_pw[k];
    u /= s_L[k];
    sv.u[((size_t)n * K + k) * w + c] = u;
    ha += s_r[k] * u;
  }
  h_a[(size_t)n * w + c] = ha;
}

// JT = 1 (JQ <= 32) and JT = 2 shapes; the JT = 2 shape of a wide row halves SCW to keep LDS under 160 KB
template <int SCW1, int NSC1, int SCW2, int NSC2, int NSLAB, int NW>
static int launch_main(const AttnFwdArgs& a, hipStream_t stream) {
  const int nitems = a.s.nsplit * a.s.N * a.s.K;
  const dim3 grid(((nitems + 7) / 8) * 8);
  if (a.tscale) {
    if (a.s.JT == 1)
      hipLaunchKernelGGL((attn_fwd_main<SCW1, NSC1, NSLAB, 1, NW, true>), grid, dim3(NW * 64), 0, stream, a);
    else
      hipLaunchKernelGGL((attn_fwd_main<SCW2, NSC2, NSLAB, 2, NW, true>), grid, dim3(NW * 64), 0, stream, a);
  } else if (a.s.JT == 1)
    hipLaunchKernelGGL((attn_fwd_main<SCW1, NSC1, NSLAB, 1, NW, false>), grid, dim3(NW * 64), 0, stream, a);
  else
    hipLaunchKernelGGL((attn_fwd_main<SCW2, NSC2, NSLAB, 2, NW, false>), grid, dim3(NW * 64), 0, stream, a);
  return 0;
}

bool wide_covers(const AttnShape& s, int G);                                     // attn_fwd_wide.hip
int wide_max_run();                                                              // tiles a workgroup of attn_fwd_wide can list
bool launch_attn_fwd_wide(const AttnFwdArgs& a, int G, hipStream_t stream);
bool shadow_covers(const AttnShape& s);                                          // attn_fwd_shadow.hip
void launch_attn_shadow_compact(const AttnShape& s, const AttnSaved& sv, const uint64_t* table, uint64_t* rowptr, hipStream_t stream);
bool launch_attn_fwd_pair16h(const AttnFwdArgs& a, int G, const uint64_t* rowptr, hipStream_t stream);

}  // namespace fvta

using namespace fvta;

// Which forward main kernel runs: the environment (FVTA_ATTN_EXACT, FVTA_ATTN_WAVE16) is read ONCE; tests and A/B
// measurements switch inside a process through fvta_attn_kernel_select.
static int g_attn_exact_override = -1, g_attn_wave16_override = -1;
static int attn_exact_mode() {
  static const int env = [] { const char* e = getenv("FVTA_ATTN_EXACT"); return (e && e[0] == '1') ? 1 : 0; }();
  return g_attn_exact_override >= 0 ? g_attn_exact_override : env;
}
static int attn_wave16_mode() {
  static const int env = [] { const char* e = getenv("FVTA_ATTN_WAVE16"); return e ? atoi(e) : FVTA_ATTN_WAVE16_DEFAULT; }();
  return g_attn_wave16_override >= 0 ? g_attn_wave16_override : env;
}
extern "C" int fvta_attn_kernel_select(int32_t exact, int32_t wave16) {
  g_attn_exact_override = exact < 0 ? -1 : (exact ? 1 : 0);
  g_attn_wave16_override = wave16 < 0 ? -1 : wave16;
  return FVTA_OK;
}

// The fault word of the forward kernels (attn_fwd_shared.h): four bytes of pinned, device-mapped host memory, allocated once.
static int* g_attn_fault_host = nullptr;
int* fvta::attn_fault_word() {
  static int* dev = [] {
    int* h = nullptr;
    int* d = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&h), sizeof(int), hipHostMallocMapped) != hipSuccess) return (int*)nullptr;
    *h = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0) != hipSuccess) return (int*)nullptr;
    g_attn_fault_host = h;
    return d;
  }();
  return dev;
}
// what an earlier launch left there (read without synchronising: reported by the first call that sees it)
static int attn_check_fault() {
  if (g_attn_fault_host && *reinterpret_cast<volatile int*>(g_attn_fault_host) == ATTN_FAULT_PAIR_WAIT) {
    fvta_set_error("attn_fwd_pair16: a wave's partner never arrived at the tile hand-shake (the kernel trapped); results of that launch are invalid");
    return FVTA_ERR_LAUNCH;
  }
  return FVTA_OK;
}

int fvta_attn_check_desc(const fvta_attn_desc* d) {
  FVTA_CHECK_ARG(d != nullptr, "attn: null descriptor");
  FVTA_CHECK_ARG(d->N > 0 && d->K > 0 && d->K <= 64 && d->T > 0, "attn: bad N/K/T (%d,%d,%d), need K<=64", d->N, d->K,
                 d->T);
  FVTA_CHECK_ARG(d->JQ > 0 && d->JQ <= 64, "attn: JQ=%d unsupported (1..64)", d->JQ);
  FVTA_CHECK_ARG(d->simi >= 1 && d->simi <= 4, "similarity matrix not implemented (simiMatrix=%d)", d->simi);
  const int w = d->w;
  FVTA_CHECK_ARG(w == 64 || w == 128 || w == 256 || w == 512 || w == 1024 || w == 2048,
                 "attn: w=%d unsupported; pad the hidden size so that w is one of 64,128,256,512,1024,2048", w);
  FVTA_CHECK_ARG(d->hinfo_stride == 0 || (d->K == 1 && d->hinfo_stride >= (int64_t)d->T * w && d->hinfo_stride % 4 == 0),
                 "attn: hinfo_stride=%lld needs K == 1, >= T*w and a multiple of 4", (long long)d->hinfo_stride);
  return FVTA_OK;
}

extern "C" size_t fvta_attn_saved_bytes(const fvta_attn_desc* d) {
  if (fvta_attn_check_desc(d)) return 0;
  return attn_saved_view(attn_shape(d, true), nullptr).bytes;
}

size_t fvta_attn_bwd_workspace_bytes(const AttnShape& s);  // attn_bwd.hip

constexpr size_t ATTN_WGTAB_BYTES = 4096 * sizeof(uint32_t);  // the pair kernel's workgroup table, behind the partials

extern "C" size_t fvta_attn_workspace_bytes(const fvta_attn_desc* d) {
  if (fvta_attn_check_desc(d)) return 0;
  const AttnShape s = attn_shape(d, true);
  // forward: the split partials, then one more partial per (n,k) for the masked rows' share under time_warp_att
  // (+ the shadow forward's compacted row addresses: two words per row)
  const size_t fwd = fvta_align_up((size_t)s.N * s.K * (s.nsplit + 1) * (s.w + 4) * sizeof(float), 256) + ATTN_WGTAB_BYTES +
                     (size_t)2 * s.N * s.K * s.T * sizeof(uint64_t);
  const size_t bwd = fvta_attn_bwd_workspace_bytes(s);
  return fwd > bwd ? fwd : bwd;
}

extern "C" int fvta_attn_fwd(const fvta_attn_desc* d, const float* hinfo, const float* hq, const uint8_t* hmask,
                             const uint8_t* qmask, const float* W, const float* b, float* h_a, float* a_logits,
                             void* saved, void* workspace, fvta_stream_t stream_) {
  return fvta_attn_fwd_tw(d, hinfo, hq, hmask, qmask, W, b, nullptr, h_a, a_logits, saved, workspace, stream_);
}

static int attn_fwd_impl(const fvta_attn_desc* d, const float* hinfo, const uint64_t* table, const float* hq, const uint8_t* hmask,
                         const uint8_t* qmask, const float* W, const float* b, const float* tscale, float* h_a,
                         float* a_logits, void* saved, void* workspace, fvta_stream_t stream_);

extern "C" int fvta_attn_fwd_tw(const fvta_attn_desc* d, const float* hinfo, const float* hq, const uint8_t* hmask,
                                const uint8_t* qmask, const float* W, const float* b, const float* tscale, float* h_a,
                                float* a_logits, void* saved, void* workspace, fvta_stream_t stream_) {
  FVTA_CHECK_ARG(hinfo != nullptr, "attn_fwd: null pointer");
  return attn_fwd_impl(d, hinfo, nullptr, hq, hmask, qmask, W, b, tscale, h_a, a_logits, saved, workspace, stream_);
}

// The same forward over the encoders' bf16 shadow rows: table[half][(n K + k) T + t] = address of the w/2 bf16 values that
// are channels half w/2 .. of row (n, k, t) (fvta_lstm_shadow_rows; EVERY entry a readable address -- rows no encoder
// writes point at w/2 zeros).  JQ <= 32, w = 512 / 1024, simiMatrix 1-3 only; results equal fvta_attn_fwd's on the
// bf16-rounded rows up to the rounding of the fp16-split logits (tests/test_gpu_shadow.py).
extern "C" int fvta_attn_fwd_shadow(const fvta_attn_desc* d, const uint64_t* table, const float* hq, const uint8_t* hmask,
                                    const uint8_t* qmask, const float* W, const float* b, float* h_a, void* saved,
                                    void* workspace, fvta_stream_t stream_) {
  FVTA_CHECK_ARG(table != nullptr, "attn_fwd_shadow: null pointer");
  return attn_fwd_impl(d, nullptr, table, hq, hmask, qmask, W, b, nullptr, h_a, nullptr, saved, workspace, stream_);
}

static int attn_fwd_impl(const fvta_attn_desc* d, const float* hinfo, const uint64_t* table, const float* hq, const uint8_t* hmask,
                         const uint8_t* qmask, const float* W, const float* b, const float* tscale, float* h_a,
                         float* a_logits, void* saved, void* workspace, fvta_stream_t stream_) {
  if (int e = fvta_attn_check_desc(d)) return e;
  if (int e = attn_check_fault()) return e;
  FVTA_CHECK_ARG((hinfo || table) && hq && h_a && saved && workspace, "attn_fwd: null pointer");
  FVTA_CHECK_ARG(d->simi == 4 || (W && b), "attn_fwd: W and b required for simiMatrix 1-3");
  FVTA_CHECK_ARG(!(tscale && d->hinfo_stride), "attn_fwd: tscale with a strided hinfo is not supported");
  hipStream_t stream = (hipStream_t)stream_;
  const bool use_mask = hmask && qmask;  // model_v2.py:146/233: only when BOTH masks are given
  const AttnShape s = attn_shape(d, use_mask);
  AttnSaved sv = attn_saved_view(s, saved);
  hipLaunchKernelGGL(attn_vecs_kernel, dim3((s.w + 255) / 256), dim3(256), 0, stream, W, s.w, s.simi, s.feat_order,
                     sv.vecs);
  // wide rows (w = 2048: BASELINE.json configs[4]): the rows-stationary, question-streaming kernel of attn_fwd_wide.hip
  const bool exact_kernel = attn_exact_mode() != 0;
  const int wave16_mode = attn_wave16_mode();
  int wideG = (256 + s.N - 1) / s.N;
  if (wideG > s.K * s.nsplit) wideG = s.K * s.nsplit;
  if (wideG < 1) wideG = 1;
  const bool use_wide = !table && !exact_kernel && wave16_mode != 0 && !a_logits && !tscale && !d->hinfo_stride && wide_covers(s, wideG);
  FVTA_CHECK_ARG(!table || (shadow_covers(s) && !d->hinfo_stride),
                 "attn_fwd_shadow: needs JQ <= 32, w = 512 or 1024, simiMatrix 1-3, no hinfo_stride (JQ=%d w=%d simi=%d)", d->JQ, d->w, d->simi);
  hipLaunchKernelGGL(attn_prep_q_kernel, dim3(s.N, s.W4 * s.JP >= 4096 ? 8 : 1), dim3(256), 0, stream, s, sv, hq, qmask, b,
                     use_wide ? 1 : 0);
  hipLaunchKernelGGL(attn_compact_kernel, dim3(s.N * s.K), dim3(256), 0, stream, s, sv, hmask);
  FVTA_CHECK_LAUNCH("attn_prep");
  if (a_logits && use_mask) {
    const size_t n = (size_t)s.N * s.K * s.T * s.JQ;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a_logits, n, FVTA_NEG);
  }
  AttnFwdArgs a;
  a.s = s;
  a.sv = sv;
  a.hinfo = hinfo;
  a.a_logits = a_logits;
  a.part = (float*)workspace;
  a.tscale = tscale;
  a.hstride = d->hinfo_stride ? (size_t)d->hinfo_stride : (size_t)s.T * s.w;
  a.ipw = 1;
  a.wgtab = nullptr;
  a.fault = attn_fault_word();
  a.dbg = fvta_diag_env("FVTA_ATTN_DBG", 0);  // -DFVTA_DIAG builds only
  // (the phase stamps land 32 MiB into the workspace: only where the workspace reaches that far)
  if ((a.dbg & 16) && fvta_attn_workspace_bytes(d) < ((size_t)32 << 20) + 64 * 16 * 8) a.dbg &= ~16;
  // (the bracket files the context attention only: the K = 1 question attention is a 15 us launch of the same kernel)
  const bool prof_it = (size_t)s.N * s.K * s.T >= 65536;
  if (prof_it) fvta_prof_begin(FVTA_PROF_ATTN_FWD_MAIN, stream);
  // (the full logit tensor is an inspection output: only the general kernel writes it)
  // (time_warp_att runs on the general kernel: the 16-row kernel's softmax logits are amax itself)
  const bool rows16 = s.JT == 1 && s.w >= 128 && s.w <= 1024 && !a_logits && !tscale && !d->hinfo_stride &&
                      !exact_kernel;
  // FVTA_ATTN_WAVE16: the one-wave-per-tile kernel for the shapes it covers (measurement switch)
  if (table) {
    int G = (256 + s.N - 1) / s.N;
    const int maxg = (s.K * s.nsplit + 3) / 4;
    if (G > maxg) G = maxg;
    if (G < 1) G = 1;
    const int nwg = s.N * G;
    const size_t part_bytes = fvta_align_up((size_t)s.N * s.K * (s.nsplit + 1) * (s.w + 4) * sizeof(float), 256);
    if (use_mask && s.N <= 64 && s.N > 1 && nwg <= 4096 && maxg <= 255) {
      uint32_t* tab = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + part_bytes);
      hipLaunchKernelGGL(attn_balance_kernel, dim3(1), dim3(64), 0, stream, s, sv, nwg, maxg, tab, 0);
      a.wgtab = tab;
    }
    uint64_t* rowptr = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) + part_bytes + ATTN_WGTAB_BYTES);
    launch_attn_shadow_compact(s, sv, table, rowptr, stream);
    launch_attn_fwd_pair16h(a, G, rowptr, stream);
  } else if (use_wide) {
    const int nwg = s.N * wideG, maxg = s.K * s.nsplit;
    if (use_mask && s.N <= 64 && s.N > 1 && nwg <= 4096 && maxg <= 255) {  // ragged albums: workgroups in proportion to the rows
      uint32_t* tab = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) +
                                                 fvta_align_up((size_t)s.N * s.K * (s.nsplit + 1) * (s.w + 4) * sizeof(float), 256));
      hipLaunchKernelGGL(attn_balance_kernel, dim3(1), dim3(64), 0, stream, s, sv, nwg, maxg, tab, wide_max_run());
      a.wgtab = tab;
    }
    launch_attn_fwd_wide(a, wideG, stream);
  } else if (rows16 && (wave16_mode == 2 || wave16_mode == 3) && s.simi != 4 && s.w >= 512) {  // two waves per tile (attn_fwd_pair16)
    const bool flags = wave16_mode == 3;
    int G = (256 + s.N - 1) / s.N;
    const int maxg = (s.K * s.nsplit + 3) / 4;
    if (G > maxg) G = maxg;
    if (G < 1) G = 1;
    const int nwg = s.N * G;
    const dim3 grid(((nwg + 7) / 8) * 8);
    const int rmode = s.simi == 1 ? 1 : (s.simi == 3 ? 3 : 2);
    // NOTE on rounding: with the table an album's workgroup count -- hence the split points of its partial sums, hence
    // the fp32 rounding of its h_a (nothing else: arg-max positions and logits do not move) -- depends on the OTHER albums
    // of the batch.  The same batch always gives the same bits; an album moved into another batch may differ in the last
    // bits (tests/test_gpu_forward.py::test_attention_pair_kernel_balance_on_skewed_batches).
    if (use_mask && s.N <= 64 && s.N > 1 && nwg <= 4096 && maxg <= 255) {  // ragged albums: workgroups in proportion to the rows
      uint32_t* tab = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) +
                                                 fvta_align_up((size_t)s.N * s.K * (s.nsplit + 1) * (s.w + 4) * sizeof(float), 256));
      hipLaunchKernelGGL(attn_balance_kernel, dim3(1), dim3(64), 0, stream, s, sv, nwg, maxg, tab, 0);
      a.wgtab = tab;
    }
#define FVTA_P16K(NBH, RM, FL)                                                                                          \
  do {                                                                                                                   \
    (void)hipFuncSetAttribute((const void*)attn_fwd_pair16<NBH, RM, FL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((attn_fwd_pair16<NBH, RM, FL>), grid, dim3(512), lds, stream, a, G);                             \
  } while (0)
#define FVTA_P16(NBH)                                                                                                    \
  do {                                                                                                                   \
    const size_t lds = (size_t)2 * 2 * (NBH / 2) * 2 * 64 * 16 + (size_t)2 * s.w * sizeof(float);                        \
    if (flags) {                                                                                                         \
      if (rmode == 1) FVTA_P16K(NBH, 1, true); else if (rmode == 2) FVTA_P16K(NBH, 2, true); else FVTA_P16K(NBH, 3, true); \
    } else {                                                                                                             \
      if (rmode == 1) FVTA_P16K(NBH, 1, false); else if (rmode == 2) FVTA_P16K(NBH, 2, false); else FVTA_P16K(NBH, 3, false); \
    }                                                                                                                    \
  } while (0)
    switch (s.w) {
      case 512: FVTA_P16(16); break;
      case 1024: FVTA_P16(32); break;
    }
#undef FVTA_P16K
#undef FVTA_P16
  } else if (rows16 && wave16_mode && s.simi != 4 && s.w >= 256) {
    int G = (256 + s.N - 1) / s.N;
    const int maxg = (s.K * s.nsplit + 3) / 4;
    if (G > maxg) G = maxg;
    if (G < 1) G = 1;
    const int nwg = s.N * G;
    const dim3 grid(((nwg + 7) / 8) * 8);
    const int rmode = s.simi == 1 ? 1 : (s.simi == 3 ? 3 : 2);
#define FVTA_W16(NBLK)                                                                                                   \
  do {                                                                                                                   \
    const size_t lds = (size_t)2 * (NBLK / 2) * 2 * 64 * 16 + (size_t)2 * s.w * sizeof(float);                           \
    if (rmode == 1) {                                                                                                    \
      (void)hipFuncSetAttribute((const void*)attn_fwd_wave16<NBLK, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((attn_fwd_wave16<NBLK, 1>), grid, dim3(256), lds, stream, a, G);                               \
    } else if (rmode == 2) {                                                                                             \
      (void)hipFuncSetAttribute((const void*)attn_fwd_wave16<NBLK, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((attn_fwd_wave16<NBLK, 2>), grid, dim3(256), lds, stream, a, G);                               \
    } else {                                                                                                             \
      (void)hipFuncSetAttribute((const void*)attn_fwd_wave16<NBLK, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((attn_fwd_wave16<NBLK, 3>), grid, dim3(256), lds, stream, a, G);                               \
    }                                                                                                                    \
  } while (0)
    switch (s.w) {
      case 256: FVTA_W16(16); break;
      case 512: FVTA_W16(32); break;
      case 1024: FVTA_W16(64); break;
    }
#undef FVTA_W16
  } else if (rows16) {
    // a workgroup streams `ipw` consecutive items: about one workgroup per CU, bounded by its LDS row list
    const int nitems = s.nsplit * s.N * s.K;
    const int tiles_per_max = (((s.T + 15) / 16) + s.nsplit - 1) / s.nsplit;
    int cap = R16_MAXT / tiles_per_max;
    if (cap > R16_MAXI) cap = R16_MAXI;
    if (cap < 1) cap = 1;
    int ipw = 1;
    for (int rounds = 1;; ++rounds) {  // whole rounds of 256 workgroups: no half-empty last round
      ipw = (nitems + 256 * rounds - 1) / (256 * rounds);
      if (ipw <= cap) break;
    }
    if (ipw < 1) ipw = 1;
    a.ipw = ipw;
    const int nwg = (nitems + ipw - 1) / ipw;
    const dim3 grid(((nwg + 7) / 8) * 8);
    const int rmode = s.simi == 1 ? 1 : (s.simi == 3 ? 3 : 2);
#define FVTA_R16(NB, NW)                                                                                          \
  do {                                                                                                            \
    if (rmode == 1) hipLaunchKernelGGL((attn_fwd_rows16<NB, NW, 1>), grid, dim3(NW * 64), 0, stream, a);          \
    else if (rmode == 2) hipLaunchKernelGGL((attn_fwd_rows16<NB, NW, 2>), grid, dim3(NW * 64), 0, stream, a);     \
    else hipLaunchKernelGGL((attn_fwd_rows16<NB, NW, 3>), grid, dim3(NW * 64), 0, stream, a);                     \
  } while (0)
    switch (s.w) {
      case 128: FVTA_R16(2, 4); break;
      case 256: FVTA_R16(4, 4); break;
      case 512: FVTA_R16(4, 8); break;
      case 1024: FVTA_R16(8, 8); break;
    }
#undef FVTA_R16
  } else
  switch (s.w) {
    case 64: launch_main<4, 1, 4, 1, 1, 4>(a, stream); break;
    case 128: launch_main<8, 1, 8, 1, 1, 4>(a, stream); break;
    case 256: launch_main<16, 1, 16, 1, 1, 4>(a, stream); break;
    case 512: launch_main<16, 2, 16, 2, 1, 4>(a, stream); break;
    case 1024: launch_main<16, 2, 8, 4, 1, 8>(a, stream); break;
    case 2048: launch_main<16, 2, 8, 4, 2, 8>(a, stream); break;
  }
  if (prof_it) fvta_prof_end(FVTA_PROF_ATTN_FWD_MAIN, 1, stream);
  FVTA_CHECK_LAUNCH("attn_fwd_main");
  float* padpart = nullptr;
  if (tscale && use_mask) {
    padpart = a.part + (size_t)s.N * s.K * s.nsplit * (s.w + 4);
    hipLaunchKernelGGL(attn_pad_terms_kernel, dim3(s.N * s.K), dim3(256), 0, stream, s, sv, hinfo, hmask, tscale, a.part,
                       padpart);
  }
  hipLaunchKernelGGL(attn_merge_kernel, dim3(s.N, (s.w + 255) / 256), dim3(256), 0, stream, s, sv, a.part, padpart, h_a);
  FVTA_CHECK_LAUNCH("attn_merge");
  return FVTA_OK;
}

// The inner softsel result u[n,k,:] (saved by fvta_attn_fwd for the backward) IS attention_keeprank1's output
// (model.py:247-314: softsel over the rows of each (n, m), no softmax over m): copy it out.
extern "C" int fvta_attn_read_u(const fvta_attn_desc* d, const void* saved, float* u_out, fvta_stream_t stream) {
  if (int e = fvta_attn_check_desc(d)) return e;
  FVTA_CHECK_ARG(saved && u_out, "attn_read_u: null pointer");
  const AttnShape s = attn_shape(d, true);
  const AttnSaved sv = attn_saved_view(s, const_cast<void*>(saved));
  FVTA_CHECK_HIP(hipMemcpyAsync(u_out, sv.u, (size_t)s.N * s.K * s.w * sizeof(float), hipMemcpyDeviceToDevice,
                                (hipStream_t)stream));
  return FVTA_OK;
}
