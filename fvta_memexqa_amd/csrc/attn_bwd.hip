// Focal attention backward for gfx950 (gradient of model_v2.py:210-298 / 125-201).
//
// Because the logits are max-pooled over the question axis (model_v2.py:268),
// only ONE (t, j) pair per context row carries gradient, so the backward needs
// no GEMM: it is a single streaming pass (read h row, write dh row) with
//   dh[t]  = p[t] r[k] g  +  dx[t] (Qs[jmax[t]] + Rh + 2 R2 h[t])
//   dQs[j] += dx[t] h[t]   for j = jmax[t]
// A workgroup sorts its chunk of rows by jmax so that every thread can keep
// the dQs accumulator of the "current j" in registers and flush it once per j
// into a private slab (no atomics, bitwise reproducible).
//
// Deviation: gradient through reduce_max goes to the FIRST arg-max
// (TF splits exact ties) and fully masked (n,k) rows pass no gradient into the
// masked logits.
#include "attn_common.h"
#include "fvta_prof.h"

#ifndef FVTA_ATTN_BWD_NT_DEFAULT
#define FVTA_ATTN_BWD_NT_DEFAULT 0
#endif

namespace fvta {

__device__ __forceinline__ f32x4 ld4b(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

constexpr int BWD_CHMAX = 1024;  // rows per backward workgroup chunk (LDS sort capacity)

constexpr int ATTN_BWD_JG = 8;  // question positions per workgroup of the slab fold: two per wave (four, one per wave: no faster)
static inline __host__ __device__ int attn_bwd_jz(const AttnShape& s) { return (s.JQ + ATTN_BWD_JG - 1) / ATTN_BWD_JG; }

struct AttnBwdWork {
  float* coef;   // [N,K]  r / L
  float* gu;     // [N,K]  g . u[k]
  float* dss;    // [N,K]  ds[k] / (#t with amax == M)
  float* slabs;  // [N*ng][bsplit][RH][JP][w]   dQs partials (ng = groups of gk consecutive k per n)
  float* dctp;   // [N*ng][bsplit][RH][JP]      d ct partials
  float* rowp;   // [N*ng][bsplit][RH][2][w]    d Rh, d R2 partials
  float* pvec;   // [N * JZ][5][w]             parameter-vector partials per n and group of 8 question positions
  float* dctn;   // [N][JP]
  float* dscr;   // [N,K,T] time_warp_att: per-row terms of d tscale[n,t] (summed over k in a fixed order afterwards)
  size_t bytes;
  size_t slab_bytes;
};

static inline int bwd_tpr(int W4) { return W4 < 256 ? W4 : 256; }
static inline int bwd_rh(int W4) { return 256 / bwd_tpr(W4); }

static AttnBwdWork bwd_work_view(const AttnShape& s, void* p) {
  FvtaCarver c(p);
  AttnBwdWork v;
  const size_t nk = (size_t)s.N * s.K;
  const int RH = bwd_rh(s.W4);
  v.coef = c.take<float>(nk);
  v.gu = c.take<float>(nk);
  v.dss = c.take<float>(nk);
  v.pvec = c.take<float>((size_t)s.N * attn_bwd_jz(s) * VEC_COUNT * s.w);
  v.dctn = c.take<float>((size_t)s.N * s.JP);
  v.dscr = c.take<float>(nk * s.T);
  const size_t before = c.off;
  const size_t ngr = (size_t)s.N * s.ng;
  v.slabs = c.take<float>(ngr * s.bsplit * RH * s.JP * s.w);
  v.dctp = c.take<float>(ngr * s.bsplit * RH * s.JP);
  v.rowp = c.take<float>(ngr * s.bsplit * RH * 2 * s.w);
  v.slab_bytes = c.off - before;
  v.bytes = c.off;
  return v;
}

// ---- per-(n,k) scalars.  grid N, 1024 threads; a wave per k (no workgroup barrier inside the k loop; 16 waves: a stream's
// dot product and tie count are one chain of load latencies, 10 streams per wave took 20 us at K = 40)
__global__ __launch_bounds__(1024) void attn_bwd_prep_kernel(AttnShape s, AttnSaved sv, AttnBwdWork wk,
                                                            const float* __restrict__ d_h_a) {
  __shared__ float s_gu[64];
  __shared__ int s_ties[64];
  const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int w = s.w, K = s.K, T = s.T;
  const float* g = d_h_a + (size_t)n * w;
  for (int k = wave; k < K; k += (int)(blockDim.x >> 6)) {
    const float* u = sv.u + ((size_t)n * K + k) * w;
    float acc = 0.f;
    for (int c = lane; c < w; c += 64) acc += g[c] * u[c];
    acc = wave_sum(acc);
    const float M = sv.M[n * K + k];
    const float* am = sv.amax + ((size_t)n * K + k) * T;
    int ties = 0;
    for (int t = lane; t < T; t += 64) ties += (am[t] == M) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ties += __shfl_xor(ties, o, 64);
    if (lane == 0) {
      s_gu[k] = acc;
      s_ties[k] = ties;
    }
  }
  // thread k finishes stream k (one thread walked all K with three dependent global loads each: ~20 of the launch's 25 us);
  // the mean is summed in k order by every thread from LDS
  __shared__ float s_r[64];
  float r = 0.f, L = 1.f;
  int am_ = 0;
  if (tid < K) {
    r = sv.r[n * K + tid];
    L = sv.L[n * K + tid];
    am_ = sv.allmasked[n * K + tid];
    s_r[tid] = r;
  }
  __syncthreads();
  if (tid < K) {
    float mean = 0.f;
    for (int k = 0; k < K; ++k) mean += s_r[k] * s_gu[k];
    wk.coef[n * K + tid] = r / L;
    wk.gu[n * K + tid] = s_gu[tid];
    const float ds = r * (s_gu[tid] - mean);
    wk.dss[n * K + tid] = am_ ? 0.f : ds / (float)max(1, s_ties[tid]);
  }
}

struct AttnBwdArgs {
  AttnShape s;
  AttnSaved sv;
  AttnBwdWork wk;
  const float* hinfo;
  const float* d_h_a;
  float* d_hinfo;
  int accumulate;
  const float* tscale;  // [N,T] or null (time_warp_att): the inner softmax ran on z = amax * tscale
  size_t hstride;       // elements between the row blocks of consecutive (n,k) of hinfo / d_hinfo (see AttnFwdArgs)
  int nt;               // bit 0: non-temporal stores of the d_hinfo rows (written once, read much later); bit 1: the h rows too
  const unsigned long long* table;  // SH kernels: [2][N K T] addresses of the rows' bf16 halves (fvta_lstm_shadow_rows), hinfo is null
};

// TPR threads cover one row (16 B each, G float4 per thread when w > 1024);
// RH = 256/TPR row groups; a tile is TR rows, each thread holds TR/RH of them.
// COS: cosine similarity (simi 4) -- a compile-time flag so that the bilinear shapes do not carry its registers.
// ACC: d_hinfo is accumulated into (accumulate == 1).  Compile-time: as a run-time switch every row's store sat behind a
// (possibly executed) load of its destination, i.e. behind an s_waitcnt vmcnt(0) that also waits for every EARLIER row's
// store -- the write stream of the kernel was one store round trip per row.  Without the load (and with the question
// operand of the current j in registers, below) the rows' stores stream.
// (16-row tiles with four workgroups per CU: 0.96 ms against 0.77 at the metric shape -- the 128-register budget spills)
// SH: the rows are read from the encoders' bf16 shadow (two half-rows per row, addressed through a.table: see
// attn_fwd_shadow.hip) -- 8 bytes per thread and row instead of 16; the chunk's 2 x rows addresses are fetched after the
// row sort and kept in LDS.  G == 1, no cosine.
template <int TPR, int G, int TR, bool COS, bool ACC, bool SH = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_main(AttnBwdArgs a) {
  static_assert(!SH || (G == 1 && !COS && TPR >= 2), "shadow rows: bilinear shapes with one float4 per thread and row");
  constexpr int RH = 256 / TPR;
  constexpr int RPT = TR / RH;
  constexpr int WPR = TPR >= 64 ? TPR / 64 : 1;  // waves that share a row
  __shared__ int s_rows[BWD_CHMAX];
  __shared__ uint8_t s_j[BWD_CHMAX];
  __shared__ int s_hist[4][65];
  __shared__ float s_dot[4][TR];
  __shared__ float s_nrm[4][TR];
  __shared__ float s_self[TR];
  __shared__ float s_pr[TR], s_dx[TR];
  __shared__ int s_tt[TR], s_jj[TR];
  __shared__ unsigned long long s_rp[SH ? 2 : 1][SH ? BWD_CHMAX : 1];
  // the workgroup's group of consecutive k: per-k scalars, and the start of each k's rows in the concatenated list
  __shared__ int s_kbase[17];
  __shared__ float s_kM[16], s_kMz[16], s_kcoef[16], s_kgu[16], s_kdss[16];
  __shared__ uint8_t s_kallm[16];

  const AttnShape& s = a.s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int cq = tid % TPR, rh = tid / TPR;
  const int grp = blockIdx.y, n = grp / s.ng, k0 = (grp % s.ng) * s.gk, split = blockIdx.x;
  const int nkl = min(s.gk, s.K - k0);  // k's in this group
  const int nk = n * s.K + k0;          // first (n,k) of the group; rows of the group are contiguous from here
  const int T = s.T, w = s.w, JP = s.JP;
  if (tid == 0) {
    int run = 0;
    for (int kl = 0; kl < nkl; ++kl) {
      s_kbase[kl] = run;
      run += a.sv.cnt[nk + kl];
    }
    s_kbase[nkl] = run;
  }
  if (tid < nkl) {
    s_kM[tid] = a.sv.M[nk + tid];
    s_kMz[tid] = a.sv.Mz[nk + tid];
    s_kcoef[tid] = a.wk.coef[nk + tid];
    s_kgu[tid] = a.wk.gu[nk + tid];
    s_kdss[tid] = a.wk.dss[nk + tid];
    s_kallm[tid] = a.sv.allmasked[nk + tid] != 0;
  }
  __syncthreads();
  const int cnt = s_kbase[nkl];
  int chunk = (cnt + s.bsplit - 1) / s.bsplit;
  chunk = (chunk + TR - 1) / TR * TR;
  const int r0 = split * chunk, r1 = min(cnt, r0 + chunk);
  if (r0 >= r1) return;  // slabs were zeroed by the launcher
  const int nrows = r1 - r0;
  const int32_t* __restrict__ idx = a.sv.idx + (size_t)nk * T;
  const float* __restrict__ amax = a.sv.amax + (size_t)nk * T;
  const uint8_t* __restrict__ jmax = a.sv.jmax + (size_t)nk * T;
  // row i of the concatenated list -> flat row (k - k0) * T + t of the group
  auto flat_row = [&](int i) {
    int kl = 0;
    while (kl + 1 < nkl && i >= s_kbase[kl + 1]) ++kl;
    return kl * T + idx[(size_t)kl * T + (i - s_kbase[kl])];
  };

  // ---- stable counting sort of the chunk's rows by jmax (4 waves, contiguous quarters)
  for (int i = tid; i < 4 * 65; i += 256) (&s_hist[0][0])[i] = 0;
  __syncthreads();
  const int q = (nrows + 3) / 4;
  const int b0 = wave * q, b1 = min(nrows, b0 + q);
  for (int base = b0; base < b1; base += 64) {
    const int i = base + lane;
    const bool ok = i < b1;
    const int key = ok ? (int)jmax[flat_row(r0 + i)] : -1;
    unsigned long long todo = __ballot(ok);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int kv = __shfl(key, leader, 64);
      const unsigned long long same = __ballot(ok && key == kv);
      if (lane == leader) s_hist[wave][kv] += __popcll(same);
      todo &= ~same;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k < 64; ++k)
      for (int wv = 0; wv < 4; ++wv) {
        const int c = s_hist[wv][k];
        s_hist[wv][k] = run;
        run += c;
      }
  }
  __syncthreads();
  for (int base = b0; base < b1; base += 64) {
    const int i = base + lane;
    const bool ok = i < b1;
    const int t = ok ? flat_row(r0 + i) : 0;
    const int key = ok ? (int)jmax[t] : -1;
    unsigned long long todo = __ballot(ok);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int kv = __shfl(key, leader, 64);
      const unsigned long long same = __ballot(ok && key == kv);
      const int basepos = s_hist[wave][kv];
      if (ok && key == kv) {
        const int pos = basepos + __popcll(same & ((1ull << lane) - 1ull));
        s_rows[pos] = t;
        s_j[pos] = (uint8_t)kv;
      }
      __builtin_amdgcn_wave_barrier();
      if (lane == leader) s_hist[wave][kv] = basepos + __popcll(same);
      __builtin_amdgcn_wave_barrier();
      todo &= ~same;
    }
  }
  __syncthreads();

  if constexpr (SH) {  // the sorted rows' addresses (independent loads: inside the sort's serial loop each one was a round trip)
    const unsigned long long* __restrict__ tab0 = a.table + (size_t)nk * T;
    const unsigned long long* __restrict__ tab1 = tab0 + (size_t)s.N * s.K * T;
    for (int i = tid; i < nrows; i += 256) {
      const int t = s_rows[i];
      s_rp[0][i] = tab0[t];
      s_rp[1][i] = tab1[t];
    }
    // (the main loop's first __syncthreads() orders these writes before the tiles' reads)
  }
  const float* __restrict__ hbase = SH ? nullptr : a.hinfo + (size_t)nk * a.hstride;
  // SH: this thread's four channels lie in half `shalf` of a row, `soff` bytes into it
  const int shalf = 4 * cq >= w / 2 ? 1 : 0, soff = 2 * (4 * cq - shalf * (w / 2));
  float* __restrict__ dhbase = a.d_hinfo + (size_t)nk * a.hstride;
  const float* __restrict__ Qs = a.sv.Qs + (size_t)n * s.W4 * JP * 4;
  constexpr bool cosine = COS;
  const size_t slot = ((size_t)grp * s.bsplit + split) * RH + rh;
  float* __restrict__ slab = a.wk.slabs + slot * JP * w;
  float* __restrict__ dctp = a.wk.dctp + slot * JP;
  float* __restrict__ rowp = a.wk.rowp + slot * 2 * w;

  f32x4 gv[G], rh4[G], r24[G], accq[G], accRh[G], accR2[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int c4 = cq + g * TPR;
    gv[g] = ld4b(a.d_h_a + (size_t)n * w + 4 * c4);
    rh4[g] = ld4b(a.sv.vecs + VEC_RH * w + 4 * c4);
    r24[g] = ld4b(a.sv.vecs + VEC_R2 * w + 4 * c4);
    accq[g] = accRh[g] = accR2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int cur_j = -1;
  float acc_ct = 0.f;
  f32x4 qs_cur[G];
#pragma unroll
  for (int g = 0; g < G; ++g) qs_cur[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto flush = [&]() {
    if (cur_j >= 0) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        *reinterpret_cast<f32x4*>(slab + (size_t)cur_j * w + 4 * (cq + g * TPR)) = accq[g];
        accq[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (cq == 0) dctp[cur_j] = acc_ct;
      acc_ct = 0.f;
    }
  };

  typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
  f32x4 hreg[SH ? 1 : RPT][G];
  u32x2s hraw[SH ? RPT : 1];  // SH: the tile stays packed (half the registers), unpacked at each use
  auto row_val = [&](int i, int g) {
    if constexpr (SH) {
      u32x2s raw = hraw[i];
      asm volatile("" : "+v"(raw));  // (unpacked again at each use: the compiler would keep the fp32 copy live instead)
      return f32x4{__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u), __uint_as_float(raw[1] << 16),
                   __uint_as_float(raw[1] & 0xffff0000u)};
    } else {
      return hreg[i][g];
    }
  };
  for (int tb = 0; tb < nrows; tb += TR) {
    __syncthreads();
    if (tid < TR) {
      const int i = tb + tid;
      s_tt[tid] = i < nrows ? s_rows[i] : -1;
      s_jj[tid] = i < nrows ? (int)s_j[i] : 0;
    }
    __syncthreads();
    float dot[RPT], nrm[RPT];
    if constexpr (SH) {
      // every row of the tile is requested before the first one is used (row_val's empty asm pins each use: in one loop
      // with the loads the compiler waited vmcnt(0) row by row -- 32 dependent round trips per tile)
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        // (a GLOBAL pointer: through a generic one the load is a FLAT instruction, which counts on lgkmcnt too)
        typedef const u32x2s __attribute__((address_space(1)))* grow8_ptr;
        hraw[i] = *(grow8_ptr)(s_rp[shalf][min(tb + rh * RPT + i, nrows - 1)] + soff);  // (padding rows of a tile: the last row's address)
      }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int t = s_tt[rh * RPT + i];
      float dsum = 0.f, nsum = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if constexpr (SH) {
          hraw[i] = t >= 0 ? hraw[i] : u32x2s{0u, 0u};
        } else {
          const f32x4* hp = reinterpret_cast<const f32x4*>(hbase + (size_t)max(t, 0) * w + 4 * (cq + g * TPR));
          const f32x4 hv = (a.nt & 2) ? __builtin_nontemporal_load(hp) : *hp;  // unconditional, then zeroed
          hreg[i][g] = t >= 0 ? hv : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const f32x4 hcur = row_val(i, g);
        const f32x4 pdt = hcur * gv[g];
        dsum += (pdt[0] + pdt[1]) + (pdt[2] + pdt[3]);
        if (cosine) {
          const f32x4 hh = hcur * hcur;
          nsum += (hh[0] + hh[1]) + (hh[2] + hh[3]);
        }
      }
      dot[i] = dsum;
      nrm[i] = nsum;
    }
    // reduce g.h over the TPR threads of each row
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      float v = dot[i];
#pragma unroll
      for (int o = 1; o < (TPR < 64 ? TPR : 64); o <<= 1) v += __shfl_xor(v, o, 64);
      dot[i] = v;
      if (cosine) {
        float u = nrm[i];
#pragma unroll
        for (int o = 1; o < (TPR < 64 ? TPR : 64); o <<= 1) u += __shfl_xor(u, o, 64);
        nrm[i] = u;
      }
    }
    if (TPR >= 64) {
      if (lane == 0)
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          s_dot[wave][rh * RPT + i] = dot[i];
          if (cosine) s_nrm[wave][rh * RPT + i] = nrm[i];
        }
    } else {
      if (cq == 0)
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          s_dot[0][rh * RPT + i] = dot[i];
          if (cosine) s_nrm[0][rh * RPT + i] = nrm[i];
        }
    }
    __syncthreads();
    if (tid < TR) {
      const int t = s_tt[tid];
      float pr = 0.f, dx = 0.f, self = 0.f;
      if (t >= 0) {
        float gh, hh = 0.f;
        if (TPR >= 64) {
          // row `tid` belongs to row group tid / RPT whose waves are [rg*WPR, rg*WPR + WPR)
          const int rgp = tid / RPT;
          gh = 0.f;
#pragma unroll
          for (int v = 0; v < WPR; ++v) {
            gh += s_dot[rgp * WPR + v][tid];
            if (cosine) hh += s_nrm[rgp * WPR + v][tid];
          }
        } else {
          gh = s_dot[0][tid];
          if (cosine) hh = s_nrm[0][tid];
        }
        const int kl = t / T;
        const float M = s_kM[kl];
        const float am = amax[t];
        // z = am * tscale[n,t] under time_warp_att (model_v2.py:269-275); p = softmax_t(z), d z = p (g.h - g.u)
        const float sc = a.tscale ? a.tscale[(size_t)n * T + (t - kl * T)] : 1.f;
        pr = expf(tw_logit(am, sc) - s_kMz[kl]) * s_kcoef[kl];
        const float dz = pr * (gh - s_kgu[kl]);
        if (a.tscale) a.wk.dscr[(size_t)nk * T + t] = dz * am;  // d tscale[n,t] += d z * amax (also for -1e30 rows, as TF)
        if (!s_kallm[kl]) {  // fully masked rows: am = -1e30, no gradient into the masked logits
          const float damax = dz * sc + (am == M ? s_kdss[kl] : 0.f);
          dx = s.add_tanh ? damax * (1.f - am * am) : damax;
          if (cosine) {  // x = (h.qn) * rh with rh = rsqrt(max(|h|^2, eps)) (model_v2.py:250-254): am IS x
            const float rhn = rsqrtf(fmaxf(hh, 1e-12f));
            self = hh > 1e-12f ? -damax * am * rhn * rhn : 0.f;  // d/dh of the normaliser
            dx = damax * rhn;                                      // coefficient of qn[j] (and of h in dQs)
          }
        }
      }
      s_pr[tid] = pr;
      s_dx[tid] = dx;
      if (cosine) s_self[tid] = self;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int row = rh * RPT + i;
      const int t = s_tt[row];
      if (t < 0) continue;
      const int j = s_jj[row];
      const float pr = s_pr[row], dx = s_dx[row], self = cosine ? s_self[row] : 0.f;
      if (j != cur_j) {  // (rows come sorted by j: at most JQ changes per chunk)
        flush();
        cur_j = j;
        // The question operand of the current j stays in registers.  As a per-row load it was a gather of 64 cache lines per
        // wave-instruction (consecutive threads lie JP * 16 bytes apart): 107 M line requests per launch against 28 M for
        // the rows themselves, and the CU's address unit takes ~0.26 lines per clock -- that gather WAS the kernel's time.
        // (The empty asm makes the load's wait happen HERE, inside the branch: left pending across the merge point the
        // compiler must wait vmcnt(0) at every row's first use, which drains the row stores one by one.)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          qs_cur[g] = ld4b(Qs + ((size_t)(cq + g * TPR) * JP + j) * 4);
          asm volatile("" : "+v"(qs_cur[g]));
        }
      }
      acc_ct += dx;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int c4 = cq + g * TPR;
        const f32x4 h = row_val(i, g);
#ifdef FVTA_ATTN_BWD_QS_PER_ROW  // (A/B: the per-row gather this replaced)
        const f32x4 qs = ld4b(Qs + ((size_t)c4 * JP + j) * 4);
#else
        const f32x4 qs = qs_cur[g];
#endif
        f32x4 dh = cosine ? gv[g] * pr + qs * dx + h * self : gv[g] * pr + (qs + rh4[g] + r24[g] * h * 2.f) * dx;
        float* dst = dhbase + (size_t)t * w + 4 * c4;
        if constexpr (ACC) dh += ld4b(dst);
        if ((a.nt & 1) && !ACC)
          __builtin_nontemporal_store(dh, reinterpret_cast<f32x4*>(dst));
        else
          *reinterpret_cast<f32x4*>(dst) = dh;
        accq[g] += h * dx;
        if (!cosine) {
          accRh[g] += h * dx;
          accR2[g] += h * h * dx;
        }
      }
    }
  }
  flush();
#pragma unroll
  for (int g = 0; g < G; ++g) {
    *reinterpret_cast<f32x4*>(rowp + 4 * (cq + g * TPR)) = accRh[g];
    *reinterpret_cast<f32x4*>(rowp + w + 4 * (cq + g * TPR)) = accR2[g];
  }
}

// ---- fold slabs per n: d_hq and per-n parameter partials.  grid (N, ceil(w/256), ceil(JQ/8)): a block takes 8 question
// positions x 256 channels; a lane owns 4 channels (16-byte loads), wave v the positions j_lo + v and j_lo + v + 4, and the
// slots of a position are loaded eight at a time before they are summed in slot order (one thread per channel walking its
// slots one dependent 4-byte load after the other: 105 us at the metric shape, 0.6 TB/s)
__global__ __launch_bounds__(256) void attn_bwd_reduce_q_kernel(AttnShape s, AttnSaved sv, AttnBwdWork wk, int RH,
                                                                const float* __restrict__ hq,
                                                                float* __restrict__ d_hq, int accumulate) {
  __shared__ float s_dct[64];
  __shared__ f32x4 s_p[3][3][64];  // waves 1..3 hand their partial pU / pCq / pC2 to wave 0
  const int n = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.y * 256 + 4 * lane;
  const int w = s.w, JP = s.JP, JQ = s.JQ;
  const int nslot = s.ng * s.bsplit * RH;
  const size_t slot0 = (size_t)n * nslot;
  if (threadIdx.x < JP) {
    float acc = 0.f;
    int sl = 0;
    for (; sl + 8 <= nslot; sl += 8) {  // eight slots in flight, summed in slot order
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = wk.dctp[(slot0 + sl + i) * JP + threadIdx.x];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
    for (; sl < nslot; ++sl) acc += wk.dctp[(slot0 + sl) * JP + threadIdx.x];
    s_dct[threadIdx.x] = acc;
    if (blockIdx.y == 0 && blockIdx.z == 0) wk.dctn[(size_t)n * JP + threadIdx.x] = acc;
  }
  __syncthreads();
  const bool live = c < w;  // (w is a multiple of 4: a lane's four channels are inside or outside together)
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  auto ld4 = [&](const float* p) { return live ? *reinterpret_cast<const f32x4*>(p) : zero; };
  // slots summed in slot order, eight loads in flight
  auto fold = [&](const float* base, size_t stride) {
    f32x4 acc = zero;
    int sl = 0;
    for (; sl + 8 <= nslot; sl += 8) {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = ld4(base + (size_t)(sl + i) * stride);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
    for (; sl < nslot; ++sl) acc += ld4(base + (size_t)sl * stride);
    return acc;
  };
  const f32x4 U = ld4(sv.vecs + VEC_U * w + c), Cq = ld4(sv.vecs + VEC_CQ * w + c), C2 = ld4(sv.vecs + VEC_C2 * w + c);
  f32x4 pU = zero, pCq = zero, pC2 = zero;
  const int jz = blockIdx.z, j_lo = jz * ATTN_BWD_JG, j_hi = min(JQ, j_lo + ATTN_BWD_JG);
  for (int j = j_lo + wave; j < j_hi; j += 4) {
    const f32x4 dQ = fold(wk.slabs + (slot0 * JP + j) * w + c, (size_t)JP * w);
    const f32x4 qv = ld4(hq + ((size_t)n * JQ + j) * w + c);
    const float dct = s_dct[j];
    const f32x4 dq = U * dQ + dct * (Cq + 2.f * C2 * qv);
    if (live) {
      f32x4* dst = reinterpret_cast<f32x4*>(d_hq + ((size_t)n * JQ + j) * w + c);
      *dst = accumulate != 0 ? *dst + dq : dq;
    }
    pU += dQ * qv;
    pCq += dct * qv;
    pC2 += dct * qv * qv;
  }
  if (wave) {
    s_p[wave - 1][0][lane] = pU;
    s_p[wave - 1][1][lane] = pCq;
    s_p[wave - 1][2][lane] = pC2;
  }
  __syncthreads();
  if (wave || !live) return;
#pragma unroll
  for (int v = 0; v < 3; ++v) {  // wave order: positions j_lo + v (+ 4) after j_lo (+ 4)
    pU += s_p[v][0][lane];
    pCq += s_p[v][1][lane];
    pC2 += s_p[v][2][lane];
  }
  f32x4 pRh = zero, pR2 = zero;
  if (jz == 0) {
    pRh = fold(wk.rowp + slot0 * 2 * w + c, (size_t)2 * w);
    pR2 = fold(wk.rowp + slot0 * 2 * w + w + c, (size_t)2 * w);
  }
  float* pv = wk.pvec + ((size_t)n * gridDim.z + jz) * VEC_COUNT * w;
  *reinterpret_cast<f32x4*>(pv + VEC_U * w + c) = pU;
  *reinterpret_cast<f32x4*>(pv + VEC_RH * w + c) = pRh;
  *reinterpret_cast<f32x4*>(pv + VEC_R2 * w + c) = pR2;
  *reinterpret_cast<f32x4*>(pv + VEC_CQ * w + c) = pCq;
  *reinterpret_cast<f32x4*>(pv + VEC_C2 * w + c) = pC2;
}

// ---- cosine similarity (simi 4): d_hq from the dqn slabs; qn = q * rq, rq = rsqrt(max(|q|^2, eps)).
// grid N*JQ, 256 threads
__global__ __launch_bounds__(256) void attn_bwd_cosine_q_kernel(AttnShape s, AttnBwdWork wk, int RH,
                                                                const float* __restrict__ hq,
                                                                float* __restrict__ d_hq, int accumulate) {
  __shared__ float s_red[2][4];
  const int n = blockIdx.x / s.JQ, j = blockIdx.x % s.JQ, tid = threadIdx.x;
  const int w = s.w, JP = s.JP;
  const int nslot = s.ng * s.bsplit * RH;
  const size_t slot0 = (size_t)n * nslot;
  const float* q = hq + ((size_t)n * s.JQ + j) * w;
  float qq = 0.f, qd = 0.f;
  for (int c = tid; c < w; c += 256) {
    float dQ = 0.f;
    for (int sl = 0; sl < nslot; ++sl) dQ += wk.slabs[((slot0 + sl) * JP + j) * w + c];
    const float qv = q[c];
    qq += qv * qv;
    qd += qv * dQ;
  }
  qq = wave_sum(qq);
  qd = wave_sum(qd);
  if ((tid & 63) == 0) {
    s_red[0][tid >> 6] = qq;
    s_red[1][tid >> 6] = qd;
  }
  __syncthreads();
  qq = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
  qd = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
  const float rq = rsqrtf(fmaxf(qq, 1e-12f));
  const float proj = qq > 1e-12f ? qd * rq * rq : 0.f;  // (qn . dqn) rq, folded
  for (int c = tid; c < w; c += 256) {
    float dQ = 0.f;
    for (int sl = 0; sl < nslot; ++sl) dQ += wk.slabs[((slot0 + sl) * JP + j) * w + c];
    const float dq = (dQ - proj * q[c]) * rq;
    float* dst = d_hq + ((size_t)n * s.JQ + j) * w + c;
    *dst = accumulate != 0 ? *dst + dq : dq;
  }
}

// ---- sum over n, map back to att_logits/W's layout.  grid ceil(w/64) + 1; 256 threads = 64 channels x 4 n-groups
// (fixed summation order: bitwise reproducible)
__global__ __launch_bounds__(256) void attn_bwd_params_kernel(AttnShape s, AttnBwdWork wk, float* __restrict__ dW,
                                                              float* __restrict__ db) {
  __shared__ float s_p[4][VEC_COUNT][64];
  __shared__ float s_b[4];
  const int tid = threadIdx.x, w = s.w;
  if (blockIdx.x == gridDim.x - 1) {  // the bias block
    float acc = 0.f;
    for (int i = tid; i < s.N * s.JP; i += 256) acc += wk.dctn[i];
    acc = wave_sum(acc);
    if ((tid & 63) == 0) s_b[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0 && db) db[0] += (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
    return;
  }
  const int cl = tid & 63, grp = tid >> 6, c = blockIdx.x * 64 + cl;
  float v[VEC_COUNT] = {0, 0, 0, 0, 0};
  if (c < w) {
    const int rows = s.N * attn_bwd_jz(s);   // rows = (n, group of question positions)
    int n = grp;
    for (; n + 12 < rows; n += 16) {         // four rows per round, their 20 loads in flight together; same order of sums
      float x[4][VEC_COUNT];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < VEC_COUNT; ++k) x[i][k] = wk.pvec[((size_t)(n + 4 * i) * VEC_COUNT + k) * w + c];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < VEC_COUNT; ++k) v[k] += x[i][k];
    }
    for (; n < rows; n += 4)
#pragma unroll
      for (int k = 0; k < VEC_COUNT; ++k) v[k] += wk.pvec[((size_t)n * VEC_COUNT + k) * w + c];
  }
#pragma unroll
  for (int k = 0; k < VEC_COUNT; ++k) s_p[grp][k][cl] = v[k];
  __syncthreads();
  if (grp == 0 && c < w) {
#pragma unroll
    for (int k = 0; k < VEC_COUNT; ++k) v[k] = (s_p[0][k][cl] + s_p[1][k][cl]) + (s_p[2][k][cl] + s_p[3][k][cl]);
    const float dU = v[VEC_U], dRh = v[VEC_RH], dR2 = v[VEC_R2], dCq = v[VEC_CQ], dC2 = v[VEC_C2];
    if (s.simi == 1) {
      dW[c] += dRh;
      dW[w + c] += dCq;
      dW[2 * w + c] += dU;
    } else if (s.simi == 2) {
      const float d1 = dU, d2 = -2.f * dU + dR2 + dC2;
      dW[c] += s.feat_order == 0 ? d1 : d2;
      dW[w + c] += s.feat_order == 0 ? d2 : d1;
    } else if (s.simi == 3) {
      dW[c] += dRh;
      dW[w + c] += dCq;
      dW[2 * w + c] += -2.f * dU + dR2 + dC2;
      dW[3 * w + c] += dU;
    }
  }
}

// ---- time_warp_att: the masked rows that carry softmax weight (see attn_pad_terms_kernel in attn_fwd.hip): their
// direct term p r g into d_hinfo and their share of d tscale (d z * -1e30, as TF computes it).  No gradient into the
// masked logits themselves (deviation (b)).  grid N*K, 256 threads.
__global__ __launch_bounds__(256) void attn_bwd_pad_kernel(AttnBwdArgs a, const uint8_t* __restrict__ hmask) {
  __shared__ float s_red[4];
  __shared__ int s_list[256];
  __shared__ int s_n;
  const AttnShape& s = a.s;
  const int nk = blockIdx.x, n = nk / s.K, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int T = s.T, w = s.w;
  if (a.sv.allmasked[nk]) return;
  const uint8_t* hm = hmask + (size_t)nk * T;
  const float* sc = a.tscale + (size_t)n * T;
  const float Mz = a.sv.Mz[nk], L = a.sv.L[nk], coef = a.wk.coef[nk];
  const float* g = a.d_h_a + (size_t)n * w;
  // Two passes over the masked rows that carry weight.  Pass 0 sums their weights and weighted g.h; when they hold
  // ALL of the softmax's weight (the usual case: exp() of every valid row's logit underflows against +1e30) the mean
  // g.u the softmax gradient subtracts is taken from these very sums, so that a single winner gets d z = 0 EXACTLY, as
  // TF's p (g - sum p g) does -- it is multiplied by -1e30 on its way into d tscale.
  float wsum = 0.f, gsum = 0.f, gbar = a.wk.gu[nk];
  for (int pass = 0; pass < 2; ++pass) {
    for (int t0 = 0; t0 < T; t0 += 256) {
      __syncthreads();
      if (tid == 0) s_n = 0;
      __syncthreads();
      const int t = t0 + tid;
      const bool on = t < T && !hm[t] && tw_logit(FVTA_NEG, sc[t]) >= Mz - 104.f;
      // ascending t within the chunk (ballot ranks), chunks in order: the sums below run in a fixed order
      const unsigned long long bal = __ballot(on);
      if (lane == 0) s_red[wave] = (float)__popcll(bal);
      __syncthreads();
      int base = 0;
      for (int v = 0; v < wave; ++v) base += (int)s_red[v];
      if (on) s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = t;
      if (tid == 0) s_n = (int)(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
      __syncthreads();
      const int cnt = s_n;
      for (int i = 0; i < cnt; ++i) {
        const int tt = s_list[i];
        const float wt = expf(tw_logit(FVTA_NEG, sc[tt]) - Mz);
        const float pr = wt * coef;
        const float* row = a.hinfo + ((size_t)nk * T + tt) * w;
        float* drow = a.d_hinfo + ((size_t)nk * T + tt) * w;
        float dot = 0.f;
        for (int c = 4 * tid; c < w; c += 1024) {
          const f32x4 hv = ld4b(row + c), gv = ld4b(g + c);
          const f32x4 pdt = hv * gv;
          dot += (pdt[0] + pdt[1]) + (pdt[2] + pdt[3]);
          if (pass == 1) {
            f32x4 dh = gv * pr;
            if (a.accumulate != 2) dh += ld4b(drow + c);  // modes 0/3 zeroed the row first, mode 1 accumulates; 2: plain store
            *reinterpret_cast<f32x4*>(drow + c) = dh;
          }
        }
        dot = wave_sum(dot);
        __syncthreads();
        if (lane == 0) s_red[wave] = dot;
        __syncthreads();
        dot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        if (pass == 0) {
          wsum += wt;
          gsum += wt * dot;
        } else if (tid == 0) {
          a.wk.dscr[(size_t)nk * T + tt] = pr * (dot - gbar) * FVTA_NEG;
        }
      }
    }
    if (pass == 0 && wsum == L) gbar = gsum / L;  // the masked rows hold all the weight
  }
}

// accumulate 0 / 3: the rows the main kernel does not write -- the MASKED rows -- are zeros.  (A memset of the whole gradient
// tensor before a kernel that overwrites every valid row of it was 1.9 GB of stores per step at the metric shape, 0.3 ms.)
// A wave per row, grid ceil(N K T / 4)
__global__ __launch_bounds__(256) void attn_zero_masked_rows_kernel(AttnShape s, const uint8_t* __restrict__ hmask,
                                                                    float* __restrict__ d_hinfo) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (size_t)s.N * s.K * s.T || hmask[row]) return;
  f32x4* dst = reinterpret_cast<f32x4*>(d_hinfo + row * s.w);
  for (int c = threadIdx.x & 63; c < s.w / 4; c += 64) dst[c] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// d_tscale[n,t] += sum_k dscr[n,k,t], fixed order
__global__ void attn_bwd_dscale_kernel(AttnShape s, const float* __restrict__ dscr, float* __restrict__ d_tscale) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= s.N * s.T) return;
  const int n = pos / s.T, t = pos % s.T;
  float acc = 0.f;
  for (int k = 0; k < s.K; ++k) acc += dscr[((size_t)n * s.K + k) * s.T + t];
  d_tscale[pos] += acc;
}

}  // namespace fvta

using namespace fvta;

size_t fvta_attn_bwd_workspace_bytes(const AttnShape& s) { return bwd_work_view(s, nullptr).bytes; }
int fvta_attn_check_desc(const fvta_attn_desc* d);  // attn_fwd.hip

extern "C" int fvta_attn_bwd(const fvta_attn_desc* d, const float* hinfo, const float* hq, const uint8_t* hmask,
                             const uint8_t* qmask, const float* W, const float* b, const float* d_h_a,
                             const void* saved, float* d_hinfo, float* d_hq, float* dW, float* db, int accumulate,
                             void* workspace, fvta_stream_t stream_) {
  return fvta_attn_bwd_tw(d, hinfo, hq, hmask, qmask, W, b, nullptr, d_h_a, saved, d_hinfo, d_hq, dW, db, nullptr,
                          accumulate, workspace, stream_);
}

static int attn_bwd_impl(const fvta_attn_desc* d, const float* hinfo, const uint64_t* table, const float* hq, const uint8_t* hmask,
                         const uint8_t* qmask, const float* W, const float* b, const float* tscale,
                         const float* d_h_a, const void* saved, float* d_hinfo, float* d_hq, float* dW, float* db,
                         float* d_tscale, int accumulate, void* workspace, fvta_stream_t stream_);

extern "C" int fvta_attn_bwd_tw(const fvta_attn_desc* d, const float* hinfo, const float* hq, const uint8_t* hmask,
                                const uint8_t* qmask, const float* W, const float* b, const float* tscale,
                                const float* d_h_a, const void* saved, float* d_hinfo, float* d_hq, float* dW, float* db,
                                float* d_tscale, int accumulate, void* workspace, fvta_stream_t stream_) {
  FVTA_CHECK_ARG(hinfo != nullptr, "attn_bwd: null pointer");
  return attn_bwd_impl(d, hinfo, nullptr, hq, hmask, qmask, W, b, tscale, d_h_a, saved, d_hinfo, d_hq, dW, db, d_tscale,
                       accumulate, workspace, stream_);
}

// The backward of fvta_attn_fwd_shadow: the rows come from the same table of bf16 half-row addresses; d_hinfo is the
// fp32 [N,K,T,w] tensor as in fvta_attn_bwd.
extern "C" int fvta_attn_bwd_shadow(const fvta_attn_desc* d, const uint64_t* table, const float* hq, const uint8_t* hmask,
                                    const uint8_t* qmask, const float* W, const float* b, const float* d_h_a,
                                    const void* saved, float* d_hinfo, float* d_hq, float* dW, float* db, int accumulate,
                                    void* workspace, fvta_stream_t stream_) {
  FVTA_CHECK_ARG(table != nullptr, "attn_bwd_shadow: null pointer");
  FVTA_CHECK_ARG(d && d->JQ <= 32 && (d->w == 512 || d->w == 1024) && d->simi != 4 && !d->hinfo_stride,
                 "attn_bwd_shadow: needs JQ <= 32, w = 512 or 1024, simiMatrix 1-3, no hinfo_stride");
  return attn_bwd_impl(d, nullptr, table, hq, hmask, qmask, W, b, nullptr, d_h_a, saved, d_hinfo, d_hq, dW, db, nullptr,
                       accumulate, workspace, stream_);
}

static int attn_bwd_impl(const fvta_attn_desc* d, const float* hinfo, const uint64_t* table, const float* hq, const uint8_t* hmask,
                         const uint8_t* qmask, const float* W, const float* b, const float* tscale,
                         const float* d_h_a, const void* saved, float* d_hinfo, float* d_hq, float* dW, float* db,
                         float* d_tscale, int accumulate, void* workspace, fvta_stream_t stream_) {
  if (int e = fvta_attn_check_desc(d)) return e;
  FVTA_CHECK_ARG((tscale == nullptr) == (d_tscale == nullptr), "attn_bwd: tscale and d_tscale go together");
  FVTA_CHECK_ARG((hinfo || table) && hq && d_h_a && saved && d_hinfo && d_hq && workspace, "attn_bwd: null pointer");
  FVTA_CHECK_ARG(d->simi == 4 || (dW && db), "attn_bwd: dW/db required");
  FVTA_CHECK_ARG(!(tscale && d->hinfo_stride), "attn_bwd: tscale with a strided hinfo is not supported");
  hipStream_t stream = (hipStream_t)stream_;
  const bool use_mask = hmask && qmask;
  const AttnShape s = attn_shape(d, use_mask);
  AttnSaved sv = attn_saved_view(s, const_cast<void*>(saved));
  AttnBwdWork wk = bwd_work_view(s, workspace);
  const int RH = bwd_rh(s.W4);
  FVTA_CHECK_HIP(hipMemsetAsync(wk.slabs, 0, wk.slab_bytes, stream));
  if (tscale) FVTA_CHECK_HIP(hipMemsetAsync(wk.dscr, 0, (size_t)s.N * s.K * s.T * sizeof(float), stream));
  if (accumulate == 0 || accumulate == 3) {
    if (d->hinfo_stride)
      FVTA_CHECK_HIP(hipMemset2DAsync(d_hinfo, (size_t)d->hinfo_stride * sizeof(float), 0, (size_t)s.T * s.w * sizeof(float),
                                      (size_t)s.N, stream));
    else if (use_mask)  // every valid row is written by the main kernel (a fully masked stream: all of its rows)
      hipLaunchKernelGGL(attn_zero_masked_rows_kernel, dim3((unsigned)(((size_t)s.N * s.K * s.T + 3) / 4)), dim3(256), 0, stream, s,
                         hmask, d_hinfo);
    // (no masks: every row is valid and written)
  }
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(s.N), dim3(s.K > 4 ? 1024 : 256), 0, stream, s, sv, wk, d_h_a);
  AttnBwdArgs a;
  a.s = s;
  a.sv = sv;
  a.wk = wk;
  a.hinfo = hinfo;
  a.table = reinterpret_cast<const unsigned long long*>(table);
  a.d_h_a = d_h_a;
  a.d_hinfo = d_hinfo;
  a.accumulate = accumulate;
  a.tscale = tscale;
  a.hstride = d->hinfo_stride ? (size_t)d->hinfo_stride : (size_t)s.T * s.w;
  {
    static const int nt = [] {
      const char* e = getenv("FVTA_ATTN_BWD_NT");
      return e ? atoi(e) : FVTA_ATTN_BWD_NT_DEFAULT;
    }();
    a.nt = nt;
  }
  const dim3 grid(s.bsplit, s.N * s.ng);
  const bool prof_it = (size_t)s.N * s.K * s.T >= 65536;  // the context attention, see attn_fwd.hip
  if (prof_it) fvta_prof_begin(FVTA_PROF_ATTN_BWD_MAIN, stream);
  if (table) {
    if (s.w == 512) {
      if (accumulate == 1) hipLaunchKernelGGL((attn_bwd_main<128, 1, 32, false, true, true>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((attn_bwd_main<128, 1, 32, false, false, true>), grid, dim3(256), 0, stream, a);
    } else {
      if (accumulate == 1) hipLaunchKernelGGL((attn_bwd_main<256, 1, 32, false, true, true>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((attn_bwd_main<256, 1, 32, false, false, true>), grid, dim3(256), 0, stream, a);
    }
  } else
  switch (s.w) {
#define FVTA_BWD_LAUNCH2(TPR, G, TR, COS)                                                                             \
  do {                                                                                                              \
    if (accumulate == 1) hipLaunchKernelGGL((attn_bwd_main<TPR, G, TR, COS, true>), grid, dim3(256), 0, stream, a);  \
    else hipLaunchKernelGGL((attn_bwd_main<TPR, G, TR, COS, false>), grid, dim3(256), 0, stream, a);                 \
  } while (0)
#define FVTA_BWD_LAUNCH(TPR, G, TR)                                                                      \
  do {                                                                                                   \
    if (s.simi == 4) FVTA_BWD_LAUNCH2(TPR, G, TR, true);                                                 \
    else FVTA_BWD_LAUNCH2(TPR, G, TR, false);                                                            \
  } while (0)
    case 64: FVTA_BWD_LAUNCH(16, 1, 32); break;
    case 128: FVTA_BWD_LAUNCH(32, 1, 32); break;
    case 256: FVTA_BWD_LAUNCH(64, 1, 32); break;
    case 512: FVTA_BWD_LAUNCH(128, 1, 32); break;
    case 1024:  // (the cosine variant of the 32-row tile would spill: 16 rows)
      if (s.simi == 4) FVTA_BWD_LAUNCH2(256, 1, 16, true);
      else FVTA_BWD_LAUNCH2(256, 1, 32, false);
      break;
    case 2048: FVTA_BWD_LAUNCH(256, 2, 16); break;
#undef FVTA_BWD_LAUNCH
#undef FVTA_BWD_LAUNCH2
  }
  if (prof_it) fvta_prof_end(FVTA_PROF_ATTN_BWD_MAIN, 1, stream);
  FVTA_CHECK_LAUNCH("attn_bwd_main");
  if (tscale) {
    if (use_mask) hipLaunchKernelGGL(attn_bwd_pad_kernel, dim3(s.N * s.K), dim3(256), 0, stream, a, hmask);
    hipLaunchKernelGGL(attn_bwd_dscale_kernel, dim3((s.N * s.T + 255) / 256), dim3(256), 0, stream, s, wk.dscr, d_tscale);
    FVTA_CHECK_LAUNCH("attn_bwd_tw");
  }
  if (s.simi == 4) {
    hipLaunchKernelGGL(attn_bwd_cosine_q_kernel, dim3(s.N * s.JQ), dim3(256), 0, stream, s, wk, RH, hq, d_hq, accumulate);
  } else {
    hipLaunchKernelGGL(attn_bwd_reduce_q_kernel, dim3(s.N, (s.w + 255) / 256, attn_bwd_jz(s)), dim3(256), 0, stream, s, sv, wk, RH, hq,
                       d_hq, accumulate);
    hipLaunchKernelGGL(attn_bwd_params_kernel, dim3((s.w + 63) / 64 + 1), dim3(256), 0, stream, s, wk, dW, db);
  }
  FVTA_CHECK_LAUNCH("attn_bwd_reduce");
  return FVTA_OK;
}
