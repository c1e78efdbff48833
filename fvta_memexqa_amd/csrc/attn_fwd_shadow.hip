// Focal attention forward over the bi-LSTM's bf16 SHADOW rows (model_v2.py:210-298; JQ <= 32, w = 512 / 1024, simi 1-3).
//
// The context tensor hall[N,K,T,w] (model_v2.py:863-914) is, row by row, the concatenation of two half-rows the encoders
// have ALREADY written as bf16 -- the forward direction's h_t and the backward direction's, the MFMA operands of their own
// next steps (fvta_lstm_shadow_rows gives their addresses).  Storing the same values a second time as fp32 is the largest
// store of the (store-bound) forward step: 119 -> 100 us per launch without it.  This kernel is attn_fwd_pair16 reading
// those shadow rows instead of the fp32 tensor:
//   * a 16-row tile belongs to a PAIR of waves; wave hv of the pair owns direction hv's half-row (w/2 channels): ONE address
//     per row and wave (`rowptr`, compacted like the row list), 16 bytes = 8 channels per lane and load, 64 contiguous bytes
//     of a row per instruction -- half the load instructions and half the registers of the fp32 tile;
//   * a bf16 value converts EXACTLY to fp16: the 3-term split collapses to hi x (Qhi + Qlo) -- two MFMAs per k-step and
//     column tile instead of three, no split arithmetic;
//   * everything else -- the pair's LDS flag hand-shake, max / first arg-max / tanh / online softmax by DPP row operations,
//     the weighted sum from the same registers with the next tile's rows taking their place, flat dealing of an album's
//     tiles to the pairs, (m, l, u) partials for attn_merge_kernel -- is attn_fwd_pair16's (attn_fwd.hip).
#include "attn_fwd_shared.h"

namespace fvta {

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
// (a row address comes out of a table as an integer: cast to a GLOBAL pointer -- through a generic one the loads are FLAT
//  instructions, which count on lgkmcnt as well, so every LDS wait of the pair hand-shake waited for the next tile's rows)
typedef const u32x4s __attribute__((address_space(1)))* grow16_ptr;

// rowptr[half][nk * T + pos] = table[half][nk * T + idx[nk * T + pos]]: the shadow addresses in the order of the compacted row
// lists (attn_compact_kernel has run).  grid (N K), 256 threads
__global__ __launch_bounds__(256) void attn_shadow_compact_kernel(AttnShape s, AttnSaved sv, const unsigned long long* __restrict__ table,
                                                                  unsigned long long* __restrict__ rowptr) {
  const int nk = blockIdx.x, T = s.T;
  const size_t nkt = (size_t)s.N * s.K * T;
  const int cnt = sv.cnt[nk];
  for (int p = threadIdx.x; p < cnt; p += 256) {
    const int t = sv.idx[(size_t)nk * T + p];
    rowptr[(size_t)nk * T + p] = table[(size_t)nk * T + t];
    rowptr[nkt + (size_t)nk * T + p] = table[nkt + (size_t)nk * T + t];
  }
}

template <int NBH, int RMODE>
__global__ __launch_bounds__(512, 1) void attn_fwd_pair16h(AttnFwdArgs a, int G_all, const unsigned long long* __restrict__ rowptr) {
  constexpr int NKS = NBH / 2;   // MFMA steps (32 channels = one 16-byte load per lane) over a wave's half of the channels
  constexpr int NU = NBH / 16;   // accumulator registers (float4) per lane
  static_assert(NBH % 16 == 0, "a wave's channel blocks are dealt round-robin to the 16 row lanes");
  extern __shared__ __attribute__((aligned(16))) char s_dyn[];
  half8(*s_qhi)[2][64] = reinterpret_cast<half8(*)[2][64]>(s_dyn);                                        // [2 NKS][2][64]
  half8(*s_qlo)[2][64] = reinterpret_cast<half8(*)[2][64]>(s_dyn + (size_t)2 * NKS * 2 * 64 * sizeof(half8));
  float* s_vec = reinterpret_cast<float*>(s_dyn + (size_t)2 * 2 * NKS * 2 * 64 * sizeof(half8));           // [2][w]
  __shared__ float s_ct[32];
  __shared__ __attribute__((aligned(16))) float s_x[8][8 * 64];  // per wave: its partial scores [jt * 4 + i][lane]
  __shared__ float s_rt[8][16];                                  // per wave: its partial row terms
  __shared__ int s_kstart[65], s_kcnt[64], s_kall[64], s_flat;
  __shared__ int s_pub[8], s_done[8];  // last round whose partials a wave has published / whose partner data it has consumed

  const AttnShape& s = a.s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pair = wave >> 1, hv = wave & 1;
  const int l15 = lane & 15, kq = lane >> 4;
  const int T = s.T, w = s.w, JP = s.JP;
  const size_t nkt = (size_t)s.N * s.K * T;
  const int nwg = s.N * G_all, per = (nwg + 7) / 8;
  const int wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (wg >= nwg || (int)(blockIdx.x >> 3) >= per) return;
  int n = wg / G_all, g0 = wg % G_all, G = G_all;
  if (a.wgtab) {
    const uint32_t e = a.wgtab[wg];
    if (e == 0xffffffffu) return;
    n = (int)(e & 0xffffu);
    g0 = (int)((e >> 16) & 0xffu);
    G = (int)(e >> 24);
  }
  {
    // the question operand, B layout of v_mfma_f32_16x16x32_f16 in the NATURAL channel order of the 16-byte row loads:
    // k-step ks of half pc, lane (j = l15 + 16 jt, q4): the 8 channels 32 ks + 8 q4 .. + 7 of the half
    const int W4c = w / 4;
    const uint16_t* qh = a.sv.Qh + (size_t)n * 2 * W4c * 32 * 4;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    constexpr int QIT = 2 * 2 * NKS * 2 * 64 / 512;
    u32x2 x0[QIT], x1[QIT];
#pragma unroll
    for (int it = 0; it < QIT; ++it) {
      const int e = tid + 512 * it;
      const int ln = e & 63, jt = (e >> 6) & 1, ks = (e >> 7) % (2 * NKS), pc = (e >> 7) / (2 * NKS);
      const int j = (ln & 15) + 16 * jt, q4 = ln >> 4;
      x0[it] = *reinterpret_cast<const u32x2*>(qh + (((size_t)pc * W4c + 8 * ks + 2 * q4) * 32 + j) * 4);
      x1[it] = *reinterpret_cast<const u32x2*>(qh + (((size_t)pc * W4c + 8 * ks + 2 * q4 + 1) * 32 + j) * 4);
    }
    float v0[2], v1[2];
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
      const u32x4s xx = __builtin_shufflevector(x0[it], x1[it], 0, 1, 2, 3);
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
  const int P = 4 * G, pg = 4 * g0 + pair;
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
  const int lo = flat ? tot * pg / P : 0, hi = flat ? tot * (pg + 1) / P : 0;
  auto empty_partial = [&](int nk, int split) {
    float* pp = a.part + ((size_t)nk * s.nsplit + split) * (w + 4);
    pp[0] = -INFINITY;
    pp[1] = 0.f;
    pp[2] = -INFINITY;
  };
  if (flat && g0 == 0) {
    for (int e = tid; e < nitems_n; e += 512) {
      const int k = e / s.nsplit, sp = e % s.nsplit;
      const int st = s_kstart[k], en = s_kstart[k + 1];
      bool filled = false;
      if (en > st) {
        const int px = ((st + 1) * P - 1) / tot + sp;
        filled = px < P && max(tot * px / P, st) < min(tot * (px + 1) / P, en);
      }
      if (!filled) empty_partial(n * s.K + k, sp);
    }
  }
  struct Seg {
    int nk, t0, t1, slot, cnt, allm;
  };
  auto item_seg = [&](int il, Seg& sg) {
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
  auto next_seg = [&](Seg& sg) {
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
      if (hv == 0 && lane == 0) empty_partial(sg.nk, sg.slot);
    }
  };
  int rounds = hi - lo;
  if (!flat) {
    rounds = 0;
    Seg sg;
    for (int il = g0 + G * pair; il < nitems_n; il += 4 * G)
      if (item_seg(il, sg)) rounds += sg.t1 - sg.t0;
  }
  const int pwv = wave ^ 1;
  typedef __attribute__((address_space(3))) int lds_int;
  auto wait_flag = [&](int* flag, int want) {
    volatile lds_int* f = (volatile lds_int*)flag;
    bool arrived = false;
    for (int spin = 0; spin < (1 << 28); ++spin) {
      if (*f >= want) {
        arrived = true;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!arrived) {  // (a wedged partner wave: see attn_fwd_pair16)
      if (a.fault) {
        __hip_atomic_store(a.fault, (int)ATTN_FAULT_PAIR_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
      }
      __builtin_trap();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  auto post_flag = [&](int* flag, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *(volatile lds_int*)flag = v;
  };

  Seg cs = {0, 0, 0, 0, 0, 0}, ns = {0, 0, 0, 0, 0, 0};
  int ctl = 0, ntl = 0;
  int t_cur = 0, t_nxt = 0;
  bool v_cur = false, v_nxt = false;
  unsigned long long p_nxt = 0;  // the next tile's shadow half-row of this lane's row (this wave's direction)
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 u[NU];
  u32x4s hb[NKS];  // the tile: 8 bf16 channels per register quad -- chunk i = channels 32 i + 8 kq .. + 7 of the half-row
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long* myptr = rowptr + (size_t)hv * nkt;
  // (an invalid row of a stream's last tile reads the stream's last valid row: finite data, weight 0)
  auto rows_of = [&](const Seg& sg, int tl, int& t, bool& v, unsigned long long& p) {
    const int lr = tl * 16 + l15;
    v = lr < sg.cnt;
    const size_t at = (size_t)sg.nk * T + max(min(lr, sg.cnt - 1), 0);
    t = a.sv.idx[at];
    p = myptr[at];
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
    unsigned long long p_cur;
    rows_of(cs, ctl, t_cur, v_cur, p_cur);
    const grow16_ptr rowp = (grow16_ptr)(p_cur + 16 * kq);
#pragma unroll
    for (int i = 0; i < NKS; ++i) hb[i] = rowp[4 * i];
    has_n = tile_after(cs, ctl, ns, ntl);
    if (!has_n) {
      ns = cs;
      ntl = ctl;
    }
    rows_of(ns, ntl, t_nxt, v_nxt, p_nxt);
  }
  const int cbase = 16 * NBH * hv;  // first channel of this wave's half
  // a chunk's 8 channels as fp32 (a bf16 is the upper half of an fp32)
  auto unpack = [](const u32x4s q, f32x4& lo4, f32x4& hi4) {
    lo4 = f32x4{__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xffff0000u)};
    hi4 = f32x4{__uint_as_float(q[2] << 16), __uint_as_float(q[2] & 0xffff0000u), __uint_as_float(q[3] << 16), __uint_as_float(q[3] & 0xffff0000u)};
  };

#pragma unroll 2
  for (int g = 0; g < rounds; ++g) {
    const bool rvalid = v_cur;
    const int t = t_cur;
    const bool allm = cs.allm != 0;
    float xown[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xown[i] = 0.f;
    if (active && !allm) {
      f32x4 ahh[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 axx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 rt4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        f32x4 f0, f1;
        unpack(hb[ks], f0, f1);
        const int c0 = cbase + 32 * ks + 8 * kq;
        if (RMODE == 1)
          rt4 += f0 * *reinterpret_cast<const f32x4*>(&s_vec[c0]) + f1 * *reinterpret_cast<const f32x4*>(&s_vec[c0 + 4]);
        else if (RMODE == 2)
          rt4 += (f0 * f0) * *reinterpret_cast<const f32x4*>(&s_vec[w + c0]) + (f1 * f1) * *reinterpret_cast<const f32x4*>(&s_vec[w + c0 + 4]);
        else
          rt4 += f0 * (*reinterpret_cast<const f32x4*>(&s_vec[c0]) + *reinterpret_cast<const f32x4*>(&s_vec[w + c0]) * f0) +
                 f1 * (*reinterpret_cast<const f32x4*>(&s_vec[c0 + 4]) + *reinterpret_cast<const f32x4*>(&s_vec[w + c0 + 4]) * f1);
        // bf16 -> fp16 is exact (8 significant bits, |h| < 1): the row side of the split has no low term
        const half8 hi8 = cat_h2(__builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(f0[0], f0[1])),
                                 __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(f0[2], f0[3])),
                                 __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(f1[0], f1[1])),
                                 __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(f1[2], f1[3])));
        const int kg = NKS * hv + ks;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          ahh[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi8, s_qhi[kg][jt][lane], ahh[jt], 0, 0, 0);
          axx[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi8, s_qlo[kg][jt][lane], axx[jt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      float rtp = (rt4[0] + rt4[1]) + (rt4[2] + rt4[3]);
      rtp += __shfl_xor(rtp, 16, 64);
      rtp += __shfl_xor(rtp, 32, 64);
      wait_flag(&s_done[pwv], g);  // the partner has read my partials of round g - 1 (rounds are numbered from 1)
      if (kq == 0) s_rt[wave][l15] = rtp;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xown[jt * 4 + i] = ahh[jt][i] + axx[jt][i] * (1.f / 2048.f);
          s_x[wave][(jt * 4 + i) * 64 + lane] = xown[jt * 4 + i];
        }
    }
    asm volatile("" ::"v"(t_nxt), "v"(p_nxt));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my partials are in LDS
    if (lane == 0) post_flag(&s_pub[wave], g + 1);
    wait_flag(&s_pub[pwv], g + 1);
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the partner's partials are in my registers
      if (lane == 0) post_flag(&s_done[wave], g + 1);
      const float m_new = fmaxf(m_run, row16_max(am));
      const float scale = expf(m_run - m_new);
      const float pr = expf(am - m_new);
      l_run = l_run * scale + row16_sum(pr);
      m_run = m_new;
      if (scale != 1.f) {
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] *= scale;
      }
      // the weighted sum: chunk i holds blocks 2 i (channels 32 i + 8 kq + 0..3) and 2 i + 1 (+ 4..7); lane b & 15 keeps block b
      const grow16_ptr rowp_next = (grow16_ptr)(p_nxt + 16 * kq);
#pragma unroll
      for (int i = 0; i < NKS; ++i) {
        f32x4 va, vb;
        unpack(hb[i], va, vb);
        va *= pr;
        vb *= pr;
        hb[i] = rowp_next[4 * i];  // (no next tile: p_nxt re-reads the current tile's rows)
        row16_sum4(va);
        row16_sum4(vb);
        if (((2 * i) & 15) == l15) u[(2 * i) >> 4] += va;
        if (((2 * i + 1) & 15) == l15) u[(2 * i + 1) >> 4] += vb;
      }
      if (ctl + 1 == cs.t1) {
        // the segment's partial (m, l, u): lane (l15, kq), u[i] = block b = 16 i + l15: channels 32 (b >> 1) + 8 kq + 4 (b & 1) ..
        float* part = a.part + ((size_t)cs.nk * s.nsplit + cs.slot) * (w + 4);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          const int b = 16 * i + l15;
          *reinterpret_cast<f32x4*>(part + 4 + cbase + 32 * (b >> 1) + 8 * kq + 4 * (b & 1)) = u[i];
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
    // (top level of the loop body: the loads' destinations ARE the loop-carried registers)
    rows_of(ns, ntl, t_nxt, v_nxt, p_nxt);
  }
}

// host side.  rowptr: [2][N K T] device words (the caller's workspace); table: fvta_lstm_shadow_rows' [2][N K T]
bool shadow_covers(const AttnShape& s) { return s.JT == 1 && (s.w == 512 || s.w == 1024) && s.simi != 4; }

void launch_attn_shadow_compact(const AttnShape& s, const AttnSaved& sv, const uint64_t* table, uint64_t* rowptr, hipStream_t stream) {
  hipLaunchKernelGGL(attn_shadow_compact_kernel, dim3(s.N * s.K), dim3(256), 0, stream, s, sv,
                     reinterpret_cast<const unsigned long long*>(table), reinterpret_cast<unsigned long long*>(rowptr));
}

bool launch_attn_fwd_pair16h(const AttnFwdArgs& a, int G, const uint64_t* rowptr, hipStream_t stream) {
  const AttnShape& s = a.s;
  if (!shadow_covers(s)) return false;
  const int nwg = s.N * G;
  const dim3 grid(((nwg + 7) / 8) * 8);
  const int rmode = s.simi == 1 ? 1 : (s.simi == 3 ? 3 : 2);
  const unsigned long long* rp = reinterpret_cast<const unsigned long long*>(rowptr);
#define FVTA_PH(NBH, RM)                                                                                                 \
  do {                                                                                                                   \
    const size_t lds = (size_t)2 * 2 * (NBH / 2) * 2 * 64 * 16 + (size_t)2 * s.w * sizeof(float);                        \
    (void)hipFuncSetAttribute((const void*)attn_fwd_pair16h<NBH, RM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((attn_fwd_pair16h<NBH, RM>), grid, dim3(512), lds, stream, a, G, rp);                             \
  } while (0)
#define FVTA_PHW(NBH)                                                                                                    \
  do {                                                                                                                   \
    if (rmode == 1) FVTA_PH(NBH, 1); else if (rmode == 2) FVTA_PH(NBH, 2); else FVTA_PH(NBH, 3);                         \
  } while (0)
  if (s.w == 1024)
    FVTA_PHW(32);
  else
    FVTA_PHW(16);
#undef FVTA_PHW
#undef FVTA_PH
  return true;
}

}  // namespace fvta
