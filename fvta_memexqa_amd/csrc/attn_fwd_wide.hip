// Focal attention forward for WIDE rows (model_v2.py:210-298 at w = 2048, JQ <= 64: BASELINE.json configs[4], the long
// album -- 120 photos x 60 tokens, hidden 1024).
//
// At w = 1024 / JQ <= 32 the question operand (fp16 hi + lo pieces of Qs = U o q) fits the LDS once per workgroup
// (attn_fwd_pair16).  Here it is 512 KB per album: it fits neither LDS nor registers, and the general kernel
// (attn_fwd_main) streams it once per 32-row tile next to rows it reads TWICE, on the exact-fp32 matrix pipe: 0.27 of HBM.
// This kernel keeps the ROWS stationary and streams the QUESTION:
//   * a workgroup of eight waves holds a tile of 32 rows in registers -- wave v the channels [256 v, 256 v + 256) of both
//     16-row halves (128 registers, lane = row in the MFMA A order, 64 contiguous bytes of a row per load instruction as in
//     the pair kernel) -- scores them on the fp16 3-term split (v_mfma_f32_16x16x32_f16) and folds them into the online
//     softmax / weighted sum FROM THE SAME REGISTERS: rows are read once;
//   * the question operand of the wave's channels comes as ready-made B fragments (attn_prep_q_kernel writes them in
//     fragment order: 1 KB per fragment, lane order inside) through a WAVE-PRIVATE LDS ring of three 4-KB stages filled by
//     LDS-DMA from L2, two stages ahead, behind counted waits -- 512 KB per 32-row tile and workgroup, the same stream for
//     every tile of the album, so it simply keeps running across tiles;
//   * a tile is scored in two passes over the question positions (j < 32, j >= 32: the 64 accumulator registers of one pass
//     are what fits beside the rows), the eight waves' partial scores of a 16-row half meet in LDS, wave v finishes rows
//     2 v, 2 v + 1 (sum in wave order, mask, max / FIRST arg-max over j, tanh), every wave folds the 16 maxima into the
//     same online softmax and adds p_t h_t of its channels (row sums by fused DPP adds), handing each block's registers to
//     the next tile's rows as it goes.
// Work: a workgroup serves one album and a run of its 32-row tiles, cut out of the album's flat (k, tile) order like the
// pair kernel's; a run's pieces of a stream end in (m, l, u) partials for attn_merge_kernel.
#include "attn_fwd_shared.h"

namespace fvta {

typedef __attribute__((address_space(3))) void* wide_lds_ptr;
constexpr int WIDE_MAXR = 256;  // tiles of a workgroup's run (the list sits in LDS)
constexpr int WIDE_NSTG = 3;    // ring stages (4 KB: the four B fragments of one k-step x 32 question positions)

template <int N>
__device__ __forceinline__ void wide_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// fragment p of a landed stage (read as the DMA wrote it: fragment order = lane order, no bank conflicts).
// (__restrict__: without the no-alias scope the compiler waits vmcnt(0) in front of every LDS read that follows an LDS-DMA;
//  the hand-over back to the DMA is ordered by an asm with a memory clobber)
__device__ __forceinline__ half8 wide_read_frag(const char* __restrict__ slot, int lane, int p) {
  return *reinterpret_cast<const half8*>(slot + 1024 * p + 16 * lane);
}

// NKS = w / 256: k-steps (32 channels) of a wave; NH = ceil(JQ / 32): passes over the question positions
template <int NKS, int NH, int RMODE>
__global__ __launch_bounds__(512, 1) void attn_fwd_wide(AttnFwdArgs a, int G_all) {
  constexpr int CW = 32 * NKS;         // channels of a wave
  constexpr int NB = 2 * NKS;          // 16-channel blocks of a wave
  constexpr int NST = NKS * NH;        // ring stages per tile
  constexpr int XLD = 36;              // floats per exchanged score row (32 positions of a pass + pad)
  static_assert(NB == 16, "a wave's 16 blocks are dealt to the 16 row lanes (w = 2048)");
  // -DFVTA_WIDE_ABL=bits (timing builds, results are wrong): 1 no question stream, 2 no score pass, 4 no exchange /
  // finishing, 8 no weighted sum (refill only), 16 no barriers
#ifdef FVTA_WIDE_ABL
  constexpr int abl = FVTA_WIDE_ABL;
#else
  constexpr int abl = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char s_dyn[];
  char* const s_ring = s_dyn;                                                             // [8 waves][WIDE_NSTG][4 KB]
  float* const s_x = reinterpret_cast<float*>(s_dyn + (size_t)8 * WIDE_NSTG * 4096);      // [8 waves][2 halves x 16 rows][XLD]
  float* const s_vec = s_x + 8 * 32 * XLD;                                                // [2][w]
  __shared__ float s_rt[2][8][16], s_am[2][16], s_ct[64];
  __shared__ int s_kstart[65], s_kcnt[64], s_kall[64], s_flat;
  __shared__ __attribute__((aligned(16))) int s_tiles[WIDE_MAXR][4];  // (nk, tile, rows of the stream, flags)

  const AttnShape& s = a.s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, kq = lane >> 4;
  const int T = s.T, w = s.w, JP = s.JP;
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
  for (int c = tid; c < w; c += 512) {
    s_vec[c] = a.sv.vecs[VEC_RH * w + c];
    s_vec[w + c] = a.sv.vecs[VEC_R2 * w + c];
  }
  if (tid < 64) s_ct[tid] = tid < JP ? a.sv.ct[(size_t)n * JP + tid] : 0.f;
  const uint64_t qvalid = a.sv.qvalid[(size_t)n * 2];
  asm volatile("" ::"s"(qvalid));
  const int nitems_n = s.K * s.nsplit;
  const int P = G, pg = g0;
  if (tid == 64) {
    int acc = 0;
    for (int k = 0; k < s.K; ++k) {
      const int c = a.sv.cnt[n * s.K + k];
      s_kcnt[k] = c;
      s_kall[k] = a.sv.allmasked[n * s.K + k];
      s_kstart[k] = acc;
      acc += (c + 31) >> 5;
    }
    s_kstart[s.K] = acc;
    int ok = acc > 0;  // FLAT dealing if no stream is cut into more pieces than it has partial slots
    for (int k = 0; k < s.K && ok; ++k) {
      const int st = s_kstart[k], en = s_kstart[k + 1];
      if (en > st && ((en * P - 1) / acc) - (((st + 1) * P - 1) / acc) + 1 > s.nsplit) ok = 0;
    }
    s_flat = ok;
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
  if (flat && g0 == 0) {  // the partial slots no run fills
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
    const int tiles_total = (c + 31) >> 5;
    const int tiles_per = (tiles_total + s.nsplit - 1) / s.nsplit;
    sg.nk = n * s.K + k;
    sg.t0 = split * tiles_per;
    sg.t1 = min(tiles_total, sg.t0 + tiles_per);
    sg.slot = split;
    sg.cnt = c;
    sg.allm = s_kall[k];
    return sg.t1 > sg.t0;
  };
  int it_k = 0, it_il = g0 - G;
  auto next_seg = [&](Seg& sg) {  // (thread 0 only)
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
      it_il += G;
      if (it_il >= nitems_n) return false;
      if (item_seg(it_il, sg)) return true;
      empty_partial(sg.nk, sg.slot);  // empty split
    }
  };
  int rounds = hi - lo;
  if (!flat) {
    rounds = 0;
    Seg sg;
    for (int il = g0; il < nitems_n; il += G)
      if (item_seg(il, sg)) rounds += sg.t1 - sg.t0;
  }
  if (rounds > WIDE_MAXR) __builtin_trap();  // (ruled out by the host's bound, wide_covers, and by attn_balance_kernel's run_cap)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  if (tid == 0) {  // the run's tiles, listed once
    Seg sg = {0, 0, 0, 0, 0, 0};
    int i = 0;
    while (i < rounds && next_seg(sg))
      for (int tl = sg.t0; tl < sg.t1 && i < rounds; ++tl, ++i)
        *reinterpret_cast<i32x4*>(s_tiles[i]) = i32x4{sg.nk, tl, sg.cnt, sg.slot | (sg.allm ? 256 : 0) | (tl + 1 == sg.t1 ? 512 : 0)};
    if (!flat) {
      Seg rest;
      while (next_seg(rest)) {
      }
    }
  }
  __syncthreads();
  if (rounds == 0) return;
  struct Tile {
    int nk, tl, cnt, flags;
  };
  auto get_tile = [&](int i) {
    const i32x4 d = *reinterpret_cast<const i32x4*>(s_tiles[min(i, rounds - 1)]);
    Tile t;
    t.nk = __builtin_amdgcn_readfirstlane(d[0]);
    t.tl = __builtin_amdgcn_readfirstlane(d[1]);
    t.cnt = __builtin_amdgcn_readfirstlane(d[2]);
    t.flags = __builtin_amdgcn_readfirstlane(d[3]);
    return t;
  };

  // ---- the question stream of this wave: NST stages of 4 KB per tile, the same for every tile of the album
  const uint16_t* qf = a.sv.Qh + (size_t)n * 2 * w * JP + (size_t)wave * NST * 2048;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qf), 0, (unsigned)(NST * 4096), 0x00020000);
  char* const my_ring = s_ring + (size_t)wave * WIDE_NSTG * 4096;
  int q_stage = 0, q_slot = 0;  // the next stage to request (position in the tile's stream) and the ring slot it goes to
  auto issue_stage = [&]() {
    if (abl & 1) return;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (wide_lds_ptr)(my_ring + q_slot * 4096 + 1024 * p), 16, (unsigned)(16 * lane),
                                               (unsigned)(q_stage * 4096 + 1024 * p), 0, 0);
    q_stage = q_stage + 1 == NST ? 0 : q_stage + 1;
    q_slot = q_slot + 1 == WIDE_NSTG ? 0 : q_slot + 1;
  };
  int c_slot = 0;  // the ring slot of the stage consumed next
  issue_stage();
  issue_stage();
  issue_stage();

  // ---- the rows: h[sb][b] = the 4 channels CW wave + 16 b + 4 kq .. of row (16 sb + l15) of the tile
  f32x4 h[2][NB];
  f32x4 u = {0.f, 0.f, 0.f, 0.f};  // lane (l15, kq) keeps block b = l15: channels CW wave + 16 l15 + 4 kq ..
  float m_run = -INFINITY, l_run = 0.f;
  const int coff = CW * wave + 4 * kq;
  Tile cur = get_tile(0), nxt = get_tile(1);
  int t_cur[2], t_nxt[2];
  auto row_ids = [&](const Tile& tl, int (&t)[2]) {  // (unconditional, clamped: rows past the stream's count re-read its last row)
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) t[sb] = a.sv.idx[(size_t)tl.nk * T + max(min(tl.tl * 32 + 16 * sb + l15, tl.cnt - 1), 0)];
  };
  row_ids(cur, t_cur);
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {
    const float* rowp = a.hinfo + (size_t)cur.nk * a.hstride + (size_t)t_cur[sb] * w + coff;
#pragma unroll
    for (int b = 0; b < NB; ++b) h[sb][b] = *reinterpret_cast<const f32x4*>(rowp + 16 * b);
  }
  row_ids(nxt, t_nxt);
  float* const my_x = s_x + (size_t)wave * 32 * XLD;

  for (int g = 0; g < rounds; ++g) {
    const bool allm = (cur.flags & 256) != 0;
    // the finishing lanes' running maximum over the passes: lane (j = lane & 31, rr = lane >> 5) serves row 2 wave + rr
    float bst[2] = {-INFINITY, -INFINITY};
    int bj[2] = {64, 64};
#pragma unroll
    for (int hj = 0; hj < NH; ++hj) {
      f32x4 ahh[2][2], axx[2][2];
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int aj = 0; aj < 2; ++aj) ahh[sb][aj] = axx[sb][aj] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 rt4[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        // the stage of (hj, ks) was requested three stages ago: the two younger ones (8 instructions) may still be in flight
        if (!(abl & 1)) wide_wait_vmcnt<8>();
        const char* slot = my_ring + c_slot * 4096;
        c_slot = c_slot + 1 == WIDE_NSTG ? 0 : c_slot + 1;
        if (!allm && !(abl & 2)) {
#pragma unroll
          for (int sb = 0; sb < 2; ++sb) {
            const f32x4 h0 = h[sb][2 * ks], h1 = h[sb][2 * ks + 1];
            if (hj == 0) {
              const float* vp = s_vec + coff + 32 * ks;
              if (RMODE == 1)
                rt4[sb] += h0 * *reinterpret_cast<const f32x4*>(vp) + h1 * *reinterpret_cast<const f32x4*>(vp + 16);
              else if (RMODE == 2)
                rt4[sb] += (h0 * h0) * *reinterpret_cast<const f32x4*>(vp + w) + (h1 * h1) * *reinterpret_cast<const f32x4*>(vp + w + 16);
              else
                rt4[sb] += h0 * (*reinterpret_cast<const f32x4*>(vp) + *reinterpret_cast<const f32x4*>(vp + w) * h0) +
                           h1 * (*reinterpret_cast<const f32x4*>(vp + 16) + *reinterpret_cast<const f32x4*>(vp + w + 16) * h1);
            }
            // (the split is redone in every pass: left to itself the compiler keeps the first pass's 16 x 2 fragments -- 128
            //  registers -- alive for the second and spills rows; the empty asm makes this pass's inputs new values)
            f32x4 s0 = h0, s1 = h1;
            if (NH > 1) asm volatile("" : "+v"(s0), "+v"(s1));
            half8 hi8, lo8;
            split_f16x8(s0, s1, hi8, lo8);
#pragma unroll
            for (int aj = 0; aj < 2; ++aj) {  // (the fragments are re-read per 16-row half: 8 registers instead of 16 live)
              const half8 Fh = wide_read_frag(slot, lane, 2 * aj), Fl = wide_read_frag(slot, lane, 2 * aj + 1);
              ahh[sb][aj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi8, Fh, ahh[sb][aj], 0, 0, 0);
              axx[sb][aj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi8, Fl, axx[sb][aj], 0, 0, 0);
              axx[sb][aj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo8, Fh, axx[sb][aj], 0, 0, 0);
            }
          }
        }
        // every read of the slot has returned (its values went into the MFMAs above): the slot takes the stage three ahead
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_stage();
        __builtin_amdgcn_sched_barrier(0);  // (nothing of the next k-step is hoisted up here: its operands would be live for nothing)
      }
      // ---- this pass's partial scores of both 16-row halves -> LDS (the exchange rows were read in the pass before: barrier)
      __builtin_amdgcn_sched_barrier(0);
      if (!(abl & 16)) lds_barrier();
      if (!allm) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          if (hj == 0) {
            float r = (rt4[sb][0] + rt4[sb][1]) + (rt4[sb][2] + rt4[sb][3]);
            r += __shfl_xor(r, 16, 64);
            r += __shfl_xor(r, 32, 64);
            if (kq == 0) s_rt[sb][wave][l15] = r;
          }
#pragma unroll
          for (int aj = 0; aj < 2; ++aj) {
            const f32x4 x4 = ahh[sb][aj] + axx[sb][aj] * (1.f / 2048.f);  // lane (j = l15 + 16 aj, kq): rows 4 kq + i
#pragma unroll
            for (int i = 0; i < 4; ++i) my_x[(sb * 16 + 4 * kq + i) * XLD + 16 * aj + l15] = x4[i];
          }
        }
      }
      if (!(abl & 16)) lds_barrier();  // every wave's partials of this pass are published
      if (!allm && !(abl & 4)) {
        const int j = 32 * hj + (lane & 31), rr = lane >> 5;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          const int row = 2 * wave + rr;
          float x = 0.f, rt = 0.f;
#pragma unroll
          for (int v = 0; v < 8; ++v) {
            x += s_x[((size_t)v * 32 + sb * 16 + row) * XLD + (lane & 31)];
            rt += s_rt[sb][v][row];
          }
          x += rt + s_ct[j];
          const bool jok = (qvalid >> j) & 1ull;
          float best = jok ? x : -INFINITY;
          int bestj = jok ? j : 64;
          row16_argmax(best, bestj);
          {
            const float ob = __shfl_xor(best, 16, 64);
            const int oj = __shfl_xor(bestj, 16, 64);
            if (ob > best || (ob == best && oj < bestj)) {
              best = ob;
              bestj = oj;
            }
          }
          if (best > bst[sb] || (best == bst[sb] && bestj < bj[sb])) {  // (positions grow with the pass: ties keep the earlier one)
            bst[sb] = best;
            bj[sb] = bestj;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // (the next pass's first k-step stays behind the finishing: its operands would only add pressure)
    }
    // ---- the rows' maxima: tanh, saved state, and the two halves' values for every wave
    {
      const int rr = lane >> 5, row = 2 * wave + rr;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const int lr = cur.tl * 32 + 16 * sb + row;
        const bool rvalid = lr < cur.cnt;
        float am = rvalid ? FVTA_NEG : -INFINITY;
        if (!allm) {
          const int bestj = bj[sb] >= 64 ? 0 : bj[sb];
          am = rvalid ? (s.add_tanh ? fvta_tanh(bst[sb]) : bst[sb]) : -INFINITY;
          const int t = __shfl(t_cur[sb], row, 64);  // (the row's number sits in lane `row` of the half's id register)
          if ((lane & 31) == 0 && rvalid) {
            a.sv.amax[(size_t)cur.nk * T + t] = am;
            a.sv.jmax[(size_t)cur.nk * T + t] = (uint8_t)bestj;
          }
        }
        if ((lane & 31) == 0) s_am[sb][row] = am;
      }
    }
    if (!(abl & 16)) lds_barrier();  // both halves' row maxima are published
    // ---- every wave folds them into the same online softmax and adds its channels' p h, refilling the registers as it goes
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const float am = s_am[sb][l15];
      const float m_new = fmaxf(m_run, row16_max(am));
      const float scale = __expf(m_run - m_new);
      const float pr = __expf(am - m_new);
      l_run = l_run * scale + row16_sum(pr);
      m_run = m_new;
      u *= scale;
      const float* rowp_next = a.hinfo + (size_t)nxt.nk * a.hstride + (size_t)t_nxt[sb] * w + coff;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x4 v = h[sb][b] * pr;
        h[sb][b] = *reinterpret_cast<const f32x4*>(rowp_next + 16 * b);
        if (!(abl & 8)) row16_sum4(v);
        if (b == l15) u += v;
      }
    }
    if (cur.flags & 512) {  // the segment's partial (m, l, u)
      float* part = a.part + ((size_t)cur.nk * s.nsplit + (cur.flags & 255)) * (w + 4);
      *reinterpret_cast<f32x4*>(part + 4 + CW * wave + 16 * l15 + 4 * kq) = u;
      u = f32x4{0.f, 0.f, 0.f, 0.f};
      if (tid == 0) {
        part[0] = m_run;
        part[1] = l_run;
        part[2] = m_run;
      }
      m_run = -INFINITY;
      l_run = 0.f;
    }
    cur = nxt;
    t_cur[0] = t_nxt[0];
    t_cur[1] = t_nxt[1];
    nxt = get_tile(g + 2);
    row_ids(nxt, t_nxt);  // (top level of the loop body: the loads' destinations ARE the loop-carried registers)
  }
  wide_wait_vmcnt<0>();  // (the stream ran two stages past the last tile)
}

// host side: shapes covered, LDS, instantiation
int wide_max_run() { return WIDE_MAXR; }
bool wide_covers(const AttnShape& s, int G) {
  if (!(s.w == 2048 && s.JT <= 2 && s.simi != 4)) return false;
  const int t32 = (s.T + 31) / 32;
  const long flat_max = ((long)s.K * t32 + G - 1) / G + 1;
  const long rr_max = (long)((s.K * s.nsplit + G - 1) / G) * ((t32 + s.nsplit - 1) / s.nsplit);
  return (flat_max > rr_max ? flat_max : rr_max) <= WIDE_MAXR;
}

bool launch_attn_fwd_wide(const AttnFwdArgs& a, int G, hipStream_t stream) {
  const AttnShape& s = a.s;
  if (!wide_covers(s, G)) return false;
  const int nwg = s.N * G;
  const dim3 grid(((nwg + 7) / 8) * 8);
  const int rmode = s.simi == 1 ? 1 : (s.simi == 3 ? 3 : 2);
#define FVTA_WD(NH, RM)                                                                                                  \
  do {                                                                                                                   \
    const size_t lds = (size_t)8 * WIDE_NSTG * 4096 + (size_t)8 * 32 * 36 * sizeof(float) + (size_t)2 * s.w * 4; \
    (void)hipFuncSetAttribute((const void*)attn_fwd_wide<8, NH, RM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((attn_fwd_wide<8, NH, RM>), grid, dim3(512), lds, stream, a, G);                                  \
  } while (0)
#define FVTA_WDH(NH)                                                                                                     \
  do {                                                                                                                   \
    if (rmode == 1) FVTA_WD(NH, 1); else if (rmode == 2) FVTA_WD(NH, 2); else FVTA_WD(NH, 3);                            \
  } while (0)
  if (s.JT == 2)
    FVTA_WDH(2);
  else
    FVTA_WDH(1);
#undef FVTA_WDH
#undef FVTA_WD
  return true;
}

}  // namespace fvta
