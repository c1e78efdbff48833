// Shared host/device structures of the bi-LSTM kernels (fp32 and bf16 engines).
#pragma once
#include "fvta_common.h"

namespace fvta {

typedef unsigned short bf16_t;

// ------------------------------------------------------------------ plan ----
struct PlanHeader {
  int64_t out_ld;
  int32_t B, J, in, d;
  int64_t x_bw_delta;  // elements from the forward direction's x (and dx) to the backward direction's: 0 = one shared input
  uint32_t magic;      // PLAN_MAGIC once plan_sort has run on this memory: `dirty` / `dirty_out` below hold real state
  int32_t pad[7];
};
constexpr uint32_t PLAN_MAGIC = 0x46565441u;

struct PlanView {
  PlanHeader* hdr;
  int32_t* order;    // [B] sequence ids, longest first (stable)
  int32_t* nactive;  // [J+1] sequences with len > t
  int32_t* len;      // [B]
  int32_t* seq_J;    // [B]
  int64_t* x_off;    // [B]
  int64_t* out_off;  // [B]
  int64_t* xo;       // [2][J][B] element offset of x row for (dir,t,sorted i), -1 = inactive
  int64_t* oo;       // [2][J][B] element offset of the output half-row
  // desc.out_pads_persist: what the LAST forward on this plan memory left in the caller's output -- rows [0, dirty[b]) of
  // sequence b may be non-zero in the buffer `dirty_out[b]` points into (everything beyond is zero); survives re-planning
  int32_t* dirty;      // [B]
  int64_t* dirty_out;  // [B] the sequence's first output row (address), 0 = nothing known
  size_t bytes;
};

static inline PlanView plan_view(const fvta_lstm_desc* d, void* p) {
  FvtaCarver c(p);
  PlanView v;
  v.hdr = c.take<PlanHeader>(1);
  v.order = c.take<int32_t>(d->B);
  v.nactive = c.take<int32_t>(d->J + 1);
  v.len = c.take<int32_t>(d->B);
  v.seq_J = c.take<int32_t>(d->B);
  v.x_off = c.take<int64_t>(d->B);
  v.out_off = c.take<int64_t>(d->B);
  v.xo = c.take<int64_t>((size_t)2 * d->J * d->B);
  v.oo = c.take<int64_t>((size_t)2 * d->J * d->B);
  v.dirty = c.take<int32_t>(d->B);
  v.dirty_out = c.take<int64_t>(d->B);
  v.bytes = c.off;
  return v;
}

// the bf16 engines (FVTA_BF16, FVTA_BF16X3) share kernels; xm = stored bf16 terms per operand value.  The split engine
// (precision bf16x3) stores TWO -- hi = bf16(v), lo = bf16(v - hi) -- for every MFMA operand (xs, hs, wt, wb, dz, wf), in the
// il32 layout (gemm_bf16.h: element e of a row at (e / 32) * 64 + e % 32, its lo term 32 elements further), and every k-step
// multiplies THREE products from the four fragments it reads once: hi hi + hi lo + lo hi.  (Rounds 2-5 stored three terms
// per value and ran the plain kernels over a K extent three times as long: 1.5x the bytes staged and read per product.)
static inline bool lstm_is_bf(const fvta_lstm_desc* d) { return d->precision == FVTA_BF16 || d->precision == FVTA_BF16X3; }
static inline int lstm_xm(const fvta_lstm_desc* d) { return d->precision == FVTA_BF16X3 ? 2 : 1; }

// bf16 engine: internal input width = in + a ones column (dbias) + zero pad to a multiple of 32, so that
// every 32-deep k-tile is wholly x or wholly h
static inline int in_internal(const fvta_lstm_desc* d) { return (d->in + 1 + 31) / 32 * 32; }

// ------------------------------------------------------------- saved state --
struct SavedView {
  float* gates;  // [2][J][B][4][d] i, tanh(j), f, o activations (fp32 engine: overwritten by dz in backward)
  bf16_t* gatesb;  // bf16 engine: [2][J][B][d][4] (unit-major, the four gates of a unit adjacent), bf16
  float* cs;     // [2][J][B][d] cell state after step t
  // bf16 engine: MFMA operand shadows, dense per (direction, step) in sorted-row order like gates/cs/dz
  bf16_t* xs;    // [2][J][B][in_i] x at the position (dir, t) visits, a 1.0 column at `in` (dbias), zero pad
  bf16_t* hs;    // [2][J][B][d]    h_t
  size_t bytes;
};
static inline SavedView saved_view(const fvta_lstm_desc* d, void* p) {
  FvtaCarver c(p);
  SavedView s;
  s.gates = s.cs = nullptr;
  s.xs = s.hs = s.gatesb = nullptr;
  if (d->training) {
    if (d->precision == FVTA_BF16)
      s.gatesb = c.take<bf16_t>((size_t)2 * d->J * d->B * 4 * d->d);
    else  // fp32 engine: [..][4][d]; split-bf16 engine: unit-major [..][d][4] like gatesb, fp32
      s.gates = c.take<float>((size_t)2 * d->J * d->B * 4 * d->d);
    s.cs = c.take<float>((size_t)2 * d->J * d->B * d->d);
  }
  if (lstm_is_bf(d)) {
    s.xs = c.take<bf16_t>((size_t)2 * d->J * d->B * in_internal(d) * lstm_xm(d));
    s.hs = c.take<bf16_t>((size_t)2 * d->J * d->B * d->d * lstm_xm(d));
  }
  s.bytes = c.off < 256 ? 256 : c.off;
  return s;
}

static inline int dw_tgroup(const fvta_lstm_desc* d) {
  // steps per split-K slice of the weight-gradient GEMM: keep <= 16 slices per direction
  int g = (d->J + 15) / 16;
  return g < 1 ? 1 : g;
}
static inline int dw_nsplit(const fvta_lstm_desc* d) { return (d->J + dw_tgroup(d) - 1) / dw_tgroup(d); }
static inline int kpad8(const fvta_lstm_desc* d) { return in_internal(d) + d->d; }  // bf16 engine K

struct WorkView {
  float* cstate;   // [2][B][d] running cell state (inference) / dc (backward)
  float* dh_rec;   // [2][B][d]
  float* slabs;    // [2*nsplit][(in+d+1)][4d] split-K partials of dW
  // bf16 engine only
  bf16_t* wt[2];   // [4d][in_i+d]  kernel^T in the internal row order, k contiguous (forward B operand)
  bf16_t* wb[2];   // [in_i+d][4d]  kernel in the internal row order (backward B operand)
  bf16_t* dzb;     // [2][J][B][d][4] gate pre-activation gradients, unit-major like gatesb (wb's k order matches)
  bf16_t* wf[2];   // [4d * (in_i+d)] the kernel in MFMA-fragment order + split bias rows (lstm_wreg.hip)
  size_t bytes;
};
static inline WorkView work_view(const fvta_lstm_desc* d, void* p) {
  FvtaCarver c(p);
  WorkView w;
  w.cstate = c.take<float>((size_t)2 * d->B * d->d);
  w.dh_rec = c.take<float>((size_t)2 * d->B * d->d);
  const size_t slab_rows = lstm_is_bf(d) ? (size_t)kpad8(d) : (size_t)(d->in + d->d + 1);
  w.slabs = c.take<float>((size_t)2 * dw_nsplit(d) * slab_rows * 4 * d->d);
  w.wt[0] = w.wt[1] = w.wb[0] = w.wb[1] = nullptr;
  w.dzb = nullptr;
  w.wf[0] = w.wf[1] = nullptr;
  if (lstm_is_bf(d)) {
    const size_t xm = (size_t)lstm_xm(d);
    for (int i = 0; i < 2; ++i) {
      w.wt[i] = c.take<bf16_t>((size_t)4 * d->d * kpad8(d) * xm);
      w.wb[i] = c.take<bf16_t>((size_t)kpad8(d) * 4 * d->d * xm);
    }
    if (d->training) w.dzb = c.take<bf16_t>((size_t)2 * d->J * d->B * 4 * d->d * xm);
    for (int i = 0; i < 2; ++i) w.wf[i] = c.take<bf16_t>((size_t)4 * d->d * kpad8(d) * xm);
  }
  w.bytes = c.off;
  return w;
}

// ----------------------------------------------------------- kernel args ----
struct StepArgs {
  PlanView plan;
  const float* x;
  float* out;
  const float* W[2];
  const bf16_t* Wt[2];  // bf16 engine
  const bf16_t* Wf[2];  // bf16 engine: fragment-order shadow for the weights-in-registers step kernel (null: shape not built)
  const bf16_t* xs;     // bf16 engine
  bf16_t* hs;           // bf16 engine
  const float* bias[2];
  float* gates;   // may be null (inference, bf16 engine)
  bf16_t* gatesb; // bf16 engine, training
  float* cs;      // may be null
  float* cstate;  // used when cs is null
  int t, B, J, in, d, Kp;
  int dbg;  // diagnostics only (-DFVTA_DIAG builds, FVTA_DEBUG_SKIP): fp32 engine ablations
  int xm;   // bf16 engines: stored bf16 terms per operand value (1; 2 in the split engine, which saves fp32 gates in `gates`)
  int nt;   // bf16 engine: stream-once data (saved gates, cell states, fp32 h rows) with non-temporal stores (measured: no
            // effect on the forward step; 0)
  int64_t out_skip;  // fvta_lstm_desc.out_skip: output half-rows at element offsets below this are NOT stored
                     // (their readers take the bf16 shadow rows: fvta_lstm_shadow_rows)
};

struct GateBwdArgs {
  PlanView plan;
  const float* d_out;
  float* gates;
  const bf16_t* gatesb;  // bf16 engine
  bf16_t* dzb;    // null: dz overwrites gates in place (fp32 engine)
  const float* cs;
  float* dc;      // [2][B][d]
  float* dh_rec;  // [2][B][d]
  int t, B, J, d;
};

struct StepBwdArgs {
  PlanView plan;
  const float* dz;      // fp32 engine: the saved gates buffer
  const bf16_t* dzb;    // bf16 engine
  const float* W[2];
  const bf16_t* Wb[2];  // bf16 engine
  float* dx;            // may be null
  float* dh_rec;
  int t, B, J, in, d, in_i;
  int dbg;
};

struct DwArgs {
  PlanView plan;
  const float* x;
  const float* out;
  const float* dz;
  const bf16_t* dzb;
  const bf16_t* xs;
  const bf16_t* hs;
  float* slabs;
  int B, J, in, d, tgroup, nsplit, in_i;
  int xm;  // stored bf16 terms per operand value (2: the split engine's il32 columns, lstm_dw_x2)
};

#ifdef __HIPCC__
// Fused gate epilogue shared by both engines.  The block tile is BM sorted sequences x
// (4 gates x 32 units) and a wave owns TM*32 rows x all four gate tiles, so lane (col = lane&31)
// holds z_i, z_j, z_f, z_o of the same (row, unit) in acc[ti][0..3].
// BasicLSTMCell: c' = c*sig(f+1) + sig(i)*tanh(j); h' = tanh(c')*sig(o).
// s_oo[row] = element offset of this step's output half-row (-1: inactive row), staged in LDS by
// the caller so that the only global loads here (c_{t-1}) are unconditional and issued 16 at a
// time (a load under a per-row branch would cost one serialised round trip per row).
template <class Mma>
__device__ __forceinline__ void lstm_gate_epilogue(const Mma& mma, const StepArgs& a, int dir, int m0, int u0,
                                                   int nact, size_t trow, const int64_t* s_oo) {
  static_assert(Mma::TN == 4 && Mma::WAVES_N == 1, "wave tile must span the four gate strips");
  const int d = a.d, t = a.t;
  const float* __restrict__ bias = a.bias[dir];
  const int u = u0 + mma.l31;
  const float bi = bias[u], bj = bias[d + u], bf = bias[2 * d + u], bo = bias[3 * d + u];
  const float* cprev_src = a.cs ? a.cs + (trow - a.B) * (size_t)d : a.cstate + (size_t)dir * a.B * d;
#pragma unroll
  for (int ti = 0; ti < Mma::TM; ++ti) {
    float cp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = min(m0 + mma.row_of(ti, r), nact - 1);  // clamped: always a valid row
      cp[r] = (t > 0 && !(a.dbg & 64)) ? cprev_src[(size_t)i * d + u] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mma.row_of(ti, r);
      const int i = m0 + row;
      const float ig = fvta_sigmoid(mma.acc[ti][0][r] + bi);
      const float jg = fvta_tanh(mma.acc[ti][1][r] + bj);
      const float fg = fvta_sigmoid(mma.acc[ti][2][r] + bf + 1.0f);  // forget_bias
      const float og = fvta_sigmoid(mma.acc[ti][3][r] + bo);
      const float c = cp[r] * fg + ig * jg;
      const float h = fvta_tanh(c) * og;
      if (i < nact) {
        if (a.cs) {
          if (!(a.dbg & 16)) a.cs[(trow + i) * d + u] = c;
          if (a.dbg & 8) {
          } else if (a.gatesb) {  // unit-major [row][u][i,j,f,o]: one 8-byte store per lane, 256 B per half-wave
            bf16x4 pk;
            pk[0] = (short)f2bf(ig);
            pk[1] = (short)f2bf(jg);
            pk[2] = (short)f2bf(fg);
            pk[3] = (short)f2bf(og);
            *reinterpret_cast<bf16x4*>(a.gatesb + (trow + i) * (size_t)(4 * d) + 4 * u) = pk;
          } else {
            float* g = a.gates + (trow + i) * (size_t)(4 * d) + u;
            g[0] = ig;
            g[d] = jg;
            g[2 * d] = fg;
            g[3 * d] = og;
          }
        } else {
          a.cstate[((size_t)dir * a.B + i) * d + u] = c;
        }
        if (!(a.dbg & 4) && s_oo[row] >= a.out_skip) a.out[s_oo[row] + u] = h;
        if (a.hs && !(a.dbg & 32)) a.hs[(trow + i) * d + u] = f2bf(h);  // bf16 shadow: next step's MFMA operand
      }
    }
  }
}
#endif

#ifdef __HIPCC__
// The same gate math with every global access staged through LDS (bf16 engine).  In the MFMA C layout a lane holds
// ONE unit of 16 rows, so the direct epilogue above moves 4 (or 2) bytes per lane and instruction: 160 VMEM
// instructions per wave and tile.  The CU's address unit takes ~40-50 cycles per VMEM wave-instruction whatever its
// width, which made the epilogue as long as the k-loop.  Here a wave transposes each 32-row x 32-unit plane through
// a private LDS scratch (scr, >= 8704 B per wave, the k-loop's stage buffers after the closing barrier) and moves
// 16 bytes per lane: 44 VMEM instructions per wave and tile.
//   cprev[ti][it]: c_{t-1} of rows ti*32 + it*8 + (lane>>3), units 4*(lane&7)..+3 -- loaded by the caller BEFORE the
//   k-loop (latency hidden), ignored when t == 0.
__device__ __forceinline__ void st16(float* p, const f32x4 v, bool nt) {
  if (nt)
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else
    *reinterpret_cast<f32x4*>(p) = v;
}

template <class Mma>
__device__ __forceinline__ void lstm_gate_epilogue_staged(const Mma& mma, const StepArgs& a, int dir, int m0, int u0,
                                                          int nact, size_t trow, const int64_t* s_oo,
                                                          const f32x4 (&cprev)[Mma::TM][4], char* scr, int t) {
  static_assert(Mma::TN == 4 && Mma::WAVES_N == 1, "wave tile: 32 TM rows x the four gate strips");
  constexpr int LDP = 36;  // floats per staged fp32 row (32 + pad, keeps 16-byte alignment)
  const int d = a.d, lane = mma.lane;  // t: the step (a.t for the per-step kernels; the sequence-stationary kernel loops over it)
  const float* __restrict__ bias = a.bias[dir];
  const int u = u0 + mma.l31;
  const float bi = bias[u], bj = bias[d + u], bf = bias[2 * d + u], bo = bias[3 * d + u];
  float* pl = reinterpret_cast<float*>(scr);
  bf16_t* plh = reinterpret_cast<bf16_t*>(scr);
  const int io_row = lane >> 3, io_c4 = lane & 7;  // fp32 planes: 8 lanes x 16 B per row, 8 rows per instruction
  auto wave_sync = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
#pragma unroll
  for (int ti = 0; ti < Mma::TM; ++ti) {
    const int wrow0 = mma.wave * Mma::WROWS + ti * 32;  // first tile row of this 32-row slice of the wave tile
    // ---- c_{t-1}: row-contiguous registers -> LDS -> MFMA layout
    float cp[16];
    if (t > 0) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(&pl[(it * 8 + io_row) * LDP + 4 * io_c4]) = cprev[ti][it];
      wave_sync();
#pragma unroll
      for (int r = 0; r < 16; ++r) cp[r] = pl[((r & 3) + 8 * (r >> 2) + 4 * mma.hf) * LDP + mma.l31];
      wave_sync();
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) cp[r] = 0.f;
    }
    float cv[16], hv[16];
    bf16x4 gv[16];
    const bool x3 = a.xm == 2;  // the split engine
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ig = fvta_sigmoid(mma.acc[ti][0][r] + bi);
      const float jg = fvta_tanh(mma.acc[ti][1][r] + bj);
      const float fg = fvta_sigmoid(mma.acc[ti][2][r] + bf + 1.0f);  // forget_bias
      const float og = fvta_sigmoid(mma.acc[ti][3][r] + bo);
      cv[r] = cp[r] * fg + ig * jg;
      hv[r] = fvta_tanh(cv[r]) * og;
      gv[r][0] = (short)f2bf(ig);
      gv[r][1] = (short)f2bf(jg);
      gv[r][2] = (short)f2bf(fg);
      gv[r][3] = (short)f2bf(og);
      if (x3 && a.gates) {  // split engine: fp32 gates, unit-major [row][u][i,j,f,o] -- 16 B per lane, 512 B per half-wave
        const int i = m0 + wrow0 + (r & 3) + 8 * (r >> 2) + 4 * mma.hf;
        if (i < nact) *reinterpret_cast<f32x4*>(a.gates + (trow + i) * (size_t)(4 * d) + 4 * u) = f32x4{ig, jg, fg, og};
      }
    }
    // ---- c_t -> cs (training) or the rolling cstate
    {
#pragma unroll
      for (int r = 0; r < 16; ++r) pl[((r & 3) + 8 * (r >> 2) + 4 * mma.hf) * LDP + mma.l31] = cv[r];
      wave_sync();
      float* dst = a.cs ? a.cs + trow * d : a.cstate + (size_t)dir * a.B * d;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int lr = it * 8 + io_row, i = m0 + wrow0 + lr;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&pl[lr * LDP + 4 * io_c4]);
        if (i < nact) st16(dst + (size_t)i * d + u0 + 4 * io_c4, v, a.nt != 0);
      }
      wave_sync();
    }
    // ---- h_t -> the caller's output rows (fp32)
    {
#pragma unroll
      for (int r = 0; r < 16; ++r) pl[((r & 3) + 8 * (r >> 2) + 4 * mma.hf) * LDP + mma.l31] = hv[r];
      wave_sync();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int lr = it * 8 + io_row;
        const int64_t oo = s_oo[wrow0 + lr];
        const f32x4 v = *reinterpret_cast<const f32x4*>(&pl[lr * LDP + 4 * io_c4]);
        if (oo >= a.out_skip) {   // (inactive rows are -1, out_skip >= 0)
          float* o = a.out + oo + u0 + 4 * io_c4;
          if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) {
            st16(o, v, a.nt != 0);
          } else {  // an output row that is not 16-byte aligned
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
          }
        }
      }
      wave_sync();
    }
    // ---- bf16 shadow of h_t (next step's MFMA operand): 4 lanes x 16 B per row, 16 rows per instruction.  Split engine:
    // two terms per unit in the il32 layout -- the plane's 32 units are one group: [hi 32 | lo 32] at 2 u0
    if (a.hs) {
      constexpr int LDH = 40;  // bf16 per staged row (32 + pad)
      const size_t hld = (size_t)d * a.xm;
#pragma unroll
      for (int term = 0; term < 2; ++term) {
        if (term >= a.xm) break;
        const bool lo = term == 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bf16_t hi = f2bf(hv[r]);
          plh[((r & 3) + 8 * (r >> 2) + 4 * mma.hf) * LDH + mma.l31] = lo ? f2bf(hv[r] - bf2f(hi)) : hi;
        }
        wave_sync();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int lr = it * 16 + (lane >> 2), c8 = lane & 3, i = m0 + wrow0 + lr;
          const f32x4 v = *reinterpret_cast<const f32x4*>(&plh[lr * LDH + 8 * c8]);
          if (i < nact) *reinterpret_cast<f32x4*>(a.hs + (trow + i) * hld + (size_t)a.xm * u0 + 32 * term + 8 * c8) = v;
        }
        wave_sync();
      }
    }
    // ---- gates, unit-major [row][u][i,j,f,o] bf16: 16 lanes x 16 B per row, 4 rows per instruction
    if (a.gatesb) {
      constexpr int LDG = 136;  // bf16 per staged row (128 + pad)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        *reinterpret_cast<bf16x4*>(&plh[((r & 3) + 8 * (r >> 2) + 4 * mma.hf) * LDG + 4 * mma.l31]) = gv[r];
      wave_sync();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int lr = it * 4 + (lane >> 4), c8 = lane & 15, i = m0 + wrow0 + lr;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&plh[lr * LDG + 8 * c8]);
        if (i < nact) st16(reinterpret_cast<float*>(a.gatesb + (trow + i) * (size_t)(4 * d) + 4 * u0 + 8 * c8), v, a.nt != 0);
      }
      wave_sync();
    }
  }
}

#endif

// bf16 engine launchers (lstm_bf16.hip)
void launch_cvt_weights_bf16(const float* W, bf16_t* wt, bf16_t* wb, int in, int in_i, int d, int xm, hipStream_t s);
void launch_cvt_x_bf16(const PlanView& pv, const float* x, bf16_t* xs, int B, int J, int in, int in_i, int xm, hipStream_t s);
void launch_step_fwd_bf16(const StepArgs& a, hipStream_t s);
// weights-in-registers forward step (lstm_wreg.hip): false = shape not built, the tiled kernel runs instead
int wreg_nct(int in_i, int d);
int wreg_mode();
bool wreg_x3_built(int in_i, int d);  // the split engine's weights-in-registers forward is built for this shape
void launch_cvt_weights_frag(const float* W, const float* bias, bf16_t* wf, int in, int in_i, int d, int xm, hipStream_t s);
bool launch_step_fwd_wreg(const StepArgs& a, hipStream_t s);
struct FusedBwdArgs {
  PlanView plan;
  const bf16_t* Wb[2];
  const bf16_t* gatesb;
  const float* cs;
  const float* d_out;
  bf16_t* dzb;
  float* dc;  // [2][B][d]
  float* dx;  // lstm_dx only
  int t, B, J, in, d, in_i;
  int nact_hint;         // active sequences of step t as the HOST knows them (fvta_bilstm_bwd_hint), -1: unknown -- picks the step's tile
  const float* gates32;  // split engine: fp32 gates, unit-major (gatesb null)
  int xm;                // bf16 terms per operand value
  int dx_accumulate;     // lstm_dx: 1 = the forward direction's launch adds to dx too (the C ABI's contract: dx is accumulated)
  int dx_both;           // lstm_dx: set by its launcher -- the first launch sums both directions when they share one input
};
void launch_bwd_fused_bf16(const FusedBwdArgs& a, hipStream_t s);
bool launch_bwd_wreg(const FusedBwdArgs& a, hipStream_t s);  // lstm_wreg_bwd.hip: steps with few rows, false: not taken
void launch_dx_bf16(const FusedBwdArgs& a, hipStream_t s);
bool dx_writes_whole_rows(const FusedBwdArgs& a);  // lstm_dx_bf16 writes every element of a valid row once (dx_overwrite)
void launch_dw_bf16(const DwArgs& a, hipStream_t s);
void launch_dw_reduce_bf16(const float* slabs, int nslab, int in, int in_i, int d, float* dW, float* dbias, hipStream_t s);
int test_gemm_bf16(int layout, int M, int N, int K, const float* A, const float* B, float* C, hipStream_t s);

}  // namespace fvta
