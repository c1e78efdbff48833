// bi-LSTM step kernels on the bf16 MFMA engine (BASELINE.json configs[2]: bf16
// compute, fp32 accumulate).  Same decomposition and epilogues as lstm.hip.
//
// Every MFMA operand is a DENSE bf16 matrix per (direction, step) in sorted-row
// order, the layout gates / cs / dz already use:
//   xs [2][J][B][in_i]  x at the position that (dir, t) visits, then a 1.0 column
//                       (its weight-gradient row is dbias), zero padded so that
//                       in_i is a multiple of 32 (k-tiles are wholly x or wholly h);
//   hs [2][J][B][d]     h_t, written by the gate epilogue next to the fp32 h;
//   dz [2][J][B][d][4]  gate pre-activation gradients, unit-major (so are the saved gate activations:
//                       one 8-byte access per cell); wb's columns follow the same 4u+g order;
//   wt [4d][in_i+d], wb [in_i+d][4d]  kernel shadows in that internal row order.
// Active rows of a step are a prefix and nest across steps, so row i of step t-1
// IS the h_{t-1} of row i of step t: no gathers anywhere, every tile streams
// through the DMA pipeline of gemm_bf16.h.  grid.x (row tile) is padded to a
// multiple of 8 so that the workgroups sharing a row tile share an XCD's L2.
// Gates, c, h, accumulators and weight-gradient slabs stay fp32; the saved gate
// activations are bf16 in this engine.
#include "gemm_bf16.h"
#include <type_traits>
#include "lstm_common.h"

#ifndef FVTA_LSTM_DX_FUSED_DEFAULT
#define FVTA_LSTM_DX_FUSED_DEFAULT 0
#endif
#ifndef FVTA_LSTM_SEQ_DEFAULT
#define FVTA_LSTM_SEQ_DEFAULT 0
#define FVTA_LSTM_FWD_DIRECT_DEFAULT 0
#define FVTA_LSTM_SMALL_ROWS_DEFAULT 1
#ifndef FVTA_LSTM_SMALL_SK_DEFAULT
#define FVTA_LSTM_SMALL_SK_DEFAULT 0  // 4: measured slower in the step (its 144 KB of LDS wait for a whole free CU), see launch
#endif
#define FVTA_LSTM_DX_2PASS_DEFAULT 0
#endif
#ifndef FVTA_GLDS_SP_DEFAULT
#define FVTA_GLDS_SP_DEFAULT 1
#endif
#ifndef FVTA_TILE128_DEFAULT
#define FVTA_TILE128_DEFAULT 0
#endif

namespace fvta {

static inline int pad8(int v) { return (v + 7) / 8 * 8; }

// Which kernels take the 256 x 256 / four-wave (128 x 128 wave tile) configuration: bit 0 forward step, 1 fused
// backward step, 2 dx, 3 weight gradient.  FVTA_LSTM_TILE128 overrides the built-in choice (measurement switch).
// Which kernels run the software-pipelined main loop: bit 0 forward step (and the row-image test GEMM), 1 fused backward
// step, 2 dx, 3 weight gradient (and the k-major test GEMM).  FVTA_GLDS_SP overrides the built-in choice.
int glds_sp_mask() {
  static const int m = [] {
    const char* e = getenv("FVTA_GLDS_SP");
    return e ? atoi(e) : FVTA_GLDS_SP_DEFAULT;
  }();
  return m;
}

// FVTA_LSTM_SMALL_ROWS: calls with <= 128 sequences (the photo cell) run their backward step on 64- / 128-row block tiles
static bool small_rows() {
  static const bool on = [] {
    const char* e = getenv("FVTA_LSTM_SMALL_ROWS");
    return e ? e[0] == '1' : (FVTA_LSTM_SMALL_ROWS_DEFAULT != 0);
  }();
  return on;
}

static int tile128_mask() {
  static const int m = [] {
    const char* e = getenv("FVTA_LSTM_TILE128");
    return e ? atoi(e) : FVTA_TILE128_DEFAULT;
  }();
  return m;
}

template <class K>
static void allow_big_lds(K kernel, int bytes) {  // > 64 KB of dynamic LDS must be opted into, per kernel symbol
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// ---- weight shadows in the internal row order [x rows | ones row | zero pad | h rows] -----------
// kernel [in+d][N4] fp32 -> wb [in_i+d][N4] bf16, wt [N4][in_i+d] bf16.  grid (N4/32, (in_i+d)/32), 256 threads
__global__ void cvt_weights_kernel(const float* __restrict__ W, bf16_t* __restrict__ wt, bf16_t* __restrict__ wb,
                                   int in, int in_i, int d) {
  __shared__ float tile[32][33];
  const int N4 = 4 * d, Ki = in_i + d;
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    const int src = k < in ? k : (k >= in_i ? in + (k - in_i) : -1);  // internal row -> kernel row
    const float v = src >= 0 ? W[(size_t)src * N4 + n] : 0.f;
    tile[r][tx] = v;
    wb[(size_t)k * N4 + 4 * (n % d) + n / d] = f2bf(v);  // column g*d+u -> 4u+g: dz rows are unit-major
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) wt[(size_t)(n0 + r) * Ki + k0 + tx] = f2bf(tile[tx][r]);
}

void launch_cvt_weights_bf16(const float* W, bf16_t* wt, bf16_t* wb, int in, int in_i, int d, hipStream_t s) {
  hipLaunchKernelGGL(cvt_weights_kernel, dim3(4 * d / 32, (in_i + d) / 32), dim3(256), 0, s, W, wt, wb, in, in_i, d);
}

// ---- input shadow: xs[dir][t][i][:] for every active (dir, t, i).  grid (ceil(B/4), J) ----------
// One wave per (sorted row i, position pos): the row is read ONCE and written to both places it is needed -- step pos
// of the forward direction and step len - 1 - pos of the backward direction (reverse_sequence).
__global__ void cvt_x_kernel(PlanView pv, const float* __restrict__ x, bf16_t* __restrict__ xs, int B, int J, int in,
                             int in_i) {
  const int pos = blockIdx.y;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= pv.nactive[pos]) return;  // len_i <= pos
  const int len = pv.len[pv.order[i]];
  const float* src = x + pv.xo[(size_t)pos * B + i];  // forward direction, step pos = position pos
  const int64_t bw_delta = pv.hdr->x_bw_delta;        // != 0: the backward direction has its own input (fvta_lstm_plan_xdir)
  bf16_t* dst_fw = xs + ((size_t)pos * B + i) * in_i;
  bf16_t* dst_bw = xs + (((size_t)J + (len - 1 - pos)) * B + i) * in_i;
  for (int c = lane * 4; c < in_i; c += 256) {
    bf16x4 o, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = c + e;
      // ones columns at `in` (its weight-gradient row is dbias) and `in + 1` (the second bias term of lstm_wreg.hip;
      // in % 4 == 0, so in + 1 < in_i always)
      o[e] = k < in ? (short)f2bf(src[k]) : ((k == in || k == in + 1) ? (short)0x3f80 : (short)0);
      ob[e] = (bw_delta && k < in) ? (short)f2bf(src[bw_delta + k]) : o[e];
    }
    *reinterpret_cast<bf16x4*>(dst_fw + c) = o;
    *reinterpret_cast<bf16x4*>(dst_bw + c) = ob;
  }
}

void launch_cvt_x_bf16(const PlanView& pv, const float* x, bf16_t* xs, int B, int J, int in, int in_i, hipStream_t s) {
  hipLaunchKernelGGL(cvt_x_kernel, dim3((B + 3) / 4, J), dim3(256), 0, s, pv, x, xs, B, J, in, in_i);
}

__device__ unsigned long long g_lstm_stamps[512];  // diagnostics (FVTA_DEBUG_SKIP & 32768)

// ------------------------------------------------------------ forward step --
// z = [xs_t | hs_{t-1}] * wt^T over the 4 gate strips of 32 units per wave column.
// One block tile of rows [m0, m0 + Cfg::BM) x the 32 WN units from ub.
// LEAN: no diagnostic switches, software-pipelined main loop only (the sequence-stationary kernel: its two loops around
// the tile leave no registers for code paths that never run)
// DIRECT: transposed accumulators (MmaBT SWAP) and the LDS-free epilogue (lstm_gate_epilogue_direct)
template <int WN, int TM, int WM, bool LATE_CPREV = (TM == 4), bool LEAN = false, bool DIRECT = false>
__device__ __forceinline__ void lstm_step_fwd_tile(const StepArgs& a, bf16_t* smem_h, int64_t* s_oo, int m0, int ub,
                                                   int dir, int nact, int t) {
  typedef TileCfgT<WN, TM, WM> Cfg;
  const int tid = threadIdx.x;
  const int d = a.d, in_i = a.Kp - a.d;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  __syncthreads();  // a workgroup that runs several tiles: the previous tile's epilogue is done with s_oo and the stage buffers
  for (int r = tid; r < Cfg::BM; r += Cfg::NT) s_oo[r] = (m0 + r < nact) ? a.plan.oo[trow + m0 + r] : -1;

  MmaBT<WN, TM, WM, DIRECT> mma;
  mma.init(tid);
  const int u0 = ub + 32 * mma.wn;
  if constexpr (DIRECT) lstm_direct_bias_init(mma, a.bias[dir], d, u0);
  // A rows m0.. of xs[dir][t] (nact rows) and of hs[dir][t-1]; B rows = the 4 gate strips of wt per wave column
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.xs + trow * in_i, (unsigned)nact * in_i * 2);
  const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.hs + (t > 0 ? trow - a.B : trow) * d, (unsigned)nact * d * 2);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.Wt[dir], (unsigned)(4 * d) * a.Kp * 2);
  RowSrc<Cfg::A_GLDS> ax, ah;
  RowSrc<Cfg::B_GLDS> bw;
  ax.setup(mma.wave_all, mma.lane, m0, nact, in_i * 2);
  ah.setup(mma.wave_all, mma.lane, m0, nact, d * 2);
#pragma unroll
  for (int j = 0; j < Cfg::B_GLDS; ++j) {  // B row r = wave column r>>7, gate strip (r>>5)&3, unit ub + 32 (r>>7) + (r&31)
    const int U = (mma.wave_all * Cfg::B_GLDS + j) * 64 + mma.lane;
    const int r = U >> 2, c = (U & 3) ^ ((r >> 2) & 3);
    bw.voff[j] = (unsigned)(((r >> 5) & 3) * d + ub + 32 * (r >> 7) + (r & 31)) * (unsigned)(a.Kp * 2) + 16u * c;
  }
  const int nx = in_i / 32, nt = (t == 0) ? nx : nx + d / 32;
  auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
    if (tile < nx)
      ax.issue(rx, As, mma.wave_all, tile * 64);
    else
      ah.issue(rh, As, mma.wave_all, (tile - nx) * 64);
    bw.issue(rw, Bs, mma.wave_all, tile * 64);
  };
  // FVTA_DEBUG_SKIP & 32768: one wave of one workgroup stamps the shader clock (tools/lstm_phases.py)
  const int lin_wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int dbg = LEAN ? 0 : a.dbg;
  unsigned long long* st = (!LEAN && (dbg & 32768) && lin_wg == ((dbg >> 16) & 0xFFF) && tid == 0 && t == 5) ? g_lstm_stamps : nullptr;
  if (st) st[0] = __builtin_readcyclecounter();
  // c_{t-1} of the wave's rows, row-contiguous (16 B per lane), requested before the k-loop hides their latency
  // (TM = 4: after it -- 64 more live registers across the k-loop spill next to the 256 accumulators)
  f32x4 cprev[TM][4];
  auto load_cprev = [&] {
    if (t > 0) {
      const float* src = a.cs ? a.cs + (trow - a.B) * (size_t)d : a.cstate + (size_t)dir * a.B * d;
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          // staged: rows it*8 + lane/8, units 4 (lane%8)..; direct: row lane%32, units 8 it + 4 (lane/32)..
          const int lr = DIRECT ? mma.l31 : it * 8 + (mma.lane >> 3), lu = DIRECT ? 8 * it + 4 * mma.hf : 4 * (mma.lane & 7);
          const int i = min(m0 + mma.wave * (32 * TM) + ti * 32 + lr, nact - 1);  // clamped: valid row
          const f32x4* cp = reinterpret_cast<const f32x4*>(src + (size_t)i * d + u0 + lu);
          cprev[ti][it] = a.nt ? __builtin_nontemporal_load(cp) : *cp;
        }
    }
  };
  if (!LATE_CPREV) load_cprev();
  if (LEAN)
    glds_mainloop_sp<false>(mma, issue, nt, smem_h);
  else if (!(dbg & 1))
    glds_mainloop<false>(mma, issue, nt, smem_h, st ? st + 8 : nullptr, ((dbg >> 17) & 3) | (a.sp ? 4 : 0));
  if (st) st[1] = __builtin_readcyclecounter();
  if (LATE_CPREV) load_cprev();
  __syncthreads();  // s_oo visible; every wave is done with the stage buffers, which become the epilogue's scratch
  if (!(dbg & 2)) {
    if constexpr (DIRECT)
      lstm_gate_epilogue_direct(mma, a, dir, m0, u0, nact, trow, s_oo, cprev, t);
    else
      lstm_gate_epilogue_staged(mma, a, dir, m0, u0, nact, trow, s_oo, cprev,
                                reinterpret_cast<char*>(smem_h) + mma.wave_all * 9216, t);
  }
  if (st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[2] = __builtin_readcyclecounter();
    st[3] = nt;
  }
}

// grid (pad8(row tiles), d/(32 WN), 2), 64 * 8/TM * WN threads.  Row tiles [0, nbig) are 256 rows; with TM = 2, WN = 1
// the tiles from nbig on are HALF tiles (128 rows, wave tile 32 x 128): the 256 x 128 tiles of a step do not divide
// into whole rounds of the chip's workgroup slots (metric shape: 1632 tiles on 512 slots = 3.19 rounds, and the partial
// round costs a full tile time), so the launcher gives whole rounds to full tiles and covers the remaining rows with
// units of half the duration.
template <int WN, int TM, int DIRECT = 0>  // DIRECT 1: LDS-free epilogue; 2: the same with c_{t-1} loaded after the k-loop
__global__ __launch_bounds__((TileCfgT<WN, TM>::NT), (WN == 1 ? 2 : 1)) void lstm_step_fwd_bf16(StepArgs a, int nbig) {
  typedef TileCfgT<WN, TM> Cfg;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  int64_t* s_oo = reinterpret_cast<int64_t*>(smem_h + Cfg::STAGES * Cfg::STAGE_ELEMS);  // [256], same array
  const int dir = blockIdx.z;
  const int nact = a.plan.nactive[a.t];
  const int ub = blockIdx.y * 32 * WN;  // first unit of the block; wave column wn owns units ub + 32 wn ..
  if constexpr (WN == 1 && TM == 2) {
    if ((int)blockIdx.x >= nbig) {
      const int m0 = nbig * 256 + ((int)blockIdx.x - nbig) * 128;
      if (m0 >= nact) return;
      lstm_step_fwd_tile<1, 1, 4, DIRECT == 2, false, DIRECT != 0>(a, smem_h, s_oo, m0, ub, dir, nact, a.t);
      return;
    }
  }
  const int m0 = blockIdx.x * Cfg::BM;
  if (m0 >= nact) return;
  lstm_step_fwd_tile<WN, TM, 8 / TM, (TM == 4) || DIRECT == 2, false, DIRECT != 0>(a, smem_h, s_oo, m0, ub, dir, nact, a.t);
}

int lstm_read_stamp(int i, long long* v) {
  if (i < 0 || i >= 512) return FVTA_ERR_INVALID_ARG;
  unsigned long long x = 0;
  if (hipMemcpyFromSymbol(&x, HIP_SYMBOL(g_lstm_stamps), 8, (size_t)i * 8, hipMemcpyDeviceToHost) != hipSuccess) return FVTA_ERR_INVALID_ARG;
  *v = (long long)x;
  return FVTA_OK;
}

// ------------------------------------------------ sequence-stationary forward --
// ONE launch for all J steps: a workgroup owns 128 sorted sequences of one direction for the whole recurrence (sequences
// are independent, so there is no cross-workgroup dependency and no grid-wide step barrier) and walks the 4d gate
// columns in chunks of 128 units (512 columns: 8 waves as 2 x 4 wave tiles of 64 x 128, the same wave tile and gate
// epilogue as the per-step kernel).  h_{t-1} and c_{t-1} come back through global memory, written by this very
// workgroup one step earlier (same CU, same L1/L2: a vmcnt(0) + barrier orders them); the weights stream from L2 once
// per step and workgroup.  What it buys over J launches of lstm_step_fwd_bf16: no launch skeletons and dispatch tails,
// the A operand re-read 4x instead of 16x, and -- workgroups drift apart -- the chip is no longer in one phase (all
// matrix pipe / all HBM) at a time.  grid (ceil(B/128), 2), 512 threads, one workgroup per CU.
// MEASURED (metric shape, text cell, train step): 7.60 ms against 5.08 ms for the 30 per-step launches (forward only:
// 6.00 vs 4.36 ms) -- it loses, and stays off (FVTA_LSTM_SEQ=1 selects it).  Why: 202 workgroups of 8 waves occupy
// 202 of 256 CUs with ONE workgroup each, so a CU's k-loop and epilogue strictly alternate (the per-step kernel keeps
// two 4-wave workgroups per CU and 1632 of them per step); the two loops around the tile push the kernel to 256 VGPRs
// + 690 B of scratch even with c_{t-1} loaded after the k-loop (exposed once per chunk); every step ends in a
// vmcnt(0) drain + barrier before h_{t-1} can be re-read.
__global__ __launch_bounds__(512, 1) void lstm_seq_fwd_bf16(StepArgs a) {
  typedef TileCfgT<4, 2, 2> Cfg;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  int64_t* s_oo = reinterpret_cast<int64_t*>(smem_h + Cfg::STAGES * Cfg::STAGE_ELEMS);
  const int dir = blockIdx.y, m0 = blockIdx.x * Cfg::BM;
  for (int t = 0; t < a.J; ++t) {
    const int nact = a.plan.nactive[t];
    if (m0 >= nact) break;  // sorted by length: once the tile's first row has ended, all of it has, for good
    for (int ub = 0; ub < a.d; ub += 128) lstm_step_fwd_tile<4, 2, 2, true, true>(a, smem_h, s_oo, m0, ub, dir, nact, t);
    // this step's h shadow / cell states are read back by the next one (DMA and plain loads through the same L1/L2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

bool launch_seq_fwd_bf16(const StepArgs& a_, hipStream_t s) {
  static const int mode = [] {
    const char* e = getenv("FVTA_LSTM_SEQ");
    return e ? atoi(e) : FVTA_LSTM_SEQ_DEFAULT;
  }();
  // (a call with few sequences would run on a handful of workgroups: the per-step kernel spreads a step over the
  // column blocks instead -- the photo cell's 64 rows took 9.0 ms here against 2.2 ms)
  if (!mode || a_.d % 128 != 0 || a_.B < 128 * 64) return false;
  StepArgs a = a_;
  a.sp = glds_sp_mask() & 1;
  a.t = 0;
  typedef TileCfgT<4, 2, 2> Cfg;
  constexpr int LDS = Cfg::LDS_BYTES + 256 * 8;
  allow_big_lds(lstm_seq_fwd_bf16, LDS);
  const dim3 grid((a.B + Cfg::BM - 1) / Cfg::BM, 2);
  hipLaunchKernelGGL(lstm_seq_fwd_bf16, grid, dim3(Cfg::NT), LDS, s, a);
  return true;
}

// Row tiles that get the full 256-row shape: as many as fill WHOLE rounds of the device's workgroup slots (2 per CU for
// the 256 x 128 kernel); the rest of the rows go to half tiles.  B is the call's sequence count (the active prefix of a
// ragged batch shrinks with t on the device; the split is a host-side choice made for the full prefix).
static int fwd_big_row_tiles(int B, int d) {
  static const int slots = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return 2 * (cus > 0 ? cus : 256);
  }();
  // FVTA_LSTM_FWD_TAIL: 0 (default) every row tile full size; 1 whole rounds of full tiles + half tiles for the rest;
  // 2 force half tiles on the upper half of the rows (tests).  Measured at the metric shape: 5.29 ms per 30 steps
  // either way -- the partial fourth round already runs faster than a full one, the kernel is bound by the aggregate
  // L2 -> LDS / HBM traffic, not by the number of rounds.
  static const int mode = [] {
    const char* e = getenv("FVTA_LSTM_FWD_TAIL");
    return e ? atoi(e) : 0;
  }();
  const int rt = (B + 255) / 256, per_rt = (d / 32) * 2;  // workgroups per row tile: column blocks x directions
  if (!mode) return rt;
  if (mode == 2) return rt / 2;  // tests: force half tiles on small shapes
  const long long total = (long long)rt * per_rt;
  if (total % slots == 0 || total < slots) return rt;     // whole rounds already / less than one round
  const int nbig = (int)((total / slots) * slots / per_rt);
  return nbig < rt ? nbig : rt;
}

void launch_step_fwd_bf16(const StepArgs& a_, hipStream_t s) {
  StepArgs a = a_;
  a.sp = glds_sp_mask() & 1;
  // the 256 x 256 tile (8 waves) halves the A-operand re-reads; it needs whole 64-unit column blocks
  // (measured on the metric shape: no faster than two 256 x 128 workgroups per CU, whose k-loops and epilogues
  // overlap better -- kept selectable: FVTA_LSTM_WIDE_TILE=1)
  static const bool wide = [] {
    const char* e = getenv("FVTA_LSTM_WIDE_TILE");
    return e && e[0] == '1';
  }();
  const int rt = (a.B + 255) / 256;
  if (a.d % 64 == 0 && (tile128_mask() & 1)) {  // 256 x 256 block tile on four waves of 128 x 128
    constexpr int LDS = TileCfgT<2, 4>::LDS_BYTES + 256 * 8;
    allow_big_lds(lstm_step_fwd_bf16<2, 4>, LDS);
    const dim3 grid(pad8(rt), a.d / 64, 2);
    hipLaunchKernelGGL((lstm_step_fwd_bf16<2, 4>), grid, dim3(256), LDS, s, a, rt);
  } else if (a.d % 64 == 0 && wide) {
    constexpr int LDS = TileCfgT<2>::LDS_BYTES + 256 * 8;
    allow_big_lds(lstm_step_fwd_bf16<2, 2>, LDS);
    const dim3 grid(pad8(rt), a.d / 64, 2);
    hipLaunchKernelGGL((lstm_step_fwd_bf16<2, 2>), grid, dim3(512), LDS, s, a, rt);
  } else {
    constexpr int LDS = TileCfgT<1>::LDS_BYTES + 256 * 8;
    allow_big_lds(lstm_step_fwd_bf16<1, 2>, LDS);
    const int nbig = fwd_big_row_tiles(a.B, a.d);
    const int nsmall = (a.B - nbig * 256 + 127) / 128;
    const dim3 grid(pad8(nbig + (nsmall > 0 ? nsmall : 0)), a.d / 32, 2);
    // FVTA_LSTM_FWD_DIRECT: transposed accumulators + the LDS-free epilogue (16-byte aligned cell-state / shadow / gate
    // rows: d % 4 == 0 always holds here, d is a multiple of 32)
    static const int direct = [] {
      const char* e = getenv("FVTA_LSTM_FWD_DIRECT");
      return e ? atoi(e) : FVTA_LSTM_FWD_DIRECT_DEFAULT;
    }();
    if (direct == 1) {
      allow_big_lds(lstm_step_fwd_bf16<1, 2, 1>, LDS);
      hipLaunchKernelGGL((lstm_step_fwd_bf16<1, 2, 1>), grid, dim3(256), LDS, s, a, nbig);
    } else if (direct == 2) {
      allow_big_lds(lstm_step_fwd_bf16<1, 2, 2>, LDS);
      hipLaunchKernelGGL((lstm_step_fwd_bf16<1, 2, 2>), grid, dim3(256), LDS, s, a, nbig);
    } else
      hipLaunchKernelGGL((lstm_step_fwd_bf16<1, 2>), grid, dim3(256), LDS, s, a, nbig);
  }
}

static constexpr int FWD_LDS = TileCfg::LDS_BYTES + 256 * 8;

// ----------------------------------------------------------- backward step --
// [dx_t | . | dh_{t-1}] = dz_t * wb^T : rows x (in_i + d), K = 4d.  grid (pad8(ceil(B/256)), (in_i+d)/128, 2)
__global__ __launch_bounds__(256, 2) void lstm_step_bwd_bf16(StepBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int tid = threadIdx.x, dir = blockIdx.z;
  const int m0 = blockIdx.x * TileCfg::BM, n0 = blockIdx.y * TileCfg::BN;
  const int nact = a.plan.nactive[a.t];
  if (m0 >= nact) return;
  const int d = a.d, in = a.in, t = a.t, in_i = a.in_i;
  const int NN = in_i + d, K = 4 * d;
  if (t == 0 && n0 >= in_i) return;                       // dh_{-1} is not needed
  if (a.dx == nullptr && n0 + TileCfg::BN <= in_i) return;  // nobody wants dx
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  MmaB mma;
  mma.init(tid);
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + trow * (size_t)K, (unsigned)nact * K * 2);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.Wb[dir], (unsigned)NN * K * 2);
  RowSrc<TileCfg::A_GLDS> az;
  RowSrc<TileCfg::B_GLDS> bw;
  az.setup(mma.wave, mma.lane, m0, nact, K * 2);
  bw.setup(mma.wave, mma.lane, n0, NN, K * 2);
  auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
    az.issue(rz, As, mma.wave, tile * 64);
    bw.issue(rw, Bs, mma.wave, tile * 64);
  };
  if (!(a.dbg & 128)) glds_mainloop<false>(mma, issue, K / 32, smem_h);
  if (a.dbg & 256) return;
  const bool xpart = n0 < in;  // tile holds dx columns: fetch the x row offsets, unconditionally and up front
#pragma unroll
  for (int ti = 0; ti < MmaB::TM; ++ti) {
    int64_t xos[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) xos[r] = xpart ? a.plan.xo[trow + min(m0 + mma.row_of(ti, r), nact - 1)] : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = m0 + mma.row_of(ti, r);
      if (i >= nact) continue;
      const int64_t xo = xos[r];
#pragma unroll
      for (int tj = 0; tj < MmaB::TN; ++tj) {
        const int n = n0 + mma.col_of(tj);
        const float v = mma.acc[ti][tj][r];
        if (n < in) {
          if (a.dx) atomicAdd(a.dx + xo + n, v);  // the two directions meet here: two addends, order-free
        } else if (n >= in_i && n < NN && t > 0) {
          a.dh_rec[((size_t)dir * a.B + i) * d + (n - in_i)] = v;
        }
      }
    }
  }
}

void launch_step_bwd_bf16(const StepBwdArgs& a, hipStream_t s) {
  allow_big_lds(lstm_step_bwd_bf16, TileCfg::LDS_BYTES);
  const dim3 grid(pad8((a.B + TileCfg::BM - 1) / TileCfg::BM), (a.in_i + a.d + TileCfg::BN - 1) / TileCfg::BN, 2);
  hipLaunchKernelGGL(lstm_step_bwd_bf16, grid, dim3(256), TileCfg::LDS_BYTES, s, a);
}

// ------------------------------------------------- fused backward step (bf16) --
// Step t of the backward recurrence in ONE launch: the k-loop computes dh_t(rec) = dz_{t+1} * wb_h^T for a
// 256-row x 128-unit tile (rows beyond step t+1's active prefix fall off the descriptor and read as 0),
// the epilogue adds the upstream d_out, runs the gate gradient and writes dz_t (packed, unit-major) and
// the running dc.  No dh round trip through HBM, no separate elementwise launch.
// grid (pad8(ceil(B/256)), d/128 (ceil), 2)
// WM < 8 / TM: block tiles of fewer rows (64 x 128 on ONE wave, 128 x 128 on two) for calls with few sequences -- the
// photo cell's 64 rows: a step is then a chain of K/32 k-tiles whose length is the DMA wave-instructions per k-tile
// (A rows + B rows, at ~40 clocks each whether or not the rows exist), 12 instead of 32.
#ifndef FVTA_BWD_EPI_ROWS
#define FVTA_BWD_EPI_ROWS 2  // rows of the epilogue whose loads are in flight together (4: 256 VGPRs + spills, no faster; 8: 2x slower)
#endif
#ifndef FVTA_BWD_RC_ONLY
#define FVTA_BWD_RC_ONLY 0   // 1: compile the "c(t) read back" path out (measurement)
#endif
// SK > 1 (one-wave block tiles only): SK waves per workgroup, each a complete one-wave pipeline of its own (own LDS
// stages, own DMA) over ONE SK-th of the k-tiles; the partial accumulators meet in LDS and are summed in wave order by
// wave 0, which alone runs the epilogue.  For calls of few sequences (the photo cell: 4 workgroups per launch) the launch
// is a chain of K/32 k-tiles at the latency of one; this cuts the chain by SK.
template <int WN, int TM, int WM = 8 / TM, int SK = 1>
__global__ __launch_bounds__((TileCfgT<WN, TM, WM>::NT * SK), (WN == 1 && SK == 1 ? 2 : 1)) void lstm_bwd_fused_bf16(FusedBwdArgs a) {
  typedef TileCfgT<WN, TM, WM> TileCfg;
  typedef MmaBT<WN, TM, WM> MmaB;
  static_assert(SK == 1 || TileCfg::NT == 64, "split-K: one-wave block tiles only");
  constexpr int NTHREADS = TileCfg::NT * SK;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_all[];
  const int skw = SK > 1 ? (int)(threadIdx.x >> 6) : 0;                              // this wave's k-slice / LDS region
  bf16_t* smem_h = smem_all + (size_t)skw * TileCfg::STAGES * TileCfg::STAGE_ELEMS;
  int64_t* s_oo = reinterpret_cast<int64_t*>(smem_all + (size_t)SK * TileCfg::STAGES * TileCfg::STAGE_ELEMS);
  const int tid = SK > 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x, dir = blockIdx.z + a.dir0;
  const int m0 = blockIdx.x * TileCfg::BM, u0 = blockIdx.y * TileCfg::BN;
  const int t = a.t, d = a.d, K = 4 * d;
  // ---- dx tiles riding on the step launch (a.dx_tiles > 0): the column tiles from d/BN on compute
  // dx_{t+1} = dz_{t+1} * wb_x^T -- the SAME A operand the step's dh tiles stream, so dz is not fetched from HBM a
  // second time by a separate pass over all steps (4.8 GB per call), and the tiles run on the CUs the single-round
  // step leaves idle.  The launch with t = -1 holds only the dx tiles of step 0.
  if ((int)blockIdx.y >= a.dh_tiles) {
    const int t1 = t + 1;
    if (t1 >= a.J) return;
    const int nn = a.plan.nactive[t1];
    if (m0 >= nn) return;
    const int n0 = ((int)blockIdx.y - a.dh_tiles) * TileCfg::BN, in = a.in;
    const size_t trow1 = ((size_t)dir * a.J + t1) * a.B;
    MmaB mma;
    mma.init(tid);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + trow1 * (size_t)K, (unsigned)nn * K * 2);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.Wb[dir], (unsigned)in * K * 2);  // the x rows of wb
    RowSrc<TileCfg::A_GLDS> az;
    RowSrc<TileCfg::B_GLDS> bw;
    az.setup(mma.wave_all, mma.lane, m0, nn, K * 2);
    bw.setup(mma.wave_all, mma.lane, n0, in, K * 2);
    auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
      az.issue(rz, As, mma.wave_all, tile * 64);
      bw.issue(rw, Bs, mma.wave_all, tile * 64);
    };
    glds_mainloop<false>(mma, issue, K / 32, smem_h, nullptr, a.sp ? 4 : 0);
#pragma unroll
    for (int ti = 0; ti < MmaB::TM; ++ti) {
      int64_t xos[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) xos[r] = a.plan.xo[trow1 + min(m0 + mma.row_of(ti, r), nn - 1)];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = m0 + mma.row_of(ti, r);
        if (i >= nn) continue;
#pragma unroll
        for (int tj = 0; tj < MmaB::TN; ++tj) {
          const int n = n0 + mma.col_of(tj);
          if (n < in) atomicAdd(a.dx + xos[r] + n, mma.acc[ti][tj][r]);  // fw and bw meet at a position: two addends
        }
      }
    }
    return;
  }
  if (t < 0) return;  // the extra launch carries dx tiles only
  const int nact = a.plan.nactive[t];
  if (m0 >= nact) return;
  const int nnext = (t + 1 < a.J) ? a.plan.nactive[t + 1] : 0;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  for (int r = (int)threadIdx.x; r < TileCfg::BM; r += NTHREADS) s_oo[r] = a.plan.oo[trow + min(m0 + r, nact - 1)];  // clamped: always a valid row
  MmaB mma;
  mma.init(tid);
  // FVTA_DEBUG_SKIP & 65536-style diagnostics: env FVTA_LSTM_STAMP_BWD=<workgroup> stamps step t = 5 (tools/lstm_phases.py)
  const int lin_wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  unsigned long long* st = (a.stamp_wg >= 0 && lin_wg == a.stamp_wg && threadIdx.x == 0 && t == 5) ? g_lstm_stamps : nullptr;
  if (m0 < nnext) {
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + (trow + a.B) * (size_t)K, (unsigned)nnext * K * 2);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.Wb[dir] + (size_t)a.in_i * K, (unsigned)d * K * 2);  // the h rows of wb
    RowSrc<TileCfg::A_GLDS> az;
    RowSrc<TileCfg::B_GLDS> bw;
#ifdef FVTA_BWD_FAKE_BLOCKED
    // timing experiment (results are garbage): the A operand addressed AS IF dz were stored k-tile-major
    // [K/32][rows][32] -- a k-tile's 256 row pieces are then one contiguous 16 KB instead of 64 B every 4 KB
    az.setup(mma.wave_all, mma.lane, m0, nnext, 64);
    bw.setup(mma.wave_all, mma.lane, u0, d, K * 2);
    auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
      az.issue(rz, As, mma.wave_all, (unsigned)tile * (unsigned)nnext * 64u);
      bw.issue(rw, Bs, mma.wave_all, tile * 64);
    };
#else
    az.setup(mma.wave_all, mma.lane, m0, nnext, K * 2);
    bw.setup(mma.wave_all, mma.lane, u0, d, K * 2);
    const int ktiles = K / 32 / SK, ktile0 = skw * ktiles;   // (K = 4d, d a multiple of 32: K / 32 is a multiple of 4)
    auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
      az.issue(rz, As, mma.wave_all, (ktile0 + tile) * 64);
      bw.issue(rw, Bs, mma.wave_all, (ktile0 + tile) * 64);
    };
#endif
    if (st) st[0] = __builtin_readcyclecounter();
#ifdef FVTA_BWD_FAKE_BLOCKED
    glds_mainloop<false>(mma, issue, K / 32, smem_h, st ? st + 8 : nullptr, a.sp ? 4 : 0);
#else
    glds_mainloop<false>(mma, issue, ktiles, smem_h, st ? st + 8 : nullptr, a.sp ? 4 : 0);
#endif
    if (st) st[1] = __builtin_readcyclecounter();
  }
  __syncthreads();
  if constexpr (SK > 1) {
    // the waves' partial sums: each into ITS OWN stage region (36 KB, the pipeline is drained), [register][lane]; wave 0
    // adds them in wave order -- one fixed order, whatever the waves' timing -- and carries on alone
    float* mine = reinterpret_cast<float*>(smem_h);
    if (skw > 0) {
#pragma unroll
      for (int i = 0; i < MmaB::TM; ++i)
#pragma unroll
        for (int j = 0; j < MmaB::TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) mine[((i * MmaB::TN + j) * 16 + r) * 64 + tid] = mma.acc[i][j][r];
    }
    __syncthreads();
    if (skw > 0) return;
#pragma unroll
    for (int i = 0; i < MmaB::TM; ++i)
#pragma unroll
      for (int j = 0; j < MmaB::TN; ++j) {
#pragma unroll
        for (int w2 = 1; w2 < SK; ++w2) {
          const float* other = reinterpret_cast<const float*>(smem_all + (size_t)w2 * TileCfg::STAGES * TileCfg::STAGE_ELEMS);
#pragma unroll
          for (int r = 0; r < 16; ++r) mma.acc[i][j][r] += other[((i * MmaB::TN + j) * 16 + r) * 64 + tid];
        }
        asm volatile("" ::: "memory");  // one accumulator tile at a time: hoisting all 384 reads spills
      }
  }
  // (An LDS-staged, row-contiguous version of this epilogue -- as in the forward step -- was measured and is NOT
  // faster here: the tile's epilogue moves 36 B per (row, unit), 484 MB per launch, and the kernel already runs at
  // ~4.2 TB/s; it is bound by HBM and by the latency of these loads, not by the number of VMEM instructions.)
  const float* __restrict__ cs_t = a.cs + trow * d;
  const float* __restrict__ cs_p = a.cs + (trow - a.B) * d;  // step t-1 (unused at t == 0)
  float* __restrict__ dcs = a.dc + (size_t)dir * a.B * d;
  // RB rows per batch: ALL their loads are issued before the first of them is used.  (Row by row, two per unrolled
  // pass, every pass waited for its own loads AND -- vmcnt retires in order, stores included -- for the previous pass's
  // stores: a chain of 16 memory round trips per workgroup, which is what kept a launch with few active rows as long as a
  // full one.)
  constexpr int RB = FVTA_BWD_EPI_ROWS;
  static_assert(16 % RB == 0, "row batches tile the 16 rows of an accumulator tile");
  // (one call per batch with compile-time tile / row numbers: as a loop the compiler declines to unroll it and indexes the
  //  accumulators through scratch)
  auto epi_batch = [&](auto ti_c, auto r0_c) {
    {
      constexpr int ti = decltype(ti_c)::value, r0 = decltype(r0_c)::value;
      bf16x4 gp[RB][MmaB::TN];
      float c[RB][MmaB::TN], cp[RB][MmaB::TN], dcv[RB][MmaB::TN], dout[RB][MmaB::TN];
      // unconditional loads of the four column tiles of the batch's rows, then math, then guarded stores
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int row = mma.row_of(ti, r0 + rr);
        const int ic = min(m0 + row, nact - 1);
        const int64_t oo = s_oo[min(row, nact - 1 - m0)];
#pragma unroll
        for (int tj = 0; tj < MmaB::TN; ++tj) {
          const int u = min(u0 + mma.col_of(tj), d - 1);
          if (a.ntl) {
            gp[rr][tj] = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(a.gatesb + (trow + ic) * (size_t)K + 4 * u));
            c[rr][tj] = FVTA_BWD_RC_ONLY ? 0.f : (a.rc ? 0.f : __builtin_nontemporal_load(cs_t + (size_t)ic * d + u));
            cp[rr][tj] = t > 0 ? __builtin_nontemporal_load(cs_p + (size_t)ic * d + u) : 0.f;
            dout[rr][tj] = __builtin_nontemporal_load(a.d_out + oo + u);
          } else {
            gp[rr][tj] = *reinterpret_cast<const bf16x4*>(a.gatesb + (trow + ic) * (size_t)K + 4 * u);
            c[rr][tj] = FVTA_BWD_RC_ONLY ? 0.f : (a.rc ? 0.f : cs_t[(size_t)ic * d + u]);
            cp[rr][tj] = t > 0 ? cs_p[(size_t)ic * d + u] : 0.f;
            dout[rr][tj] = a.d_out[oo + u];
          }
          dcv[rr][tj] = dcs[(size_t)ic * d + u];
        }
      }
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int i = m0 + mma.row_of(ti, r0 + rr);
#pragma unroll
        for (int tj = 0; tj < MmaB::TN; ++tj) {
          const int u = u0 + mma.col_of(tj);
          const float ig = bf2f((bf16_t)gp[rr][tj][0]), jg = bf2f((bf16_t)gp[rr][tj][1]), fg = bf2f((bf16_t)gp[rr][tj][2]),
                      og = bf2f((bf16_t)gp[rr][tj][3]);
          const float dh = dout[rr][tj] + mma.acc[ti][tj][r0 + rr];
          const float tc = fvta_tanh((FVTA_BWD_RC_ONLY || a.rc) ? cp[rr][tj] * fg + ig * jg : c[rr][tj]);
          const float dc = dcv[rr][tj] + dh * og * (1.f - tc * tc);
          bf16x4 pk;
          pk[0] = (short)f2bf(dc * jg * ig * (1.f - ig));
          pk[1] = (short)f2bf(dc * ig * (1.f - jg * jg));
          pk[2] = (short)f2bf(dc * cp[rr][tj] * fg * (1.f - fg));
          pk[3] = (short)f2bf(dh * tc * og * (1.f - og));
          if (i < nact && u < d) {
            *reinterpret_cast<bf16x4*>(a.dzb + (trow + i) * (size_t)K + 4 * u) = pk;
            dcs[(size_t)i * d + u] = dc * fg;
          }
        }
      }
    }
  };
  auto epi_tile = [&](auto ti_c) {
    epi_batch(ti_c, std::integral_constant<int, 0>{});
    if constexpr (RB < 16) epi_batch(ti_c, std::integral_constant<int, RB>{});
    if constexpr (RB < 8) {
      epi_batch(ti_c, std::integral_constant<int, 2 * RB>{});
      epi_batch(ti_c, std::integral_constant<int, 3 * RB>{});
    }
    if constexpr (RB < 4) {
      epi_batch(ti_c, std::integral_constant<int, 4 * RB>{});
      epi_batch(ti_c, std::integral_constant<int, 5 * RB>{});
      epi_batch(ti_c, std::integral_constant<int, 6 * RB>{});
      epi_batch(ti_c, std::integral_constant<int, 7 * RB>{});
    }
  };
  static_assert(RB == 2 || RB == 4 || RB == 8 || RB == 16, "epilogue row batch");
  static_assert(MmaB::TM <= 4, "epilogue tiles");
  epi_tile(std::integral_constant<int, 0>{});
  if constexpr (MmaB::TM > 1) epi_tile(std::integral_constant<int, 1>{});
  if constexpr (MmaB::TM > 2) epi_tile(std::integral_constant<int, 2>{});
  if constexpr (MmaB::TM > 3) epi_tile(std::integral_constant<int, 3>{});
  if (st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[2] = __builtin_readcyclecounter();
    st[3] = K / 32;
  }
}

// How many 256-wide dx column tiles the fused backward step should carry (0: dx stays a separate pass).
// FVTA_LSTM_DX_FUSED=0 switches it off (measurement).
int bwd_fused_dx_tiles(int in, int d) {
  static const int mode = [] {
    const char* e = getenv("FVTA_LSTM_DX_FUSED");
    return e ? atoi(e) : FVTA_LSTM_DX_FUSED_DEFAULT;
  }();
  static const bool narrow = [] {
    const char* e = getenv("FVTA_LSTM_BWD_NARROW_TILE");
    return e && e[0] == '1';
  }();
  if (!mode || d % 256 != 0 || narrow) return 0;
  return (in + 255) / 256;
}

void launch_bwd_fused_bf16(const FusedBwdArgs& a_, hipStream_t s) {
  FusedBwdArgs a = a_;
  a.sp = (glds_sp_mask() >> 1) & 1;
  static const bool narrow = [] {
    const char* e = getenv("FVTA_LSTM_BWD_NARROW_TILE");
    return e && e[0] == '1';
  }();
  // (a.dx_tiles: extra column tiles of the launch that compute dx_{t+1}; set by bwd_fused_dx_tiles())
  if (a.d % 256 == 0 && (tile128_mask() & 2)) {
    constexpr int LDS = TileCfgT<2, 4>::LDS_BYTES + 256 * 8;
    allow_big_lds(lstm_bwd_fused_bf16<2, 4>, LDS);
    a.dh_tiles = a.d / 256;
    const dim3 grid(pad8((a.B + 255) / 256), a.dh_tiles + a.dx_tiles, a.ndir);
    hipLaunchKernelGGL((lstm_bwd_fused_bf16<2, 4>), grid, dim3(256), LDS, s, a);
  } else if (a.B <= 128 && small_rows() && a.dx_tiles == 0) {  // few sequences (the photo cell): row tiles of 64 / 128
    a.dh_tiles = (a.d + 127) / 128;
    // FVTA_LSTM_SMALL_SK=4: the k-loop of a 64-row call split over four waves of the workgroup (see the kernel).  Off:
    // parity-green, but each wave's private three-stage pipeline makes the workgroup 144 KB of LDS -- it no longer fits
    // beside a text-cell workgroup (96 KB) and waits for a whole free CU: 200 us per launch instead of 132 beside the text
    // cell's recurrence, dense step 15.0 -> 15.7 ms, ragged 8.22 -> 8.17 (tools/r02_ah.sh, tools/r02_ai.sh)
    static const int small_sk = [] {
      const char* e = getenv("FVTA_LSTM_SMALL_SK");
      return e ? atoi(e) : FVTA_LSTM_SMALL_SK_DEFAULT;
    }();
    if (a.B <= 64 && small_sk == 4 && (a.d / 8) % 4 == 0) {
      constexpr int LDS = 4 * TileCfgT<1, 2, 1>::LDS_BYTES + 64 * 8;
      allow_big_lds(lstm_bwd_fused_bf16<1, 2, 1, 4>, LDS);
      hipLaunchKernelGGL((lstm_bwd_fused_bf16<1, 2, 1, 4>), dim3((a.B + 63) / 64, a.dh_tiles, a.ndir), dim3(256), LDS, s, a);
    } else if (a.B <= 64) {
      constexpr int LDS = TileCfgT<1, 2, 1>::LDS_BYTES + 64 * 8;
      allow_big_lds(lstm_bwd_fused_bf16<1, 2, 1>, LDS);
      hipLaunchKernelGGL((lstm_bwd_fused_bf16<1, 2, 1>), dim3((a.B + 63) / 64, a.dh_tiles, a.ndir), dim3(64), LDS, s, a);
    } else {
      constexpr int LDS = TileCfgT<1, 2, 2>::LDS_BYTES + 128 * 8;
      allow_big_lds(lstm_bwd_fused_bf16<1, 2, 2>, LDS);
      hipLaunchKernelGGL((lstm_bwd_fused_bf16<1, 2, 2>), dim3((a.B + 127) / 128, a.dh_tiles, a.ndir), dim3(128), LDS, s, a);
    }
  } else if (a.d % 256 == 0 && !narrow &&
             !(a.nact_hint >= 0 && a.dx_tiles == 0 && 8 * ((a.nact_hint + 255) / 256) <= 256)) {
    // 256 x 256 tile: dz (the A operand, K = 4d wide) is re-read d/256 instead of d/128 times.  NOT for a step whose
    // active rows make at most 256 of the 256 x 128 workgroups (host hint): a launch lasts as long as one workgroup's
    // chain of k-tiles and its epilogue, and the small workgroup alone on a CU is through both sooner (ragged batches:
    // DESIGN.md 4.3.1)
    constexpr int LDS = TileCfgT<2>::LDS_BYTES + 256 * 8;
    allow_big_lds(lstm_bwd_fused_bf16<2, 2>, LDS);
    a.dh_tiles = a.d / 256;
    const dim3 grid(pad8((a.B + 255) / 256), a.dh_tiles + a.dx_tiles, a.ndir);
    hipLaunchKernelGGL((lstm_bwd_fused_bf16<2, 2>), grid, dim3(512), LDS, s, a);
  } else {
    allow_big_lds(lstm_bwd_fused_bf16<1, 2>, FWD_LDS);
    a.dh_tiles = (a.d + 127) / 128;
    a.dx_tiles = 0;  // (bwd_fused_dx_tiles() never asks for them with the narrow tile)
    const dim3 grid(pad8((a.B + 255) / 256), a.dh_tiles, a.ndir);
    hipLaunchKernelGGL((lstm_bwd_fused_bf16<1, 2>), grid, dim3(256), FWD_LDS, s, a);
  }
}

// dx = dz * wb_x^T for the steps [t0, t0 + nt) of both directions.  grid (pad8(ceil(B/256)), ceil(in/128), 2*nt)
template <int WN, int TM>
__global__ __launch_bounds__((TileCfgT<WN, TM>::NT), (WN == 1 ? 2 : 1)) void lstm_dx_bf16(FusedBwdArgs a) {
  typedef TileCfgT<WN, TM> TileCfg;
  typedef MmaBT<WN, TM> MmaB;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int tid = threadIdx.x;
  const int t = a.t0 + blockIdx.z % a.nt, dir = a.dir0 + blockIdx.z / a.nt;
  const int m0 = blockIdx.x * TileCfg::BM, n0 = blockIdx.y * TileCfg::BN;
  const int nact = a.plan.nactive[t];
  if (m0 >= nact) return;
  const int d = a.d, K = 4 * d, in = a.in;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  MmaB mma;
  mma.init(tid);
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + trow * (size_t)K, (unsigned)nact * K * 2);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.Wb[dir], (unsigned)in * K * 2);  // the x rows of wb
  RowSrc<TileCfg::A_GLDS> az;
  RowSrc<TileCfg::B_GLDS> bw;
  az.setup(mma.wave_all, mma.lane, m0, nact, K * 2);
  bw.setup(mma.wave_all, mma.lane, n0, in, K * 2);
  auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
    az.issue(rz, As, mma.wave_all, tile * 64);
    bw.issue(rw, Bs, mma.wave_all, tile * 64);
  };
  glds_mainloop<false>(mma, issue, K / 32, smem_h, nullptr, a.sp ? 4 : 0);
#pragma unroll
  for (int ti = 0; ti < MmaB::TM; ++ti) {
    int64_t xos[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) xos[r] = a.plan.xo[trow + min(m0 + mma.row_of(ti, r), nact - 1)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = m0 + mma.row_of(ti, r);
      if (i >= nact) continue;
#pragma unroll
      for (int tj = 0; tj < MmaB::TN; ++tj) {
        const int n = n0 + mma.col_of(tj);
        if (n < in) {
          // fw and bw meet at a position: two addends.  One launch: atomics.  One launch per direction: within a direction
          // every (sequence, position) is written once, so the first stores and the second adds, without atomics.
          float* o = a.dx + xos[r] + n;
          if (a.dxmode == 0)
            atomicAdd(o, mma.acc[ti][tj][r]);
          else if (a.dxmode == 1)
            *o = mma.acc[ti][tj][r];
          else
            *o += mma.acc[ti][tj][r];
        }
      }
    }
  }
}

void launch_dx_bf16(const FusedBwdArgs& a_, hipStream_t s) {
  FusedBwdArgs a = a_;
  a.sp = (glds_sp_mask() >> 2) & 1;
  static const bool wide = [] {  // (one 256-wide column tile instead of two 128-wide ones: measured slower at in = 200)
    const char* e = getenv("FVTA_LSTM_DX_WIDE_TILE");
    return e && e[0] == '1';
  }();
  static const int two_pass = [] {  // FVTA_LSTM_DX_2PASS: one launch per direction (store, then add) instead of atomics
    const char* e = getenv("FVTA_LSTM_DX_2PASS");
    return e ? atoi(e) : FVTA_LSTM_DX_2PASS_DEFAULT;
  }();
  if (two_pass && a.dxmode == 0 && a.ndir == 2 && a.dir0 == 0) {
    FusedBwdArgs p = a_;
    p.dir0 = 0; p.ndir = 1; p.dxmode = 1;
    launch_dx_bf16(p, s);
    p.dir0 = 1; p.dxmode = 2;
    launch_dx_bf16(p, s);
    return;
  }
  const int zdirs = a.dxmode == 0 ? 2 : 1;
  if (a.in > 128 && (tile128_mask() & 4)) {
    constexpr int LDS = TileCfgT<2, 4>::LDS_BYTES;
    allow_big_lds(lstm_dx_bf16<2, 4>, LDS);
    const dim3 grid(pad8((a.B + 255) / 256), (a.in + 255) / 256, zdirs * a.nt);
    hipLaunchKernelGGL((lstm_dx_bf16<2, 4>), grid, dim3(256), LDS, s, a);
  } else if (a.in > 128 && wide) {
    allow_big_lds(lstm_dx_bf16<2, 2>, TileCfgT<2>::LDS_BYTES);
    const dim3 grid(pad8((a.B + 255) / 256), (a.in + 255) / 256, zdirs * a.nt);
    hipLaunchKernelGGL((lstm_dx_bf16<2, 2>), grid, dim3(512), TileCfgT<2>::LDS_BYTES, s, a);
  } else {
    allow_big_lds(lstm_dx_bf16<1, 2>, TileCfgT<1>::LDS_BYTES);
    const dim3 grid(pad8((a.B + 255) / 256), (a.in + 127) / 128, zdirs * a.nt);
    hipLaunchKernelGGL((lstm_dx_bf16<1, 2>), grid, dim3(256), TileCfgT<1>::LDS_BYTES, s, a);
  }
}

// -------------------------------------------------------- weight gradient --
// slab(dir, split) [in_i+d][4d] = sum over the split's steps of [xs_t | hs_{t-1}]^T * dz_t.  Both operands
// are k-major in memory: staged as they lie, read through the transposing LDS read.
// grid (xtiles + htiles, 4d/128, 2*nsplit): m-tiles never mix x and h columns.
// WN = 2: 256 x 256 output tile, 8 waves -- the operands are re-read 8 + 3 instead of 16 + 3 times (this kernel runs
// at the rate the address unit feeds the LDS).
template <int WN, int TM>
__global__ __launch_bounds__((TileCfgT<WN, TM>::NT), (WN == 1 ? 2 : 1)) void lstm_dw_bf16(DwArgs a) {
  typedef TileCfgT<WN, TM> TileCfg;
  typedef MmaBT<WN, TM> MmaB;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int tid = threadIdx.x;
  const int d = a.d, in_i = a.in_i, N4 = 4 * d;
  const int xtiles = (in_i + TileCfg::BM - 1) / TileCfg::BM;
  // XCD-aware decode of the 1-D grid: workgroups are dealt round-robin over the 8 XCDs, and every tile of one
  // (direction, step group) slice streams the SAME rows -- so a slice's tiles all go to ONE XCD, where the re-reads
  // hit its L2 instead of crossing the fabric 8 times (measured: 18 GB of fabric reads per call before, 4.4 GB unique).
  const int mtiles = xtiles + (d + TileCfg::BM - 1) / TileCfg::BM, ntiles = N4 / TileCfg::BN, per = mtiles * ntiles;
  int bzl, tile_id;  // slice (direction, group) within this launch, tile within the slice
  if (a.xcd_aware) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    bzl = xcd + 8 * (slot / per);
    tile_id = slot % per;
  } else {
    bzl = blockIdx.x / per;
    tile_id = blockIdx.x % per;
  }
  if (bzl >= 2 * a.nsl) return;
  const int bz = (bzl / a.nsl) * a.nsplit + a.split0 + bzl % a.nsl;  // slab index: dir * nsplit + split
  const int bx = tile_id % mtiles, by = tile_id / mtiles;
  const bool isx = bx < xtiles;
  const int col0 = isx ? bx * TileCfg::BM : (bx - xtiles) * TileCfg::BM;  // within x / h columns
  const int ncols = isx ? in_i : d;
  const int n0 = by * TileCfg::BN;
  const int split = bz % a.nsplit, dir = bz / a.nsplit;
  MmaB mma;
  mma.init(tid);
  KMajorSrc<TileCfg::BM, TileCfg::A_GLDS> sa;
  KMajorSrc<TileCfg::BN, TileCfg::B_GLDS> sb;
  sa.setup(mma.wave_all, mma.lane, col0, ncols, (unsigned)ncols * 2);
  sb.setup(mma.wave_all, mma.lane, n0, N4, (unsigned)N4 * 2);
  const int t_begin = split * a.tgroup, t_end = min(a.J, t_begin + a.tgroup);
  for (int t = t_begin; t < t_end; ++t) {
    const int nact = a.plan.nactive[t];
    if (nact == 0) break;
    if (!isx && t == 0) continue;  // h_{-1} = 0
    const size_t trow = ((size_t)dir * a.J + t) * a.B;
    const bf16_t* A = isx ? a.xs + trow * in_i : a.hs + (trow - a.B) * d;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (unsigned)nact * ncols * 2);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + trow * (size_t)N4, (unsigned)nact * N4 * 2);
    auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
      sa.issue(ra, As, mma.wave_all, (unsigned)tile * 32u * ncols * 2u);
      sb.issue(rz, Bs, mma.wave_all, (unsigned)tile * 32u * N4 * 2u);
    };
    glds_mainloop<true>(mma, issue, (nact + 31) / 32, smem_h, nullptr, a.sp ? 4 : 0);
    __builtin_amdgcn_s_barrier();  // every wave is done with the ring before the next step refills it
  }
  float* slab = a.slabs + (size_t)bz * (in_i + d) * N4;
  const int mrow0 = isx ? col0 : in_i + col0;
#pragma unroll
  for (int ti = 0; ti < MmaB::TM; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = mma.row_of(ti, r);
      if (col0 + ml >= ncols) continue;
#pragma unroll
      for (int tj = 0; tj < MmaB::TN; ++tj) slab[(size_t)(mrow0 + ml) * N4 + n0 + mma.col_of(tj)] = mma.acc[ti][tj][r];
    }
}

void launch_dw_bf16(const DwArgs& a_, hipStream_t s) {
  DwArgs a = a_;
  a.sp = (glds_sp_mask() >> 3) & 1;
  const int xtiles = (a.in_i + 255) / 256, htiles = (a.d + 255) / 256;
  static const bool narrow = [] {
    const char* e = getenv("FVTA_LSTM_DW_NARROW_TILE");
    return e && e[0] == '1';
  }();
  if ((4 * a.d) % 256 == 0 && (tile128_mask() & 8)) {
    constexpr int LDS = TileCfgT<2, 4>::LDS_BYTES;
    allow_big_lds(lstm_dw_bf16<2, 4>, LDS);
    const int per = (xtiles + htiles) * (4 * a.d / 256);
    const dim3 grid(a.xcd_aware ? 8 * per * ((2 * a.nsl + 7) / 8) : per * 2 * a.nsl);
    hipLaunchKernelGGL((lstm_dw_bf16<2, 4>), grid, dim3(256), LDS, s, a);
  } else if ((4 * a.d) % 256 == 0 && !narrow) {
    allow_big_lds(lstm_dw_bf16<2, 2>, TileCfgT<2>::LDS_BYTES);
    const int per = (xtiles + htiles) * (4 * a.d / 256);
    const dim3 grid(a.xcd_aware ? 8 * per * ((2 * a.nsl + 7) / 8) : per * 2 * a.nsl);
    hipLaunchKernelGGL((lstm_dw_bf16<2, 2>), grid, dim3(512), TileCfgT<2>::LDS_BYTES, s, a);
  } else {
    allow_big_lds(lstm_dw_bf16<1, 2>, TileCfgT<1>::LDS_BYTES);
    const int per = (xtiles + htiles) * (4 * a.d / 128);
    const dim3 grid(a.xcd_aware ? 8 * per * ((2 * a.nsl + 7) / 8) : per * 2 * a.nsl);
    hipLaunchKernelGGL((lstm_dw_bf16<1, 2>), grid, dim3(256), TileCfgT<1>::LDS_BYTES, s, a);
  }
}

// slabs (internal row order) -> dkernel [in+d][4d] and dbias [4d], accumulated, fixed summation order
__global__ void lstm_dw_reduce_bf16(const float* __restrict__ slabs, int nslab, int in, int in_i, int d,
                                    float* __restrict__ dW, float* __restrict__ dbias) {
  const int N4 = 4 * d;
  const size_t slab_elems = (size_t)(in_i + d) * N4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= slab_elems) return;
  const int row = (int)(idx / N4), np = (int)(idx % N4);
  const int n = (np & 3) * d + (np >> 2);  // slab columns follow dz's unit-major order 4u+g
  if (row > in && row < in_i) return;     // zero pad rows
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += slabs[(size_t)k * slab_elems + idx];
  if (row < in)
    dW[(size_t)row * N4 + n] += s;
  else if (row == in)
    dbias[n] += s;  // the ones column
  else
    dW[(size_t)(in + row - in_i) * N4 + n] += s;
}

void launch_dw_reduce_bf16(const float* slabs, int nslab, int in, int in_i, int d, float* dW, float* dbias,
                           hipStream_t s) {
  const size_t n = (size_t)(in_i + d) * 4 * d;
  hipLaunchKernelGGL(lstm_dw_reduce_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slabs, nslab, in, in_i, d, dW,
                     dbias);
}

// ------------------------------------------------------------------ test gemm
// layout 1: C = A[M,K] * B[N,K]^T (row images);  layout 2: C = A[K,M]^T * B[K,N] (k-major images, tr reads)
__global__ void cvt_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = f2bf(src[i]);
}

template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void test_gemm_bf16_kernel(int M, int N, int K, const bf16_t* __restrict__ A,
                                                                const bf16_t* __restrict__ B, float* __restrict__ C, int sp) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * TileCfg::BM, n0 = blockIdx.y * TileCfg::BN;
  MmaB mma;
  mma.init(tid);
  if (LAYOUT == 1) {
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (unsigned)M * K * 2), rb = make_rsrc(B, (unsigned)N * K * 2);
    RowSrc<TileCfg::A_GLDS> sa;
    RowSrc<TileCfg::B_GLDS> sb;
    sa.setup(mma.wave, mma.lane, m0, M, K * 2);
    sb.setup(mma.wave, mma.lane, n0, N, K * 2);
    auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
      sa.issue(ra, As, mma.wave, tile * 64);
      sb.issue(rb, Bs, mma.wave, tile * 64);
    };
    glds_mainloop<false>(mma, issue, K / 32, smem_h, nullptr, sp ? 4 : 0);
  } else {
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (unsigned)K * M * 2), rb = make_rsrc(B, (unsigned)K * N * 2);
    KMajorSrc<TileCfg::BM, TileCfg::A_GLDS> sa;
    KMajorSrc<TileCfg::BN, TileCfg::B_GLDS> sb;
    sa.setup(mma.wave, mma.lane, m0, M, (unsigned)M * 2);
    sb.setup(mma.wave, mma.lane, n0, N, (unsigned)N * 2);
    auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
      sa.issue(ra, As, mma.wave, (unsigned)tile * 32u * M * 2u);
      sb.issue(rb, Bs, mma.wave, (unsigned)tile * 32u * N * 2u);
    };
    glds_mainloop<true>(mma, issue, (K + 31) / 32, smem_h, nullptr, sp ? 4 : 0);
  }
#pragma unroll
  for (int ti = 0; ti < MmaB::TM; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mma.row_of(ti, r);
#pragma unroll
      for (int tj = 0; tj < MmaB::TN; ++tj) {
        const int n = n0 + mma.col_of(tj);
        if (m < M && n < N) C[(size_t)m * N + n] = mma.acc[ti][tj][r];
      }
    }
}

// Test hook only: rounds A and B to bf16 in stream-ordered temporaries (the one place the library allocates).
// Row images need K % 32 == 0; k-major images take any K (rows past the end fall off the descriptor).
int test_gemm_bf16(int layout, int M, int N, int K, const float* A, const float* B, float* C, hipStream_t s) {
  if (layout != 1 && layout != 2) return FVTA_ERR_UNSUPPORTED;
  if (layout == 1 && K % 32) return FVTA_ERR_UNSUPPORTED;
  const size_t na = (size_t)M * K, nb = (size_t)N * K;
  bf16_t *Ab = nullptr, *Bb = nullptr;
  if (hipMallocAsync((void**)&Ab, na * 2, s) != hipSuccess || hipMallocAsync((void**)&Bb, nb * 2, s) != hipSuccess)
    return FVTA_ERR_LAUNCH;
  hipLaunchKernelGGL(cvt_f32_bf16_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, A, Ab, na);
  hipLaunchKernelGGL(cvt_f32_bf16_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, B, Bb, nb);
  const dim3 grid((M + TileCfg::BM - 1) / TileCfg::BM, (N + TileCfg::BN - 1) / TileCfg::BN);
  if (layout == 1) {
    allow_big_lds(test_gemm_bf16_kernel<1>, TileCfg::LDS_BYTES);
    hipLaunchKernelGGL(test_gemm_bf16_kernel<1>, grid, dim3(256), TileCfg::LDS_BYTES, s, M, N, K, Ab, Bb, C, glds_sp_mask() & 1);
  } else {
    allow_big_lds(test_gemm_bf16_kernel<2>, TileCfg::LDS_BYTES);
    hipLaunchKernelGGL(test_gemm_bf16_kernel<2>, grid, dim3(256), TileCfg::LDS_BYTES, s, M, N, K, Ab, Bb, C, (glds_sp_mask() >> 3) & 1);
  }
  (void)hipFreeAsync(Ab, s);
  (void)hipFreeAsync(Bb, s);
  return FVTA_OK;
}

}  // namespace fvta
