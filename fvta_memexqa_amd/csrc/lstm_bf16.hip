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

namespace fvta {

static inline int pad8(int v) { return (v + 7) / 8 * 8; }

template <int B, int E, class F>
__device__ __forceinline__ void bf_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    bf_static_for<B + 1, E>(f);
  }
}

template <class K>
static void allow_big_lds(K kernel, int bytes) {  // > 64 KB of dynamic LDS must be opted into, per kernel symbol
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// ---- weight shadows in the internal row order [x rows | ones row | zero pad | h rows] -----------
// kernel [in+d][N4] fp32 -> wb [in_i+d][xm N4] bf16, wt [N4][xm (in_i+d)] bf16.  grid (N4/32, (in_i+d)/32), 256 threads
// xm = 2 (split engine): two terms per value in the il32 layout (gemm_bf16.h) -- element e of a row at (e / 32) * 64 + e % 32,
// its low term 32 further; a wt row is [x part: il32 of in_i][h part: il32 of d] (a k-tile is wholly x or wholly h).
__device__ __forceinline__ int il32(int e) { return ((e >> 5) << 6) + (e & 31); }
__global__ void cvt_weights_kernel(const float* __restrict__ W, bf16_t* __restrict__ wt, bf16_t* __restrict__ wb,
                                   int in, int in_i, int d, int xm) {
  __shared__ float tile[32][33];
  const int N4 = 4 * d, Ki = in_i + d;
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    const int src = k < in ? k : (k >= in_i ? in + (k - in_i) : -1);  // internal row -> kernel row
    const float v = src >= 0 ? W[(size_t)src * N4 + n] : 0.f;
    tile[r][tx] = v;
    const int c = 4 * (n % d) + n / d;  // column g*d+u -> 4u+g: dz rows are unit-major
    const bf16_t hi = f2bf(v);
    bf16_t* row = wb + (size_t)k * N4 * xm;
    if (xm == 1) {
      row[c] = hi;
    } else {
      row[il32(c)] = hi;
      row[il32(c) + 32] = f2bf(v - bf2f(hi));
    }
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + tx;
    const float v = tile[tx][r];
    const bf16_t hi = f2bf(v);
    bf16_t* row = wt + (size_t)(n0 + r) * Ki * xm;
    if (xm == 1) {
      row[k] = hi;
    } else {
      const int base = k < in_i ? 0 : 2 * in_i, kk = k < in_i ? k : k - in_i;
      row[base + il32(kk)] = hi;
      row[base + il32(kk) + 32] = f2bf(v - bf2f(hi));
    }
  }
}

void launch_cvt_weights_bf16(const float* W, bf16_t* wt, bf16_t* wb, int in, int in_i, int d, int xm, hipStream_t s) {
  hipLaunchKernelGGL(cvt_weights_kernel, dim3(4 * d / 32, (in_i + d) / 32), dim3(256), 0, s, W, wt, wb, in, in_i, d, xm);
}

// ---- input shadow: xs[dir][t][i][:] for every active (dir, t, i).  grid (ceil(B/16), J), 256 threads ----------
// Sixteen lanes per (sorted row i, position pos), 16 bytes of the fp32 row per lane and pass: the row is read ONCE and
// written to both places it is needed -- step pos of the forward direction and step len - 1 - pos of the backward
// direction (reverse_sequence).
// xm = 2 (split engine): the shadow row holds two terms per value in the il32 layout.
__global__ __launch_bounds__(256) void cvt_x_kernel(PlanView pv, const float* __restrict__ x, bf16_t* __restrict__ xs, int B,
                                                    int J, int in, int in_i, int xm) {
  const int pos = blockIdx.y;
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4), c16 = threadIdx.x & 15;
  if (i >= pv.nactive[pos]) return;  // len_i <= pos
  const int len = pv.len[pv.order[i]];
  const float* src = x + pv.xo[(size_t)pos * B + i];  // forward direction, step pos = position pos
  const int64_t bw_delta = pv.hdr->x_bw_delta;        // != 0: the backward direction has its own input (fvta_lstm_plan_xdir)
  bf16_t* dst_fw = xs + ((size_t)pos * B + i) * in_i * xm;
  bf16_t* dst_bw = xs + (((size_t)J + (len - 1 - pos)) * B + i) * in_i * xm;
  const bool al = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)(bw_delta * 4)) & 15) == 0;  // 16-byte aligned rows
  for (int c = c16 * 4; c < in_i; c += 64) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f}, vb;
    if (c + 4 <= in && al) {
      v = *reinterpret_cast<const f32x4*>(src + c);
      vb = bw_delta ? *reinterpret_cast<const f32x4*>(src + bw_delta + c) : v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = c + e;
        // ones columns at `in` (its weight-gradient row is dbias) and `in + 1` (the second bias term of lstm_wreg.hip;
        // in % 4 == 0, so in + 1 < in_i always)
        v[e] = k < in ? src[k] : ((k == in || k == in + 1) ? 1.0f : 0.f);
        vb[e] = (bw_delta && k < in) ? src[bw_delta + k] : v[e];
      }
    }
    bf16x4 o, ob, l, lb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (short)f2bf(v[e]);
      ob[e] = (short)f2bf(vb[e]);
      l[e] = (short)f2bf(v[e] - bf2f((bf16_t)o[e]));
      lb[e] = (short)f2bf(vb[e] - bf2f((bf16_t)ob[e]));
    }
    if (xm == 1) {
      *reinterpret_cast<bf16x4*>(dst_fw + c) = o;
      *reinterpret_cast<bf16x4*>(dst_bw + c) = ob;
    } else {  // (c is a multiple of 4: the four values share an il32 group)
      *reinterpret_cast<bf16x4*>(dst_fw + il32(c)) = o;
      *reinterpret_cast<bf16x4*>(dst_bw + il32(c)) = ob;
      *reinterpret_cast<bf16x4*>(dst_fw + il32(c) + 32) = l;
      *reinterpret_cast<bf16x4*>(dst_bw + il32(c) + 32) = lb;
    }
  }
}

void launch_cvt_x_bf16(const PlanView& pv, const float* x, bf16_t* xs, int B, int J, int in, int in_i, int xm, hipStream_t s) {
  hipLaunchKernelGGL(cvt_x_kernel, dim3((B + 15) / 16, J), dim3(256), 0, s, pv, x, xs, B, J, in, in_i, xm);
}

// ------------------------------------------------------------ forward step --
// The TILED forward step: z = [xs_t | hs_{t-1}] * wt^T over the 4 gate strips of 32 units, one 256 x 128 block tile per
// workgroup (rows [m0, m0 + 256) x the 32 units from ub), both operands through the LDS-DMA ring, the gate math as the
// epilogue.  It serves the shapes lstm_wreg.hip is not built for (launch_step_fwd_wreg returns false); at the metric
// shape the weights-in-registers kernel runs instead.
// grid (pad8(row tiles), d/32, 2), 256 threads, two workgroups per CU
// XM = 2 (split engine): 64-element stage rows (32 k of hi | lo), two stages of 48 KB, the plain loop
template <int XM>
__global__ __launch_bounds__(256, XM == 1 ? 2 : 1) void lstm_step_fwd_bf16(StepArgs a) {
  typedef TileCfgT<1, 2, 4, XM == 1 ? 3 : 2, XM == 1 ? 32 : 64> Cfg;
  constexpr int BK = Cfg::BK;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  int64_t* s_oo = reinterpret_cast<int64_t*>(smem_h + Cfg::STAGES * Cfg::STAGE_ELEMS);  // [256], same array
  const int dir = blockIdx.z, t = a.t;
  const int nact = a.plan.nactive[t];
  const int ub = blockIdx.y * 32;  // first unit of the block
  const int m0 = blockIdx.x * Cfg::BM;
  if (m0 >= nact) return;
  const int tid = threadIdx.x;
  const int d = a.d, in_i = a.Kp - a.d;
  const int in_k = in_i * a.xm, d_k = d * a.xm, Kk = a.Kp * a.xm;  // K extents of the x / h operand rows (split engine: two stored terms)
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  for (int r = tid; r < Cfg::BM; r += Cfg::NT) s_oo[r] = (m0 + r < nact) ? a.plan.oo[trow + m0 + r] : -1;

  MmaBT<1, 2, 4, Cfg::STAGES, BK, XM == 2> mma;
  mma.init(tid);
  const int u0 = ub;
  // A rows m0.. of xs[dir][t] (nact rows) and of hs[dir][t-1]; B rows = the 4 gate strips of wt
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.xs + trow * in_k, (unsigned)nact * in_k * 2);
  const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.hs + (t > 0 ? trow - a.B : trow) * d_k, (unsigned)nact * d_k * 2);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.Wt[dir], (unsigned)(4 * d) * Kk * 2);
  RowSrc<Cfg::A_GLDS, BK> ax, ah;
  RowSrc<Cfg::B_GLDS, BK> bw;
  ax.setup(mma.wave_all, mma.lane, m0, nact, in_k * 2);
  ah.setup(mma.wave_all, mma.lane, m0, nact, d_k * 2);
#pragma unroll
  for (int j = 0; j < Cfg::B_GLDS; ++j) {  // B row r = gate strip (r>>5)&3, unit ub + (r&31)
    constexpr int CPR = BK / 8;
    const int U = (mma.wave_all * Cfg::B_GLDS + j) * 64 + mma.lane;
    const int r = U / CPR, c = (U % CPR) ^ row_swz<BK>(r);
    bw.voff[j] = (unsigned)(((r >> 5) & 3) * d + ub + (r & 31)) * (unsigned)(Kk * 2) + 16u * c;
  }
  const int nx = in_k / BK, nt = (t == 0) ? nx : nx + d_k / BK;
  auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
    if (tile < nx)
      ax.issue(rx, As, mma.wave_all, tile * (BK * 2));
    else
      ah.issue(rh, As, mma.wave_all, (tile - nx) * (BK * 2));
    bw.issue(rw, Bs, mma.wave_all, tile * (BK * 2));
  };
  // c_{t-1} of the wave's rows, row-contiguous (16 B per lane), requested before the k-loop hides their latency
  f32x4 cprev[2][4];
  if (t > 0) {
    const float* src = a.cs ? a.cs + (trow - a.B) * (size_t)d : a.cstate + (size_t)dir * a.B * d;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int it = 0; it < 4; ++it) {  // rows it*8 + lane/8, units 4 (lane%8)..
        const int i = min(m0 + mma.wave * 64 + ti * 32 + it * 8 + (mma.lane >> 3), nact - 1);  // clamped: valid row
        cprev[ti][it] = *reinterpret_cast<const f32x4*>(src + (size_t)i * d + u0 + 4 * (mma.lane & 7));
      }
  }
  if constexpr (XM == 1)
    glds_mainloop_sp(mma, issue, nt, smem_h);
  else
    glds_mainloop<false>(mma, issue, nt, smem_h);
  __syncthreads();  // s_oo visible; every wave is done with the stage buffers, which become the epilogue's scratch
  lstm_gate_epilogue_staged(mma, a, dir, m0, u0, nact, trow, s_oo, cprev, reinterpret_cast<char*>(smem_h) + mma.wave_all * 9216, t);
}

void launch_step_fwd_bf16(const StepArgs& a, hipStream_t s) {
  const dim3 grid(pad8((a.B + 255) / 256), a.d / 32, 2);
  if (a.xm == 1) {
    constexpr int LDS = TileCfg::LDS_BYTES + 256 * 8;
    allow_big_lds(lstm_step_fwd_bf16<1>, LDS);
    hipLaunchKernelGGL(lstm_step_fwd_bf16<1>, grid, dim3(256), LDS, s, a);
  } else {
    constexpr int LDS = TileCfgT<1, 2, 4, 2, 64>::LDS_BYTES + 256 * 8;
    allow_big_lds(lstm_step_fwd_bf16<2>, LDS);
    hipLaunchKernelGGL(lstm_step_fwd_bf16<2>, grid, dim3(256), LDS, s, a);
  }
}



// ------------------------------------------------- fused backward step (bf16) --
// Step t of the backward recurrence in ONE launch: the k-loop computes dh_t(rec) = dz_{t+1} * wb_h^T for a block tile
// of rows x units (rows beyond step t+1's active prefix fall off the descriptor and read as 0), the epilogue adds the
// upstream d_out, runs the gate gradient and writes dz_t (packed, unit-major) and the running dc.  No dh round trip
// through HBM, no separate elementwise launch.  c_t is not read back: it is rebuilt as f c_{t-1} + i j from the saved
// bf16 gates (it only enters through tanh(c_t), next to gate values of the same precision): 32 B per (row, unit).
// grid (pad8(ceil(B/BM)), ceil(d/BN), 2)
// WM < 4: block tiles of fewer rows (64 x 128 on ONE wave, 128 x 128 on two) for calls with few sequences -- the photo
// cell's 64 rows: a step is then a chain of K/32 k-tiles whose length is the DMA wave-instructions per k-tile (A rows + B
// rows, at ~40 clocks each whether or not the rows exist), 12 instead of 32.
// BK = 64 (two stages): the wide tile's ring in whole-line DMA pieces (gemm_bf16.h)
// One step of one block tile.  (A sequence-stationary kernel -- one launch for all J steps, a workgroup owning 128 rows x all
// 512 units so that it depends on nobody else's dz, workgroups started out of phase so that k-loops and epilogues mix on
// the chip -- ran this same body in a t loop: 134.7 us per step against 129.0 for the per-step launches, staggered or not.)
#ifndef FVTA_BWD_EPD
#define FVTA_BWD_EPD 2
#endif
// first k-tile of a workgroup's rotated k-loop (lstm_dx_bf16): every workgroup would otherwise read the SAME 128-byte column
// of its 4-KB dz rows at the same moment -- the same few L2 channels.  Workgroups are dealt round-robin over the 8 XCDs (each
// with its own L2), so the workgroups of ONE XCD get consecutive rotations.  Only the order of a dot product's partial sums
// changes (fixed per tile: dx stays bitwise reproducible).
#ifndef FVTA_KT_ROT
#define FVTA_KT_ROT 1
#endif
__device__ __forceinline__ int kt_rot(int nkt) {
  const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  return FVTA_KT_ROT ? (int)((bid >> 3) * FVTA_KT_ROT % (unsigned)nkt) : 0;
}
// RPW (rows per 64-row wave tile, a multiple of 8): 64 = the plain 256-row block tile.  RPW < 64 (WM = 4 only): the block tile
// covers 4 RPW rows -- tile row r is global row m0 + (r / 64) RPW + r % 64, rows r % 64 >= RPW are padding (zero operand rows,
// their epilogue passes compiled out) -- so that the launch's grid fills the CUs: at the metric shape 58 tiles of 224 rows
// x 2 column tiles x 2 directions = 232 workgroups instead of 204 of 256 rows on 256 CUs, each with 7/8 of the epilogue.
template <int WN, int WM, int XM, int BK, int ST, int RPW = 64, int TN = 4, int EPD_ = FVTA_BWD_EPD>  // XM = 2: the split engine (two stored bf16 terms per operand value, three products; fp32 saved gates)
__device__ __forceinline__ void lstm_bwd_tile_step(const FusedBwdArgs& a, int t, int dir, int m0, int u0, bf16_t* smem_h) {
  static_assert(RPW == 64 || (WM == 4 && RPW % 8 == 0 && RPW > 32 && RPW < 64), "rows per wave tile");
  auto grow_of = [&](int r) { return RPW == 64 ? m0 + r : m0 + (r >> 6) * RPW + (r & 63); };  // tile row -> global sorted row
  typedef TileCfgT<WN, 2, WM, ST, BK, TN> TileCfg;
  typedef MmaBT<WN, 2, WM, ST, BK, XM == 2, TN> MmaB;
  int64_t* s_oo = reinterpret_cast<int64_t*>(smem_h + (size_t)TileCfg::STAGES * TileCfg::STAGE_ELEMS);
  const int tid = (int)threadIdx.x;
  const int d = a.d, N4 = 4 * d, K = N4 * XM;  // K: the dz row (split engine: two terms per value, il32 layout)
  const int nact = a.plan.nactive[t];
  if (m0 >= nact) return;
  const int nnext = (t + 1 < a.J) ? a.plan.nactive[t + 1] : 0;
  const size_t trow = ((size_t)dir * a.J + t) * a.B;
  for (int r = tid; r < TileCfg::BM; r += TileCfg::NT) s_oo[r] = a.plan.oo[trow + min(grow_of(r), nact - 1)];  // clamped: always a valid row
  // compile-time ablations (timing experiments, -DFVTA_TBWD_ABL=bits; results are garbage): 1 no k-loop, 2 no epilogue
  // stores, 4 no epilogue loads
#ifdef FVTA_TBWD_ABL
  constexpr int abl = FVTA_TBWD_ABL;
#else
  constexpr int abl = 0;
#endif
  MmaB mma;
  mma.init(tid);
  if (m0 < nnext && !(abl & 1)) {
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + (trow + a.B) * (size_t)K, (unsigned)nnext * K * 2);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.Wb[dir] + (size_t)a.in_i * K, (unsigned)d * K * 2);  // the h rows of wb
    RowSrc<TileCfg::A_GLDS, BK> az;
    RowSrc<TileCfg::B_GLDS, BK> bw;
    az.template setup<RPW>(mma.wave_all, mma.lane, m0, nnext, K * 2);
    bw.setup(mma.wave_all, mma.lane, u0, d, K * 2);
    // (a k-tile order rotated per workgroup, as in lstm_dx_bf16 and the pipelined kernel: 134.5 vs 132.3 us here)
    auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
      az.issue(rz, As, mma.wave_all, tile * (BK * 2));
      bw.issue(rw, Bs, mma.wave_all, tile * (BK * 2));
    };
    glds_mainloop<false>(mma, issue, K / BK, smem_h);
  }
  __syncthreads();
  // ---- gate gradient with every global access 16 bytes of a row: each 32-row x 32-unit plane of dh goes through a
  // wave-private LDS scratch (the operand ring, drained) and comes back row-contiguous -- lane = (row lane / 8 of a pass of
  // 8 rows, units 4 (lane % 8) ..), 256 vector-memory instructions per wave tile instead of the 1,536 four- and
  // eight-byte accesses the MFMA C layout would give (a wave stands ~124 cycles at every one, whatever its width:
  // tools/probes/store_issue_probe.hip).  Measured -1.7 %: the epilogue's 538 MB per launch sit on the HBM roof.
  constexpr int LDP = 36;  // floats per scratch row (32 + pad, keeps 16-byte alignment)
  float* pl_ = reinterpret_cast<float*>(smem_h) + (size_t)mma.wave_all * (32 * LDP);
  const int io_row = mma.lane >> 3, io_c4 = mma.lane & 7;
  const float* __restrict__ cs_p = a.cs + (trow - a.B) * d;  // step t-1 (unused at t == 0)
  float* __restrict__ dcs = a.dc + (size_t)dir * a.B * d;
  // (pinned in scalar registers: the compiler re-loaded the kernel argument in front of every pass's stores -- 32 scalar
  //  loads per wave tile, each behind an s_waitcnt lgkmcnt(0))
  // (as an INTEGER: an asm on the pointer itself makes it a generic pointer, and the stores through it FLAT instructions --
  //  which count on lgkmcnt too, so that every wave_sync of the epilogue waited for the dz stores)
  unsigned long long dz_base = (unsigned long long)reinterpret_cast<uintptr_t>(a.dzb + trow * (size_t)K);
  asm volatile("" : "+s"(dz_base));
  typedef f32x4 __attribute__((address_space(1)))* gf4_ptr;
  auto wave_sync = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  struct In {
    f32x4 g0, g1, g2, g3, cp, dout, dcv;  // g0, g1: packed bf16 gates of four units; split engine: g0..g3 fp32 gates, one unit each
  };
  // -DFVTA_TBWD_NT=bits (measurement): 1 the read-once streams (gates, c, d_out) with non-temporal loads (the default),
  // 2 dz stores non-temporal, 4 dc stores non-temporal, 8 dc loads non-temporal
#ifndef FVTA_TBWD_NT
#define FVTA_TBWD_NT 1
#endif
  constexpr int ntb = FVTA_TBWD_NT;
  auto ldnt = [](const float* p) {
    if constexpr ((ntb & 1) != 0) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));  // read once
    else return *reinterpret_cast<const f32x4*>(p);
  };
  auto st16 = [](float* p, const f32x4 v, bool nt) {
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
  };
  // The wave tile's TM x TN planes of four passes each form ONE sequence of passes; the loads of pass P + EPD are requested
  // when pass P is done, also across plane boundaries (the loads do not depend on the plane's scratch): EPD passes of
  // 5 KB are in flight per wave all through the epilogue.
  constexpr int EPD = EPD_, NPASS = MmaB::TM * MmaB::TN * 4;
  auto plane_u = [&](int pl) { return u0 + mma.wn * MmaB::WCOLS + (pl % MmaB::TN) * 32 + 4 * io_c4; };
  auto load_pass = [&](int P, In& in) {
    if constexpr ((abl & 4) != 0) {
      in.g0 = in.g1 = in.g2 = in.g3 = in.cp = in.dout = in.dcv = f32x4{0.5f, 0.25f, 0.125f, 0.75f};
      return;
    }
    const int pl = P >> 2, it = P & 3, ti = pl / MmaB::TN;
    const int u = plane_u(pl);
    const int lr = it * 8 + io_row, row = mma.wave * MmaB::WROWS + ti * 32 + lr;
    const int ic = min(grow_of(row), nact - 1);  // clamped: always a valid row
    const int64_t oo = s_oo[row];                 // (filled with the same clamp)
    const int uc = min(u, d - 4);            // (a plane past the last unit: loads a valid address, stores nothing)
    if constexpr (XM == 1) {
      const float* gp = reinterpret_cast<const float*>(a.gatesb + (trow + ic) * (size_t)N4 + 4 * uc);
      in.g0 = ldnt(gp);
      in.g1 = ldnt(gp + 4);
    } else {
      const float* gp = a.gates32 + (trow + ic) * (size_t)N4 + 4 * uc;
      in.g0 = ldnt(gp);
      in.g1 = ldnt(gp + 4);
      in.g2 = ldnt(gp + 8);
      in.g3 = ldnt(gp + 12);
    }
    // c_{t-1}: an UNCONDITIONAL load, scaled by 0 at step 0 (which reads its own, finite, c_0 slab).  Under
    // `t > 0 ? load : 0` the compiler branches around the load and waits vmcnt(0) inside the branch -- which drains every
    // load of the passes in flight and every store of the pass before: the pass pipeline ran one pass at a time
    in.cp = ldnt((t > 0 ? cs_p : a.cs + trow * d) + (size_t)ic * d + uc) * (t > 0 ? 1.f : 0.f);
    const float* dp = a.d_out + oo + uc;
    if ((reinterpret_cast<uintptr_t>(dp) & 15) == 0)
      in.dout = ldnt(dp);
    else
      in.dout = f32x4{dp[0], dp[1], dp[2], dp[3]};  // an output row that is not 16-byte aligned
    // dc of a row that was not active at step t + 1 is zero by definition (the engine does not zero the buffer): an
    // unconditional load + select
    const f32x4 dcl = (ntb & 8) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dcs + (size_t)ic * d + uc))
                                : *reinterpret_cast<const f32x4*>(dcs + (size_t)ic * d + uc);
    in.dcv = grow_of(row) < nnext ? dcl : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto do_pass = [&](int P, const In& in) {
    const int pl = P >> 2, it = P & 3, ti = pl / MmaB::TN;
    const int u = plane_u(pl);
    const int lr = it * 8 + io_row, i = grow_of(mma.wave * MmaB::WROWS + ti * 32 + lr);
    const f32x4 dh4 = *reinterpret_cast<const f32x4*>(&pl_[lr * LDP + 4 * io_c4]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 ga = __builtin_bit_cast(u32x4, in.g0), gb = __builtin_bit_cast(u32x4, in.g1);
    u32x4 za, zb, la, lb;  // dz of the four units, packed bf16 (i, j | f, o); la / lb: the low terms (split engine)
    f32x4 dco;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ig, jg, fg, og;
      if constexpr (XM == 1) {
        const unsigned w0 = e < 2 ? ga[2 * e] : gb[2 * (e - 2)], w1 = e < 2 ? ga[2 * e + 1] : gb[2 * (e - 2) + 1];
        ig = bf2f((bf16_t)(w0 & 0xffff)), jg = bf2f((bf16_t)(w0 >> 16)), fg = bf2f((bf16_t)(w1 & 0xffff)), og = bf2f((bf16_t)(w1 >> 16));
      } else {
        const f32x4 g4 = e == 0 ? in.g0 : (e == 1 ? in.g1 : (e == 2 ? in.g2 : in.g3));
        ig = g4[0], jg = g4[1], fg = g4[2], og = g4[3];
      }
      const float dh = in.dout[e] + dh4[e];
      const float tc = fvta_tanh(in.cp[e] * fg + ig * jg);
      const float dc = in.dcv[e] + dh * og * (1.f - tc * tc);
      const float dzi = dc * jg * ig * (1.f - ig), dzj = dc * ig * (1.f - jg * jg), dzf = dc * in.cp[e] * fg * (1.f - fg),
                  dzo = dh * tc * og * (1.f - og);
      const bf16_t hi_i = f2bf(dzi), hi_j = f2bf(dzj), hi_f = f2bf(dzf), hi_o = f2bf(dzo);
      const unsigned z0 = (unsigned)hi_i | ((unsigned)hi_j << 16), z1 = (unsigned)hi_f | ((unsigned)hi_o << 16);
      unsigned l0 = 0, l1 = 0;
      if constexpr (XM == 2) {
        l0 = (unsigned)f2bf(dzi - bf2f(hi_i)) | ((unsigned)f2bf(dzj - bf2f(hi_j)) << 16);
        l1 = (unsigned)f2bf(dzf - bf2f(hi_f)) | ((unsigned)f2bf(dzo - bf2f(hi_o)) << 16);
      }
      if (e < 2) {
        za[2 * e] = z0, za[2 * e + 1] = z1, la[2 * e] = l0, la[2 * e + 1] = l1;
      } else {
        zb[2 * (e - 2)] = z0, zb[2 * (e - 2) + 1] = z1, lb[2 * (e - 2)] = l0, lb[2 * (e - 2) + 1] = l1;
      }
      dco[e] = dc * fg;
    }
    if (i < nact && u < d && (!(abl & 2) || dco[0] == 1234.5f)) {
      // dz element 4u + g; split engine: il32 -- the lane's four units (16 elements, u a multiple of 4) share a group of 32:
      // hi at 2 (4u) - (4u) % 32, i.e. ((4u) / 32) * 64 + (4u) % 32, lo 32 elements (four 16-byte units) further
      const size_t e0 = XM == 1 ? (size_t)(4 * u) : (size_t)(((4 * u) >> 5) << 6) + ((4 * u) & 31);
      const gf4_ptr zp = (gf4_ptr)(dz_base + ((size_t)i * K + e0) * sizeof(bf16_t));   // (16-byte units below)
      if ((ntb & 2) != 0) {
        __builtin_nontemporal_store(__builtin_bit_cast(f32x4, za), zp);
        __builtin_nontemporal_store(__builtin_bit_cast(f32x4, zb), zp + 1);
      } else {
        zp[0] = __builtin_bit_cast(f32x4, za);
        zp[1] = __builtin_bit_cast(f32x4, zb);
      }
      if constexpr (XM == 2) {
        zp[4] = __builtin_bit_cast(f32x4, la);
        zp[5] = __builtin_bit_cast(f32x4, lb);
      }
      st16(dcs + (size_t)i * d + u, dco, (ntb & 4) != 0);
    }
  };
  In ins[EPD];
  // (RPW < 64: the passes over a wave tile's padding rows -- rows (pl / TN) 32 + 8 it >= RPW -- do not exist)
  auto pass_live = [](int P) constexpr { return ((P >> 2) / MmaB::TN) * 32 + (P & 3) * 8 < RPW; };
  bf_static_for<0, EPD>([&](auto P_c) { load_pass(decltype(P_c)::value, ins[decltype(P_c)::value]); });
  bf_static_for<0, NPASS>([&](auto P_c) {  // (compile-time indices: the pass buffers stay in registers)
    constexpr int P = decltype(P_c)::value;
    if constexpr ((P & 3) == 0) {  // a new plane: its 32 x 32 accumulators -> scratch, row-contiguous
      constexpr int pl = P >> 2, ti = pl / MmaB::TN, tj = pl % MmaB::TN;
      if constexpr (P != 0) wave_sync();  // the previous plane's reads are done
#pragma unroll
      for (int r = 0; r < 16; ++r) pl_[((r & 3) + 8 * (r >> 2) + 4 * mma.hf) * LDP + mma.l31] = mma.acc[ti][tj][r];
      wave_sync();
    }
    if constexpr (pass_live(P)) do_pass(P, ins[P % EPD]);
    if constexpr (P + EPD < NPASS && pass_live(P + EPD)) load_pass(P + EPD, ins[P % EPD]);
  });
}

template <int WN, int WM = 4, int XM = 1, int BK = 32, int ST = 3, int RPW = 64, int TN = 4>
__global__ __launch_bounds__((TileCfgT<WN, 2, WM>::NT), (WN == 1 && XM == 1 ? 2 : 1)) void lstm_bwd_fused_bf16(FusedBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  typedef TileCfgT<WN, 2, WM, ST, BK, TN> TileCfg;
  lstm_bwd_tile_step<WN, WM, XM, BK, ST, RPW, TN>(a, a.t, blockIdx.z, blockIdx.x * (RPW == 64 ? TileCfg::BM : 4 * RPW), blockIdx.y * TileCfg::BN, smem_h);
}

static int bwd_cus() {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  return cus;
}

template <int XM>
static void launch_bwd_fused_xm(const FusedBwdArgs& a, hipStream_t s) {
  // narrow tiles: 32-deep stages x 3; split engine: 64-element stage rows (32 k of hi | lo) x 2
  constexpr int NBK = XM == 1 ? 32 : 64, NST = XM == 1 ? 3 : 2;
  if (XM == 2 && a.B <= 64 && a.d % 128 == 0) {
    // the split engine's photo cell: the 64 x 128 tile on FOUR waves of 64 x 32 (TN = 1) -- one wave's chain of 64 k-tiles was
    // 24 DMA pieces + 48 MFMAs per tile in ONE instruction stream (~360 us per step under the text cell, 40 steps: longer than the
    // text cell's own backward); four waves share them
    constexpr int LDS = TileCfgT<4, 2, 1, NST, NBK, 1>::LDS_BYTES + 64 * 8;
    allow_big_lds(lstm_bwd_fused_bf16<4, 1, XM, NBK, NST, 64, 1>, LDS);
    hipLaunchKernelGGL((lstm_bwd_fused_bf16<4, 1, XM, NBK, NST, 64, 1>), dim3((a.B + 63) / 64, a.d / 128, 2), dim3(256), LDS, s, a);
  } else if (a.B <= 64) {  // few sequences (the photo cell): row tiles of 64 / 128
    constexpr int LDS = TileCfgT<1, 2, 1, NST, NBK>::LDS_BYTES + 64 * 8;
    allow_big_lds(lstm_bwd_fused_bf16<1, 1, XM, NBK, NST>, LDS);
    hipLaunchKernelGGL((lstm_bwd_fused_bf16<1, 1, XM, NBK, NST>), dim3((a.B + 63) / 64, (a.d + 127) / 128, 2), dim3(64), LDS, s, a);
  } else if (a.B <= 128) {
    constexpr int LDS = TileCfgT<1, 2, 2, NST, NBK>::LDS_BYTES + 128 * 8;
    allow_big_lds(lstm_bwd_fused_bf16<1, 2, XM, NBK, NST>, LDS);
    hipLaunchKernelGGL((lstm_bwd_fused_bf16<1, 2, XM, NBK, NST>), dim3((a.B + 127) / 128, (a.d + 127) / 128, 2), dim3(128), LDS, s, a);
  } else if (a.d % 256 == 0 && !(a.nact_hint >= 0 && 4 * ((a.nact_hint + 255) / 256) <= 160)) {
    // 256 x 256 tile: dz (the A operand, K = 4d wide) is re-read d/256 instead of d/128 times.  NOT for a step whose
    // active rows (the host's lengths, when it has them: fvta_bilstm_bwd_hint) fill at most 160 of the CUs with such
    // tiles: a launch lasts as long as one workgroup's chain of 64 k-tiles plus its epilogue, and the 256 x 128 tiles, two
    // workgroups per CU, spread the same rows over twice as many (ragged batches: 7.06 -> 6.74 ms per step; dense
    // batches keep the wide tile: 5.48 vs 5.72 ms per backward)
    constexpr int BK = 64, ST = 2;      // whole-line DMA pieces (gemm_bf16.h); 4d XM is a multiple of 64
    constexpr int LDS = TileCfgT<2, 2, 4, ST, BK>::LDS_BYTES + 256 * 8;
    // row tiles of 224 rows (RPW 56) when the 256-row tiles leave CUs idle that the 224-row tiles would use: one dispatch
    // round either way (-DFVTA_BWD_RPW=64, or FVTA_BWD_RPW=64 in the environment: off)
#ifndef FVTA_BWD_RPW
#define FVTA_BWD_RPW 56
#endif
    constexpr int RPW = FVTA_BWD_RPW;
    if constexpr (RPW != 64) {
      const int rows = a.nact_hint >= 0 ? a.nact_hint : a.B, percol = 2 * (a.d / 256), cus = bwd_cus();
      const int t256 = (rows + 255) / 256 * percol, tr = (rows + 4 * RPW - 1) / (4 * RPW) * percol;
      static const bool rpw_on = [] { const char* e = getenv("FVTA_BWD_RPW"); return !(e && e[0] == '6' && e[1] == '4'); }();  // A/B runs
      if (rpw_on && t256 <= cus && tr <= cus && tr > t256) {
        allow_big_lds(lstm_bwd_fused_bf16<2, 4, XM, BK, ST, RPW>, LDS);
        hipLaunchKernelGGL((lstm_bwd_fused_bf16<2, 4, XM, BK, ST, RPW>), dim3(pad8((a.B + 4 * RPW - 1) / (4 * RPW)), a.d / 256, 2), dim3(512), LDS, s, a);
        return;
      }
    }
    allow_big_lds(lstm_bwd_fused_bf16<2, 4, XM, BK, ST>, LDS);
    hipLaunchKernelGGL((lstm_bwd_fused_bf16<2, 4, XM, BK, ST>), dim3(pad8((a.B + 255) / 256), a.d / 256, 2), dim3(512), LDS, s, a);
  } else {
    constexpr int LDS = TileCfgT<1, 2, 4, NST, NBK>::LDS_BYTES + 256 * 8;
    allow_big_lds(lstm_bwd_fused_bf16<1, 4, XM, NBK, NST>, LDS);
    hipLaunchKernelGGL((lstm_bwd_fused_bf16<1, 4, XM, NBK, NST>), dim3(pad8((a.B + 255) / 256), (a.d + 127) / 128, 2), dim3(256), LDS, s, a);
  }
}

extern long long g_bwd_step_counts[3];  // lstm_wreg_bwd.hip
void launch_bwd_fused_bf16(const FusedBwdArgs& a, hipStream_t s) {
  if (launch_bwd_wreg(a, s)) return;  // few rows: the weights-stationary step
  if (a.B > 64) ++g_bwd_step_counts[0];  // (calls of more than 64 sequences: the photo cell's 64 rows are not what the roofline is quoted for)
  if (a.xm == 2)
    launch_bwd_fused_xm<2>(a, s);
  else
    launch_bwd_fused_xm<1>(a, s);
}

// dx = dz * wb_x^T.  grid (pad8(ceil(B/256)), ceil(in/BN), J)
// The two directions meet at every input position (the backward direction visits position len - 1 - t at step t).
// BOTH (one shared input, x_bw_delta = 0 -- no input dropout): ONE launch per position p sums both directions in the
// accumulators -- a k-loop over [dz_fw(step p) | dz_bw(step len_i - 1 - p)] x [wb_fw ; wb_bw], the backward direction's rows
// gathered through per-lane row addresses (sorted row i is row i of every step it is active in) -- and touches dx once
// (k-loop 0.39 ms + epilogue 0.17 ms per direction and launch before: one epilogue less).
// Otherwise one launch per direction: the forward direction's launch adds its product to dx, the backward direction's launch,
// behind it in stream order, adds to that with a plain load / store -- each element has exactly one contribution per
// direction, so no atomics (617 MB of float atomics at the memory side's ~1.3 TB/s were 0.4 ms of this kernel's 1.28) and a
// fixed summation order either way.
// WN = 2: one 256-wide column tile when the input fits it (dz, K = 4d wide, is read once), 64-deep stages.
// NOACC: the launcher's promise that accumulate == 0, at compile time -- as a run-time `if (accumulate) old = load(dst)` every
// row group's store sat behind an s_waitcnt vmcnt(0) that also waits for every EARLIER store of the wave: 56 store round trips
// per wave tile, one after the other (the same pattern cost the attention backward 17 % in round 4).
// WM / TN: the block tile's wave rows and a wave's column tiles (4 / 4; the split engine's photo cell: 1 / 1 with WN = 4 -- a
// 64 x 128 tile on four waves, as in its backward step)
template <int WN, int BK, int ST, bool BOTH = false, bool NOACC = false, int XM = 1, int WM = 4, int TN = 4>
__global__ __launch_bounds__((TileCfgT<WN, 2, WM>::NT), (WN == 1 && XM == 1 ? 2 : 1)) void lstm_dx_bf16(FusedBwdArgs a, int dir, int accumulate) {
  typedef TileCfgT<WN, 2, WM, ST, BK, TN> TileCfg;
  typedef MmaBT<WN, 2, WM, ST, BK, XM == 2, TN> MmaB;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int tid = threadIdx.x;
  const int t = blockIdx.z;
  const int m0 = blockIdx.x * TileCfg::BM, n0 = blockIdx.y * TileCfg::BN;
  const int nact = a.plan.nactive[t];
  if (m0 >= nact) return;
  // (the launcher cannot see the plan's header: with separate inputs per direction the BOTH launch serves the forward
  //  direction alone and the launch behind it the backward direction; with one input that second launch returns here)
  const bool shared_x = a.plan.hdr->x_bw_delta == 0;
  if (!BOTH && a.dx_both && dir == 1 && shared_x) return;
  const bool both = BOTH && shared_x;
  const int d = a.d, K = 4 * d * a.xm, in = a.in;
  const size_t trow = ((size_t)(BOTH ? 0 : dir) * a.J + t) * a.B;
  MmaB mma;
  mma.init(tid);
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + trow * (size_t)K, (unsigned)nact * K * 2);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.Wb[BOTH ? 0 : dir], (unsigned)in * K * 2);  // the x rows of wb
  RowSrc<TileCfg::A_GLDS, BK> az;
  RowSrc<TileCfg::B_GLDS, BK> bw;
  az.setup(mma.wave_all, mma.lane, m0, nact, K * 2);
  bw.setup(mma.wave_all, mma.lane, n0, in, K * 2);
  // BOTH: the backward direction's dz (all steps: one descriptor, < 4 GB) and its rows of this position
  const __amdgpu_buffer_rsrc_t rz1 = make_rsrc(a.dzb + (size_t)a.J * a.B * K, BOTH ? (unsigned)((size_t)a.J * a.B * K * 2) : 0u);
  const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(a.Wb[1], (unsigned)in * K * 2);
  RowSrc<TileCfg::A_GLDS, BK> az1;
  if (BOTH) {
    constexpr int CPR = BK / 8;
    // (two dependent gathers per DMA piece: every load unconditional on a clamped row and issued before the first use --
    //  under `if (i < nact)` each pair was its own two round trips, A_GLDS of them in a row at the head of every workgroup)
    int ord[TileCfg::A_GLDS], ln[TileCfg::A_GLDS];
#pragma unroll
    for (int j = 0; j < TileCfg::A_GLDS; ++j) {
      const int U = (mma.wave_all * TileCfg::A_GLDS + j) * 64 + mma.lane;
      ord[j] = a.plan.order[min(m0 + U / CPR, nact - 1)];
    }
#pragma unroll
    for (int j = 0; j < TileCfg::A_GLDS; ++j) ln[j] = a.plan.len[ord[j]];
#pragma unroll
    for (int j = 0; j < TileCfg::A_GLDS; ++j) {
      const int U = (mma.wave_all * TileCfg::A_GLDS + j) * 64 + mma.lane;
      const int row = U / CPR, c = (U % CPR) ^ row_swz<BK>(row);
      const int i = m0 + row;
      const int sb = ln[j] - 1 - t;  // the backward direction's step at this position (>= 0 where i is active)
      az1.voff[j] = i < nact ? (unsigned)(((size_t)sb * a.B + i) * (size_t)(K * 2)) + 16u * c : GLDS_OOB;
    }
  }
  const int nkt = K / BK, nkt_all = both ? 2 * nkt : nkt;
  const int rot = kt_rot(nkt_all);  // measured 1.11-1.16 vs 1.17-1.24 ms without
  auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
    const int kt = tile + rot - (tile + rot >= nkt_all ? nkt_all : 0);
    if (!BOTH || kt < nkt) {
      az.issue(rz, As, mma.wave_all, kt * (BK * 2));
      bw.issue(rw, Bs, mma.wave_all, kt * (BK * 2));
    } else {
      az1.issue(rz1, As, mma.wave_all, (kt - nkt) * (BK * 2));
      bw.issue(rw1, Bs, mma.wave_all, (kt - nkt) * (BK * 2));
    }
  };
  // the rows' input offsets (step t of this direction visits them), staged once: every lane of the epilogue needs its row's
  int64_t* s_xo = reinterpret_cast<int64_t*>(smem_h + (size_t)TileCfg::STAGES * TileCfg::STAGE_ELEMS);  // [BM]
  for (int r = tid; r < TileCfg::BM; r += TileCfg::NT) s_xo[r] = a.plan.xo[trow + min(m0 + r, nact - 1)];  // clamped: always a valid row
#ifdef FVTA_DX_ABL  // timing experiments (results are garbage): 1 no k-loop, 2 no epilogue loads, 4 no epilogue stores
  constexpr int abl = FVTA_DX_ABL;
#else
  constexpr int abl = 0;
#endif
  if (!(abl & 1)) glds_mainloop<false>(mma, issue, nkt_all, smem_h);
  __syncthreads();  // s_xo visible; every wave is done with the stage buffers, which become the epilogue's scratch
  // ---- epilogue with every global access 16 bytes of a row: each 32 x 32 plane goes through a wave-private LDS scratch and
  // comes back row-contiguous (lane = row lane / 8 of a pass of 8 rows, columns 4 (lane % 8) ..): a quarter of the
  // vector-memory instructions of the accumulator layout's 4-byte accesses, which is what bounds this epilogue
  constexpr int LDP = 36;
  float* pl = reinterpret_cast<float*>(smem_h) + (size_t)mma.wave_all * (32 * LDP);
  const int io_row = mma.lane >> 3, io_c4 = mma.lane & 7;
  auto wave_sync = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
#pragma unroll
  for (int ti = 0; ti < MmaB::TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < MmaB::TN; ++tj) {
      const int nb = n0 + mma.wn * MmaB::WCOLS + tj * 32;  // first column of the plane
      if (nb >= in) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) pl[((r & 3) + 8 * (r >> 2) + 4 * mma.hf) * LDP + mma.l31] = mma.acc[ti][tj][r];
      wave_sync();
      const int n = nb + 4 * io_c4;
      f32x4 old[4];
      float* ptr[4];
      bool ok[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = mma.wave * MmaB::WROWS + ti * 32 + it * 8 + io_row;
        ptr[it] = a.dx + s_xo[row] + n;
        ok[it] = m0 + row < nact && n < in;  // (in is a multiple of 4: a 4-column group is wholly inside or outside)
        old[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!NOACC) if (ok[it] && accumulate && !(abl & 2)) {
          if ((reinterpret_cast<uintptr_t>(ptr[it]) & 15) == 0)
            old[it] = *reinterpret_cast<const f32x4*>(ptr[it]);
          else
            old[it] = f32x4{ptr[it][0], ptr[it][1], ptr[it][2], ptr[it][3]};
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const f32x4 v = old[it] + *reinterpret_cast<const f32x4*>(&pl[(it * 8 + io_row) * LDP + 4 * io_c4]);
        if (ok[it] && (!(abl & 4) || v[0] == 1234.5f)) {
          if ((reinterpret_cast<uintptr_t>(ptr[it]) & 15) == 0) {
            *reinterpret_cast<f32x4*>(ptr[it]) = v;
          } else {
            ptr[it][0] = v[0]; ptr[it][1] = v[1]; ptr[it][2] = v[2]; ptr[it][3] = v[3];
          }
        }
      }
      wave_sync();  // the plane's reads are done before the next plane overwrites the scratch
    }
}

#ifndef FVTA_DX_WIDE
#define FVTA_DX_WIDE 1
#endif
#ifndef FVTA_DX_BOTH
#define FVTA_DX_BOTH 1
#endif
// the wide path with both directions in one launch: every element of a valid dx row is written exactly once
// (desc.dx_overwrite: no zero fill, no read of dx)
static bool dx_wide_both(const FusedBwdArgs& a) {
  return FVTA_DX_WIDE && FVTA_DX_BOTH && a.in > 128 && a.in <= 256 && (4 * a.d * a.xm) % 64 == 0 &&
         (size_t)a.J * a.B * (4 * a.d * a.xm) * 2 < ((size_t)1 << 32);
}
bool dx_writes_whole_rows(const FusedBwdArgs& a) { return dx_wide_both(a); }

template <int XM>
static void launch_dx_xm(const FusedBwdArgs& a, hipStream_t s) {
  if (FVTA_DX_WIDE && a.in > 128 && a.in <= 256 && (4 * a.d * a.xm) % 64 == 0) {
#ifndef FVTA_DX_BK
#define FVTA_DX_BK 64
#define FVTA_DX_ST 2
#endif
    static_assert(XM == 1 || FVTA_DX_BK == 64, "split engine: 64-element stage rows");
    typedef TileCfgT<2, 2, 4, FVTA_DX_ST, FVTA_DX_BK> Cfg;
    constexpr int LDS = Cfg::LDS_BYTES + 256 * 8;
    allow_big_lds(lstm_dx_bf16<2, FVTA_DX_BK, FVTA_DX_ST, false, false, XM>, LDS);
    const dim3 grid(pad8((a.B + 255) / 256), 1, a.J);
    // both directions in one launch when one descriptor covers a direction's dz (the kernel falls back by itself when the
    // directions have separate inputs)
    if (dx_wide_both(a)) {
      FusedBwdArgs b = a;
      b.dx_both = 1;
      if (!a.dx_accumulate) {
        allow_big_lds(lstm_dx_bf16<2, FVTA_DX_BK, FVTA_DX_ST, true, true, XM>, LDS);
        hipLaunchKernelGGL((lstm_dx_bf16<2, FVTA_DX_BK, FVTA_DX_ST, true, true, XM>), grid, dim3(512), LDS, s, b, 0, 0);
      } else {
        allow_big_lds(lstm_dx_bf16<2, FVTA_DX_BK, FVTA_DX_ST, true, false, XM>, LDS);
        hipLaunchKernelGGL((lstm_dx_bf16<2, FVTA_DX_BK, FVTA_DX_ST, true, false, XM>), grid, dim3(512), LDS, s, b, 0, a.dx_accumulate);
      }
      hipLaunchKernelGGL((lstm_dx_bf16<2, FVTA_DX_BK, FVTA_DX_ST, false, false, XM>), grid, dim3(512), LDS, s, b, 1, a.dx_accumulate);  // (separate inputs: its own dx copy)
      return;
    }
    for (int dir = 0; dir < 2; ++dir)
      hipLaunchKernelGGL((lstm_dx_bf16<2, FVTA_DX_BK, FVTA_DX_ST, false, false, XM>), grid, dim3(512), LDS, s, a, dir, a.dx_accumulate || dir);
    return;
  }
  constexpr int NBK = XM == 1 ? 32 : 64, NST = XM == 1 ? 3 : 2;
  if (XM == 2 && a.B <= 64 && a.in <= 128) {  // the split engine's photo cell: a 64 x 128 tile on four waves (no DMA of 192 rows that do not exist)
    constexpr int LDSP = TileCfgT<4, 2, 1, NST, NBK, 1>::LDS_BYTES + 64 * 8;
    allow_big_lds(lstm_dx_bf16<4, NBK, NST, false, false, XM, 1, 1>, LDSP);
    const dim3 gridp((a.B + 63) / 64, 1, a.J);
    for (int dir = 0; dir < 2; ++dir)
      hipLaunchKernelGGL((lstm_dx_bf16<4, NBK, NST, false, false, XM, 1, 1>), gridp, dim3(256), LDSP, s, a, dir, a.dx_accumulate || dir);
    return;
  }
  constexpr int LDS1 = TileCfgT<1, 2, 4, NST, NBK>::LDS_BYTES + 256 * 8;
  allow_big_lds(lstm_dx_bf16<1, NBK, NST, false, false, XM>, LDS1);
  const dim3 grid(pad8((a.B + 255) / 256), (a.in + 127) / 128, a.J);
  for (int dir = 0; dir < 2; ++dir)
    hipLaunchKernelGGL((lstm_dx_bf16<1, NBK, NST, false, false, XM>), grid, dim3(256), LDS1, s, a, dir, a.dx_accumulate || dir);
}

void launch_dx_bf16(const FusedBwdArgs& a, hipStream_t s) {
  if (a.xm == 2)
    launch_dx_xm<2>(a, s);
  else
    launch_dx_xm<1>(a, s);
}

// -------------------------------------------------------- weight gradient --
#ifndef FVTA_DW_STAGES
#define FVTA_DW_STAGES 4
#endif
constexpr int dw_stages(int wn) { return wn == 2 ? FVTA_DW_STAGES : 3; }  // ring depth (TileCfgT)

// slab(dir, split) [in_i+d][4d] = sum over the split's steps of [xs_t | hs_{t-1}]^T * dz_t.  Both operands
// are k-major in memory: staged as they lie, read through the transposing LDS read.
// 1-D grid over (slice = (direction, step group), m-tile, n-tile): m-tiles never mix x and h columns.
// WN = 2: 256 x 256 output tile, 8 waves -- the operands are re-read 8 + 3 instead of 16 + 3 times (this kernel runs
// at the rate the address unit feeds the LDS).
template <int WN>
__global__ __launch_bounds__((TileCfgT<WN>::NT), (WN == 1 ? 2 : 1)) void lstm_dw_bf16(DwArgs a) {
  typedef TileCfgT<WN, 2, 4, dw_stages(WN)> TileCfg;
  typedef MmaBT<WN, 2, 4, dw_stages(WN)> MmaB;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int tid = threadIdx.x;
  const int d = a.d, in_i = a.in_i, N4 = 4 * d;
  const int xtiles = (in_i + TileCfg::BM - 1) / TileCfg::BM;
  // XCD-aware decode of the 1-D grid: workgroups are dealt round-robin over the 8 XCDs, and every tile of one
  // (direction, step group) slice streams the SAME rows -- so a slice's tiles all go to ONE XCD, where the re-reads
  // hit its L2 instead of crossing the fabric 8 times (measured: 18 GB of fabric reads per call before, 4.4 GB unique).
  const int mtiles = xtiles + (d + TileCfg::BM - 1) / TileCfg::BM, ntiles = N4 / TileCfg::BN, per = mtiles * ntiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bz = xcd + 8 * (slot / per);  // slab index: dir * nsplit + split
  const int tile_id = slot % per;
  if (bz >= 2 * a.nsplit) return;
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
    const int nact = a.plan.nactive[t];  // k-rows = sequence rows
    if (nact == 0) break;
    if (!isx && t == 0) continue;  // h_{-1} = 0
    const size_t trow = ((size_t)dir * a.J + t) * a.B;
    const bf16_t* A = isx ? a.xs + trow * in_i : a.hs + (trow - (size_t)a.B) * d;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (unsigned)nact * ncols * 2);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + trow * (size_t)N4, (unsigned)nact * N4 * 2);
    auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
      sa.issue(ra, As, mma.wave_all, (unsigned)tile * 32u * ncols * 2u);
      sb.issue(rz, Bs, mma.wave_all, (unsigned)tile * 32u * N4 * 2u);
    };
    glds_mainloop<true>(mma, issue, (nact + 31) / 32, smem_h);
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

// The split engine's weight gradient (gemm_bf16.h: DwX2Cfg / MmaX2K): the same slices, tiles and slabs; both operands' rows are
// il32 two-term rows, so a 256-column logical tile is 512 physical columns and a stage holds 16 sequence rows.
__global__ __launch_bounds__(512, 1) void lstm_dw_x2(DwArgs a) {
  typedef DwX2Cfg TileCfg;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int tid = threadIdx.x;
  const int d = a.d, in_i = a.in_i, N4 = 4 * d;
  const int xtiles = (in_i + TileCfg::BM - 1) / TileCfg::BM;
  const int mtiles = xtiles + (d + TileCfg::BM - 1) / TileCfg::BM, ntiles = (N4 + TileCfg::BN - 1) / TileCfg::BN, per = mtiles * ntiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bz = xcd + 8 * (slot / per);  // slab index: dir * nsplit + split
  const int tile_id = slot % per;
  if (bz >= 2 * a.nsplit) return;
  const int bx = tile_id % mtiles, by = tile_id / mtiles;
  const bool isx = bx < xtiles;
  const int col0 = isx ? bx * TileCfg::BM : (bx - xtiles) * TileCfg::BM;  // within x / h columns (logical)
  const int ncols = isx ? in_i : d;
  const int n0 = by * TileCfg::BN;
  const int split = bz % a.nsplit, dir = bz / a.nsplit;
  MmaX2K mma;
  mma.init(tid);
  KMajorSrc<TileCfg::PCOLS, TileCfg::A_GLDS> sa;
  KMajorSrc<TileCfg::PCOLS, TileCfg::B_GLDS> sb;
  sa.setup(mma.wave_all, mma.lane, 2 * col0, 2 * ncols, (unsigned)ncols * 4);   // physical columns: two per logical one
  sb.setup(mma.wave_all, mma.lane, 2 * n0, 2 * N4, (unsigned)N4 * 4);
  const int t_begin = split * a.tgroup, t_end = min(a.J, t_begin + a.tgroup);
  for (int t = t_begin; t < t_end; ++t) {
    const int nact = a.plan.nactive[t];
    if (nact == 0) break;
    if (!isx && t == 0) continue;  // h_{-1} = 0
    const size_t trow = ((size_t)dir * a.J + t) * a.B;
    const bf16_t* A = isx ? a.xs + trow * (size_t)(2 * in_i) : a.hs + (trow - (size_t)a.B) * (size_t)(2 * d);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (unsigned)nact * ncols * 4);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dzb + trow * (size_t)(2 * N4), (unsigned)nact * N4 * 4);
    auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
      sa.issue(ra, As, mma.wave_all, (unsigned)tile * TileCfg::KR * ncols * 4u);
      sb.issue(rz, Bs, mma.wave_all, (unsigned)tile * TileCfg::KR * N4 * 4u);
    };
    glds_mainloop<true>(mma, issue, (nact + TileCfg::KR - 1) / TileCfg::KR, smem_h);
    __builtin_amdgcn_s_barrier();  // every wave is done with the ring before the next step refills it
  }
  float* slab = a.slabs + (size_t)bz * (in_i + d) * N4;
  const int mrow0 = isx ? col0 : in_i + col0;
#pragma unroll
  for (int ti = 0; ti < MmaX2K::TM; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = mma.row_of(ti, r);
      if (col0 + ml >= ncols) continue;
#pragma unroll
      for (int tj = 0; tj < MmaX2K::TN; ++tj)
        if (n0 + mma.col_of(tj) < N4) slab[(size_t)(mrow0 + ml) * N4 + n0 + mma.col_of(tj)] = mma.acc[ti][tj][r];
    }
}

void launch_dw_bf16(const DwArgs& a, hipStream_t s) {
  const int slices8 = (2 * a.nsplit + 7) / 8;  // slices per XCD
  if (a.xm == 2) {  // (4d is a multiple of 128; a 256-column tile past the end stores nothing)
    allow_big_lds(lstm_dw_x2, DwX2Cfg::LDS_BYTES);
    const int per = ((a.in_i + 255) / 256 + (a.d + 255) / 256) * ((4 * a.d + 255) / 256);
    hipLaunchKernelGGL(lstm_dw_x2, dim3(8 * per * slices8), dim3(512), DwX2Cfg::LDS_BYTES, s, a);
    return;
  }
  if ((4 * a.d) % 256 == 0) {
    constexpr int LDS2 = TileCfgT<2, 2, 4, dw_stages(2)>::LDS_BYTES;
    allow_big_lds(lstm_dw_bf16<2>, LDS2);
    const int per = ((a.in_i + 255) / 256 + (a.d + 255) / 256) * (4 * a.d / 256);
    hipLaunchKernelGGL((lstm_dw_bf16<2>), dim3(8 * per * slices8), dim3(512), LDS2, s, a);
  } else {
    allow_big_lds(lstm_dw_bf16<1>, TileCfgT<1>::LDS_BYTES);
    const int per = ((a.in_i + 255) / 256 + (a.d + 255) / 256) * (4 * a.d / 128);
    hipLaunchKernelGGL((lstm_dw_bf16<1>), dim3(8 * per * slices8), dim3(256), TileCfgT<1>::LDS_BYTES, s, a);
  }
}

// slabs (internal row order) -> dkernel [in+d][4d] and dbias [4d], accumulated, fixed summation order
// (a thread takes the four gates of one unit: 16-byte slab reads, eight in flight)
__global__ __launch_bounds__(256) void lstm_dw_reduce_bf16(const float* __restrict__ slabs, int nslab, int in, int in_i, int d,
                                                            float* __restrict__ dW, float* __restrict__ dbias) {
  const int N4 = 4 * d;
  const size_t slab_elems = (size_t)(in_i + d) * N4;
  const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (idx >= slab_elems) return;
  const int row = (int)(idx / N4), u = (int)(idx % N4) >> 2;  // slab columns follow dz's unit-major order 4u+g
  if (row > in && row < in_i) return;                           // zero pad rows
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= nslab; k += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(slabs + (size_t)(k + j) * slab_elems + idx);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; k < nslab; ++k) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)k * slab_elems + idx);
  float* dst = row < in ? dW + (size_t)row * N4 : (row == in ? dbias : dW + (size_t)(in + row - in_i) * N4);  // row == in: the ones column
#pragma unroll
  for (int g = 0; g < 4; ++g) dst[g * d + u] += s[g];
}

void launch_dw_reduce_bf16(const float* slabs, int nslab, int in, int in_i, int d, float* dW, float* dbias,
                           hipStream_t s) {
  const size_t n = (size_t)(in_i + d) * d;  // threads: one per (row, unit)
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
                                                                const bf16_t* __restrict__ B, float* __restrict__ C) {
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
    glds_mainloop_sp(mma, issue, K / 32, smem_h);
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
    glds_mainloop<true>(mma, issue, (K + 31) / 32, smem_h);
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

// rows of `len` fp32 values -> il32 two-term bf16 rows (2 len elements; len a multiple of 32)
__global__ void cvt_f32_il32_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t rows, int len) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * (size_t)len) return;
  const size_t r = i / len;
  const int e = (int)(i % len);
  const float v = src[i];
  const bf16_t hi = f2bf(v);
  bf16_t* row = dst + r * (size_t)(2 * len);
  row[il32(e)] = hi;
  row[il32(e) + 32] = f2bf(v - bf2f(hi));
}

// layouts 3 / 4: the split engine's row / k-major tiles on il32 two-term operands
__global__ __launch_bounds__(256, 1) void test_gemm_x2_rows_kernel(int M, int N, int K, const bf16_t* __restrict__ A,
                                                                    const bf16_t* __restrict__ B, float* __restrict__ C) {
  typedef TileCfgT<1, 2, 4, 2, 64> Cfg;
  typedef MmaBT<1, 2, 4, 2, 64, true> Mma;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int m0 = blockIdx.x * Cfg::BM, n0 = blockIdx.y * Cfg::BN;
  Mma mma;
  mma.init(threadIdx.x);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (unsigned)M * K * 4), rb = make_rsrc(B, (unsigned)N * K * 4);
  RowSrc<Cfg::A_GLDS, 64> sa;
  RowSrc<Cfg::B_GLDS, 64> sb;
  sa.setup(mma.wave_all, mma.lane, m0, M, K * 4);
  sb.setup(mma.wave_all, mma.lane, n0, N, K * 4);
  auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
    sa.issue(ra, As, mma.wave_all, tile * 128);
    sb.issue(rb, Bs, mma.wave_all, tile * 128);
  };
  glds_mainloop<false>(mma, issue, K / 32, smem_h);
#pragma unroll
  for (int ti = 0; ti < Mma::TM; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mma.row_of(ti, r);
#pragma unroll
      for (int tj = 0; tj < Mma::TN; ++tj) {
        const int n = n0 + mma.col_of(tj);
        if (m < M && n < N) C[(size_t)m * N + n] = mma.acc[ti][tj][r];
      }
    }
}

__global__ __launch_bounds__(512, 1) void test_gemm_x2_kmajor_kernel(int M, int N, int K, const bf16_t* __restrict__ A,
                                                                      const bf16_t* __restrict__ B, float* __restrict__ C) {
  typedef DwX2Cfg Cfg;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_h[];
  const int m0 = blockIdx.x * Cfg::BM, n0 = blockIdx.y * Cfg::BN;
  MmaX2K mma;
  mma.init(threadIdx.x);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (unsigned)K * M * 4), rb = make_rsrc(B, (unsigned)K * N * 4);
  KMajorSrc<Cfg::PCOLS, Cfg::A_GLDS> sa;
  KMajorSrc<Cfg::PCOLS, Cfg::B_GLDS> sb;
  sa.setup(mma.wave_all, mma.lane, 2 * m0, 2 * M, (unsigned)M * 4);
  sb.setup(mma.wave_all, mma.lane, 2 * n0, 2 * N, (unsigned)N * 4);
  auto issue = [&](int tile, bf16_t* As, bf16_t* Bs) {
    sa.issue(ra, As, mma.wave_all, (unsigned)tile * Cfg::KR * M * 4u);
    sb.issue(rb, Bs, mma.wave_all, (unsigned)tile * Cfg::KR * N * 4u);
  };
  glds_mainloop<true>(mma, issue, (K + Cfg::KR - 1) / Cfg::KR, smem_h);
#pragma unroll
  for (int ti = 0; ti < MmaX2K::TM; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mma.row_of(ti, r);
#pragma unroll
      for (int tj = 0; tj < MmaX2K::TN; ++tj) {
        const int n = n0 + mma.col_of(tj);
        if (m < M && n < N) C[(size_t)m * N + n] = mma.acc[ti][tj][r];
      }
    }
}

// Test hook only: rounds A and B to bf16 in stream-ordered temporaries (the one place the library allocates).
// Row images need K % 32 == 0; k-major images take any K (rows past the end fall off the descriptor).
// Layouts 3 / 4 (split engine): A [M,K], B [N,K] (3) or A [K,M], B [K,N] (4), every row converted to il32 two-term rows:
// row lengths must be multiples of 32.
int test_gemm_bf16(int layout, int M, int N, int K, const float* A, const float* B, float* C, hipStream_t s) {
  if (layout < 1 || layout > 4) return FVTA_ERR_UNSUPPORTED;
  if ((layout == 1 || layout == 3) && K % 32) return FVTA_ERR_UNSUPPORTED;
  if (layout == 4 && (M % 32 || N % 32)) return FVTA_ERR_UNSUPPORTED;
  const size_t na = (size_t)M * K, nb = (size_t)N * K, xm = layout >= 3 ? 2 : 1;
  bf16_t *Ab = nullptr, *Bb = nullptr;
  if (hipMallocAsync((void**)&Ab, na * 2 * xm, s) != hipSuccess || hipMallocAsync((void**)&Bb, nb * 2 * xm, s) != hipSuccess)
    return FVTA_ERR_LAUNCH;
  if (layout >= 3) {
    const int la = layout == 3 ? K : M, lb = layout == 3 ? K : N;
    hipLaunchKernelGGL(cvt_f32_il32_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, A, Ab, na / la, la);
    hipLaunchKernelGGL(cvt_f32_il32_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, B, Bb, nb / lb, lb);
    if (layout == 3) {
      constexpr int LDS = TileCfgT<1, 2, 4, 2, 64>::LDS_BYTES;
      allow_big_lds(test_gemm_x2_rows_kernel, LDS);
      hipLaunchKernelGGL(test_gemm_x2_rows_kernel, dim3((M + 255) / 256, (N + 127) / 128), dim3(256), LDS, s, M, N, K, Ab, Bb, C);
    } else {
      allow_big_lds(test_gemm_x2_kmajor_kernel, DwX2Cfg::LDS_BYTES);
      hipLaunchKernelGGL(test_gemm_x2_kmajor_kernel, dim3((M + 255) / 256, (N + 255) / 256), dim3(512), DwX2Cfg::LDS_BYTES, s, M, N, K, Ab, Bb, C);
    }
    (void)hipFreeAsync(Ab, s);
    (void)hipFreeAsync(Bb, s);
    return FVTA_OK;
  }
  hipLaunchKernelGGL(cvt_f32_bf16_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, A, Ab, na);
  hipLaunchKernelGGL(cvt_f32_bf16_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, B, Bb, nb);
  const dim3 grid((M + TileCfg::BM - 1) / TileCfg::BM, (N + TileCfg::BN - 1) / TileCfg::BN);
  if (layout == 1) {
    allow_big_lds(test_gemm_bf16_kernel<1>, TileCfg::LDS_BYTES);
    hipLaunchKernelGGL(test_gemm_bf16_kernel<1>, grid, dim3(256), TileCfg::LDS_BYTES, s, M, N, K, Ab, Bb, C);
  } else {
    allow_big_lds(test_gemm_bf16_kernel<2>, TileCfg::LDS_BYTES);
    hipLaunchKernelGGL(test_gemm_bf16_kernel<2>, grid, dim3(256), TileCfg::LDS_BYTES, s, M, N, K, Ab, Bb, C);
  }
  (void)hipFreeAsync(Ab, s);
  (void)hipFreeAsync(Bb, s);
  return FVTA_OK;
}

}  // namespace fvta

#ifdef FVTA_LOOP_STAMP
// the k-loop stamps of the LAST launch of a tiled kernel of this file (workgroup (8, 0, 3)): out[wave][4] cycles
extern "C" int fvta_debug_loop_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fvta::g_loop_stamp), sizeof(unsigned long long) * 16 * 4);
}
#endif
