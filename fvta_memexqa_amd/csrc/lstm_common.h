// Shared host/device structures of the bi-LSTM kernels (fp32 and bf16 engines).
#pragma once
#include "fvta_common.h"

namespace fvta {

typedef unsigned short bf16_t;

// ------------------------------------------------------------------ plan ----
struct PlanHeader {
  int64_t out_ld;
  int32_t B, J, in, d;
  int32_t pad[10];
};

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
  v.bytes = c.off;
  return v;
}

// ------------------------------------------------------------- saved state --
struct SavedView {
  float* gates;  // [2][J][B][4][d] i, tanh(j), f, o activations (fp32 engine: overwritten by dz in backward)
  float* cs;     // [2][J][B][d] cell state after step t
  size_t bytes;
};
static inline SavedView saved_view(const fvta_lstm_desc* d, void* p) {
  FvtaCarver c(p);
  SavedView s;
  s.gates = c.take<float>((size_t)2 * d->J * d->B * 4 * d->d);
  s.cs = c.take<float>((size_t)2 * d->J * d->B * d->d);
  s.bytes = c.off;
  return s;
}

static inline int dw_tgroup(const fvta_lstm_desc* d) {
  // steps per split-K slice of the weight-gradient GEMM: keep <= 16 slices per direction
  int g = (d->J + 15) / 16;
  return g < 1 ? 1 : g;
}
static inline int dw_nsplit(const fvta_lstm_desc* d) { return (d->J + dw_tgroup(d) - 1) / dw_tgroup(d); }
static inline int kpad8(const fvta_lstm_desc* d) { return (d->in + d->d + 7) / 8 * 8; }

struct WorkView {
  float* cstate;   // [2][B][d] running cell state (inference) / dc (backward)
  float* dh_rec;   // [2][B][d]
  float* slabs;    // [2*nsplit][(in+d+1)][4d] split-K partials of dW
  // bf16 engine only
  bf16_t* wt[2];   // [4d][Kp]   kernel^T, k-contiguous rows (forward B operand)
  bf16_t* wb[2];   // [in+d][4d] kernel, canonical layout (backward B operand)
  bf16_t* dzb;     // [2][J][B][4d] gate pre-activation gradients
  size_t bytes;
};
static inline WorkView work_view(const fvta_lstm_desc* d, void* p) {
  FvtaCarver c(p);
  WorkView w;
  w.cstate = c.take<float>((size_t)2 * d->B * d->d);
  w.dh_rec = c.take<float>((size_t)2 * d->B * d->d);
  w.slabs = c.take<float>((size_t)2 * dw_nsplit(d) * (d->in + d->d + 1) * 4 * d->d);
  w.wt[0] = w.wt[1] = w.wb[0] = w.wb[1] = nullptr;
  w.dzb = nullptr;
  if (d->precision == FVTA_BF16) {
    for (int i = 0; i < 2; ++i) {
      w.wt[i] = c.take<bf16_t>((size_t)4 * d->d * kpad8(d));
      w.wb[i] = c.take<bf16_t>((size_t)(d->in + d->d) * 4 * d->d);
    }
    if (d->training) w.dzb = c.take<bf16_t>((size_t)2 * d->J * d->B * 4 * d->d);
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
  const float* bias[2];
  float* gates;   // may be null (inference)
  float* cs;      // may be null
  float* cstate;  // used when cs is null
  int t, B, J, in, d, Kp;
};

struct GateBwdArgs {
  PlanView plan;
  const float* d_out;
  float* gates;
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
  int t, B, J, in, d;
};

struct DwArgs {
  PlanView plan;
  const float* x;
  const float* out;
  const float* dz;
  const bf16_t* dzb;
  float* slabs;
  int B, J, in, d, tgroup, nsplit;
};

#ifdef __HIPCC__
// Fused gate epilogue shared by both engines.  The block tile is 128 sorted sequences x
// (4 gates x 32 units) and a wave owns 32 rows x all four gate tiles, so lane (col = lane&31)
// holds z_i, z_j, z_f, z_o of the same (row, unit) in acc[0][0..3].
// BasicLSTMCell: c' = c*sig(f+1) + sig(i)*tanh(j); h' = tanh(c')*sig(o).
template <class Mma>
__device__ __forceinline__ void lstm_gate_epilogue(const Mma& mma, const StepArgs& a, int dir, int m0, int u0,
                                                   int nact, size_t trow) {
  const int d = a.d, t = a.t;
  const float* __restrict__ bias = a.bias[dir];
  const int u = u0 + mma.l31;
  const float bi = bias[u], bj = bias[d + u], bf = bias[2 * d + u], bo = bias[3 * d + u];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = mma.row_of(0, r);
    const int i = m0 + row;
    if (i >= nact) continue;
    const float ig = fvta_sigmoid(mma.acc[0][0][r] + bi);
    const float jg = tanhf(mma.acc[0][1][r] + bj);
    const float fg = fvta_sigmoid(mma.acc[0][2][r] + bf + 1.0f);  // forget_bias
    const float og = fvta_sigmoid(mma.acc[0][3][r] + bo);
    float cprev = 0.f;
    if (t > 0) cprev = a.cs ? a.cs[(trow - a.B + i) * d + u] : a.cstate[((size_t)dir * a.B + i) * d + u];
    const float c = cprev * fg + ig * jg;
    const float h = tanhf(c) * og;
    if (a.cs) {
      a.cs[(trow + i) * d + u] = c;
      float* g = a.gates + (trow + i) * (size_t)(4 * d) + u;
      g[0] = ig;
      g[d] = jg;
      g[2 * d] = fg;
      g[3 * d] = og;
    } else {
      a.cstate[((size_t)dir * a.B + i) * d + u] = c;
    }
    const int64_t oo = a.plan.oo[trow + i];
    a.out[oo + u] = h;
  }
}
#endif

// bf16 engine launchers (lstm_bf16.hip)
void launch_cvt_weights_bf16(const float* W, bf16_t* wt, bf16_t* wb, int K, int Kp, int N4, hipStream_t s);
void launch_step_fwd_bf16(const StepArgs& a, dim3 grid, hipStream_t s);
void launch_step_bwd_bf16(const StepBwdArgs& a, dim3 grid, hipStream_t s);
void launch_dw_bf16(const DwArgs& a, dim3 grid, hipStream_t s);
int test_gemm_bf16(int layout, int M, int N, int K, const float* A, const float* B, float* C, hipStream_t s);

}  // namespace fvta
