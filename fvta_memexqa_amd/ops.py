"""Thin host-side handles over the C ABI (include/fvta_hip.h).

These classes own the descriptor + scratch/saved buffers of one kernel family
and pass raw device pointers to libfvta_hip.so.  No arithmetic happens here and
there is no CPU path: every call needs a GPU and the built library.
"""
import ctypes

import torch

from . import _lib
from ._lib import (F32, BF16, AttnDesc, EmbedDesc, ImgTransDesc, LstmDesc, ScorerDesc, TimewarpDesc, check, ptr,
                   stream_ptr)


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.FvtaError("fvta_memexqa_amd needs an MI355X (no GPU visible); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _f32c(t):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expected a contiguous float32 CUDA tensor"
    return t


def _bytes(n, dev):
    return torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)


# ------------------------------------------------------------------ side stream
def concurrency_ratio(main, cand, spin_us=400, repeats=3, others=()):
    """time(one spin wave on each of the streams main, others..., cand) / time(one spin wave on `main`): about 1 when
    they all run concurrently, 2 or more when HIP mapped some of them onto the same hardware queue."""
    import time
    lib = _lib.load()

    def run(streams):
        best = float("inf")
        for _ in range(repeats):
            for s in streams:
                s.synchronize()
            t0 = time.perf_counter()
            for s in streams:
                check(lib.fvta_probe_spin(spin_us, ctypes.c_void_p(s.cuda_stream)), "fvta_probe_spin")
            for s in streams:
                s.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    group = [main] + list(others) + [cand]
    run(group)                              # first use creates the hardware queues
    return run(group) / run([main])


def pick_side_stream(dev, main=None, candidates=8, others=(), priority=0):
    """A HIP stream that really runs beside `main` (and beside every stream in `others`).  HIP maps streams onto a small pool of hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default) and two streams on one queue serialise; which streams collide depends on how
    many were created before -- an RCCL communicator created ahead of the model moved the side stream onto the main
    stream's queue and the training step lost the photo-cell overlap (21.9 ms instead of 15.7 ms,
    tools/dist_probe.py).  So the choice is measured: candidates are tried in turn with fvta_probe_spin and the
    first one that overlaps with `main` wins.  Rejected candidates stay referenced until the search ends so that the
    next one is mapped to another queue."""
    main = main if main is not None else torch.cuda.current_stream(dev)
    tried, best = [], None
    for _ in range(candidates):
        cand = torch.cuda.Stream(device=dev, priority=priority)
        r = concurrency_ratio(main, cand, others=others)
        tried.append(cand)
        if best is None or r < best[0]:
            best = (r, cand)
        if r < 1.4:
            break
    return best[1], best[0]


# ------------------------------------------------------------------ test hook
def test_gemm(A, B, layout, precision=F32):
    lib = _lib.load()
    require_gpu()
    if layout == 0:
        M, K = A.shape
        N = B.shape[1]
    elif layout in (1, 3):      # row images (3: the split engine's two-term il32 rows)
        M, K = A.shape
        N = B.shape[0]
    else:                       # k-major images (4: split engine)
        K, M = A.shape
        N = B.shape[1]
    C = torch.empty(M, N, device=A.device, dtype=torch.float32)
    check(lib.fvta_test_gemm(precision, layout, M, N, K, ptr(_f32c(A)), ptr(_f32c(B)), ptr(C), stream_ptr()), "test_gemm")
    return C


# ------------------------------------------------------------------- encoders
class BiLstm:
    """One bi-LSTM call group (all sequences sharing a cell): model_v2.py:694-823.

    x_off/out_off: int64 [B] element offsets of each sequence's first row in the
    x / out arenas; seq_J: padded length per sequence; out_ld: output row stride.
    """

    def __init__(self, B, J, in_dim, d, x_off, out_off, seq_J, out_ld, share_fw_bw=True, precision=F32,
                 training=False, prof_tag=0, x_bw_delta=0, dx_overwrite=False, out_pads_persist=False, out_skip=0):
        """x_bw_delta > 0: the backward direction reads x (writes dx) that many elements behind the forward direction's
        -- x = [x_fw | x_bw], the two dropped copies of DropoutWrapper's inputs (dropout_pair_fwd).
        dx_overwrite: backward() writes dx (zeros at padded positions) instead of adding to it: no memset by the caller.
        out_pads_persist: the caller leaves `out` alone between forward calls: only rows that turn into padding are zeroed.
        out_skip (bf16 engine): output half-rows at element offsets below it are not stored in fp32 -- their readers take
        the bf16 shadow rows (shadow_rows; FocalAttention.forward_shadow / backward_shadow)."""
        self.x_bw_delta = int(x_bw_delta)
        self.lib = _lib.load()
        self.dev = require_gpu()
        self.desc = LstmDesc(B, J, in_dim, d, int(share_fw_bw), precision, int(training), int(prof_tag), int(dx_overwrite), int(out_pads_persist),
                             int(out_skip))
        self.B, self.J, self.in_dim, self.d = B, J, in_dim, d
        self.x_off = x_off.to(self.dev, torch.int64).contiguous()
        self.out_off = out_off.to(self.dev, torch.int64).contiguous()
        self.seq_J = seq_J.to(self.dev, torch.int32).contiguous()
        self.out_ld = int(out_ld)
        r = ctypes.byref(self.desc)
        self.plan = _bytes(self.lib.fvta_lstm_plan_bytes(r), self.dev).zero_()   # (no stale plan state: out_pads_persist)
        self.saved = _bytes(self.lib.fvta_lstm_saved_bytes(r), self.dev)
        self.work = _bytes(self.lib.fvta_lstm_workspace_bytes(r), self.dev)
        self.training = training
        self.share = share_fw_bw

    def make_plan(self, lens):
        lens = lens.to(self.dev, torch.int32).contiguous()
        check(self.lib.fvta_lstm_plan_xdir(ctypes.byref(self.desc), ptr(lens), ptr(self.seq_J), ptr(self.x_off),
                                           ptr(self.out_off), self.out_ld, self.x_bw_delta, ptr(self.plan), stream_ptr()),
              "fvta_lstm_plan")

    def forward(self, x, out, kernel_fw, bias_fw, kernel_bw=None, bias_bw=None):
        check(self.lib.fvta_bilstm_fwd(ctypes.byref(self.desc), ptr(self.plan), ptr(_f32c(x)), ptr(_f32c(out)),
                                       ptr(_f32c(kernel_fw)), ptr(_f32c(bias_fw)), ptr(kernel_bw), ptr(bias_bw),
                                       ptr(self.saved), ptr(self.work), stream_ptr()), "fvta_bilstm_fwd")

    def shadow_rows(self, table, nrows):
        """table int64 [2, nrows] (device): the addresses of this call's bf16 output half-rows, by output row (out offset //
        out_ld) -- only the rows t < len of this plan's sequences are written, see fvta_lstm_shadow_rows.  After forward()."""
        assert table.dtype == torch.int64 and table.is_contiguous() and table.numel() == 2 * nrows
        check(self.lib.fvta_lstm_shadow_rows(ctypes.byref(self.desc), ptr(self.plan), ptr(self.saved), int(nrows), ptr(table),
                                             stream_ptr()), "fvta_lstm_shadow_rows")

    def set_active_hint(self, lens_host):
        """What the host knows about the lengths of the NEXT plan (a numpy / list of the B lengths, or None): the backward
        recurrence sizes each step's launch by its active sequences (fvta_bilstm_bwd_hint).  Optional; a stale hint costs
        time, never correctness."""
        if lens_host is None:
            self.active_hint = None
            return
        import numpy as np
        ln = np.clip(np.asarray(lens_host, dtype=np.int64).reshape(-1), 0, self.J)
        below = np.cumsum(np.bincount(ln, minlength=self.J + 1))[:self.J]      # sequences with len <= t
        self.active_hint = np.ascontiguousarray(ln.size - below, dtype=np.int32)   # nactive[t] = #(len > t)

    def backward(self, x, out, d_out, kernel_fw, kernel_bw, dx, dk_fw, db_fw, dk_bw=None, db_bw=None, side_stream=None):
        """side_stream (a torch stream, bf16 engine): dx and the per-step-group weight gradients run there, beside the
        recurrence (fvta_bilstm_bwd_overlap); the current stream has joined it again when this returns."""
        side = ctypes.c_void_p(side_stream.cuda_stream) if side_stream is not None else None
        hint = getattr(self, "active_hint", None)
        hp = hint.ctypes.data_as(ctypes.c_void_p) if hint is not None else None
        check(self.lib.fvta_bilstm_bwd_hint(ctypes.byref(self.desc), ptr(self.plan), ptr(x), ptr(out), ptr(_f32c(d_out)),
                                            ptr(kernel_fw), ptr(kernel_bw), ptr(self.saved), ptr(dx), ptr(dk_fw),
                                            ptr(db_fw), ptr(dk_bw), ptr(db_bw), ptr(self.work), stream_ptr(), side, hp),
              "fvta_bilstm_bwd")

    def last_state(self, out, s0, count, dst):
        check(self.lib.fvta_lstm_last_state(ctypes.byref(self.desc), ptr(self.plan), ptr(out), s0, count, ptr(dst),
                                            stream_ptr()), "fvta_lstm_last_state")

    def last_state_bwd(self, d_dst, s0, count, d_out):
        check(self.lib.fvta_lstm_last_state_bwd(ctypes.byref(self.desc), ptr(self.plan), ptr(_f32c(d_dst)), s0, count,
                                                ptr(d_out), stream_ptr()), "fvta_lstm_last_state_bwd")


def rows_from_shadow(table, nrows, d, out_ld, out):
    """out [nrows, out_ld] fp32 <- the bf16 rows behind a shadow table (inspection outputs, tests)"""
    check(_lib.load().fvta_rows_from_shadow(ptr(table), int(nrows), int(d), int(out_ld), ptr(out), stream_ptr()),
          "fvta_rows_from_shadow")


def dropout_pair_fwd(x, x2, keep_prob, seed):
    """x2 [2, numel(x)] = the forward / backward direction's dropped copy of x (DropoutWrapper, model_v2.py:657-661)"""
    check(_lib.load().fvta_dropout_pair_fwd(ptr(x), ptr(x2), x.numel(), float(keep_prob), int(seed) & (2 ** 64 - 1), stream_ptr()),
          "fvta_dropout_pair_fwd")


def dropout_pair_bwd(dx2, dx, keep_prob, seed, accumulate=False):
    check(_lib.load().fvta_dropout_pair_bwd(ptr(dx2), ptr(dx), dx.numel(), float(keep_prob), int(seed) & (2 ** 64 - 1),
                                            int(accumulate), stream_ptr()), "fvta_dropout_pair_bwd")


def bilstm_simple(x, lens, kernel_fw, bias_fw, kernel_bw=None, bias_bw=None, training=False, precision=F32):
    """x [B,J,in] dense, lens [B] -> out [B,J,2d], last [B,2d] (test/convenience form)."""
    B, J, din = x.shape
    d = kernel_fw.shape[1] // 4
    ar = torch.arange(B, dtype=torch.int64)
    op = BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                share_fw_bw=kernel_bw is None, precision=precision, training=training)
    op.make_plan(lens)
    out = torch.empty(B, J, 2 * d, device=x.device, dtype=torch.float32)
    op.forward(x.contiguous(), out, kernel_fw, bias_fw, kernel_bw, bias_bw)
    last = torch.empty(B, 2 * d, device=x.device, dtype=torch.float32)
    op.last_state(out, 0, B, last)
    return out, last, op


# ------------------------------------------------------------------ attention
class FocalAttention:
    """attention_3d (model_v2.py:210-298) / attention (125-201, K=1) handle."""

    def __init__(self, N, K, T, JQ, w, simi, add_tanh, feat_order=0, hinfo_stride=0):
        """hinfo_stride (K == 1): floats between consecutive batch rows of hinfo / d_hinfo, for a stream that lives
        inside a wider arena; the tensors handed to forward / backward then start at the stream's first row."""
        self.lib = _lib.load()
        self.dev = require_gpu()
        self.desc = AttnDesc(N, K, T, JQ, w, simi, feat_order, int(add_tanh), int(hinfo_stride))
        r = ctypes.byref(self.desc)
        sb = self.lib.fvta_attn_saved_bytes(r)
        if sb == 0:
            raise _lib.FvtaError("attention: " + self.lib.fvta_last_error().decode())
        self.saved = _bytes(sb, self.dev)
        self.work = _bytes(self.lib.fvta_attn_workspace_bytes(r), self.dev)
        self.N, self.K, self.T, self.JQ, self.w = N, K, T, JQ, w

    def forward(self, hinfo, hq, hmask, qmask, W, b, want_logits=False, tscale=None):
        """tscale [N,T] (time_warp_att, model_v2.py:269-275): the softmax over t runs on amax * tscale."""
        h_a = torch.empty(self.N, self.w, device=self.dev, dtype=torch.float32)
        a_logits = torch.empty(self.N, self.K, self.T, self.JQ, device=self.dev, dtype=torch.float32) if want_logits else None
        check(self.lib.fvta_attn_fwd_tw(ctypes.byref(self.desc), ptr(_f32c(hinfo)), ptr(_f32c(hq)), ptr(hmask), ptr(qmask),
                                        ptr(W), ptr(b), ptr(tscale), ptr(h_a), ptr(a_logits), ptr(self.saved),
                                        ptr(self.work), stream_ptr()), "fvta_attn_fwd")
        return h_a, a_logits

    def forward_shadow(self, table, hq, hmask, qmask, W, b):
        """forward() over the encoders' bf16 shadow rows: table int64 [2, N*K*T] (BiLstm.shadow_rows)"""
        h_a = torch.empty(self.N, self.w, device=self.dev, dtype=torch.float32)
        check(self.lib.fvta_attn_fwd_shadow(ctypes.byref(self.desc), ptr(table), ptr(_f32c(hq)), ptr(hmask), ptr(qmask),
                                            ptr(W), ptr(b), ptr(h_a), ptr(self.saved), ptr(self.work), stream_ptr()),
              "fvta_attn_fwd_shadow")
        return h_a

    def backward_shadow(self, table, hq, hmask, qmask, W, b, d_h_a, d_hinfo, d_hq, dW, db, accumulate):
        check(self.lib.fvta_attn_bwd_shadow(ctypes.byref(self.desc), ptr(table), ptr(hq), ptr(hmask), ptr(qmask), ptr(W),
                                            ptr(b), ptr(_f32c(d_h_a)), ptr(self.saved), ptr(d_hinfo), ptr(d_hq),
                                            ptr(dW), ptr(db), int(accumulate), ptr(self.work), stream_ptr()),
              "fvta_attn_bwd_shadow")

    def backward(self, hinfo, hq, hmask, qmask, W, b, d_h_a, d_hinfo, d_hq, dW, db, accumulate, tscale=None,
                 d_tscale=None):
        check(self.lib.fvta_attn_bwd_tw(ctypes.byref(self.desc), ptr(hinfo), ptr(hq), ptr(hmask), ptr(qmask), ptr(W),
                                        ptr(b), ptr(tscale), ptr(_f32c(d_h_a)), ptr(self.saved), ptr(d_hinfo), ptr(d_hq),
                                        ptr(dW), ptr(db), ptr(d_tscale), int(accumulate), ptr(self.work), stream_ptr()),
              "fvta_attn_bwd")


def linear_fwd(x, W, b, y, M, din, dout, add_tanh=False, blk=None):
    """y [M,dout] = x [M,din] W [din,dout] + b (+ tanh); blk = (rows_per_blk, blk_stride): x rows in blocks"""
    rpb, bs = blk if blk else (M, 0)
    check(_lib.load().fvta_linear_fwd_blk(ptr(x), ptr(W), ptr(b), ptr(y), M, din, dout, int(add_tanh), rpb, bs, stream_ptr()),
          "fvta_linear_fwd")


def linear_bwd(x, W, y, dy, dx, dW, db, M, din, dout, add_tanh=False, accumulate_dx=False, blk=None):
    """dx = dyt W^T (overwritten / added to), dW += x^T dyt, db += sum dyt; dyt = dy (1 - y^2) under add_tanh"""
    rpb, bs = blk if blk else (M, 0)
    check(_lib.load().fvta_linear_bwd_blk(ptr(x), ptr(W), ptr(y), ptr(dy), ptr(dx), ptr(dW), ptr(db), M, din, dout,
                                          int(add_tanh), int(accumulate_dx), rpb, bs, stream_ptr()), "fvta_linear_bwd")


def softmax_fwd(x, p, rows, J):
    check(_lib.load().fvta_softmax_fwd(ptr(x), ptr(p), rows, J, stream_ptr()), "fvta_softmax_fwd")


def softmax_bwd(p, dp, dx, rows, J):
    check(_lib.load().fvta_softmax_bwd(ptr(p), ptr(dp), ptr(dx), rows, J, stream_ptr()), "fvta_softmax_bwd")


def exp_mask(val, mask, out, n):
    check(_lib.load().fvta_exp_mask(ptr(val), ptr(mask), ptr(out), n, stream_ptr()), "fvta_exp_mask")


def wsum_fwd(target, weights, out, rows, J, d, target_ld=None):
    check(_lib.load().fvta_wsum_fwd_ld(ptr(target), ptr(weights), ptr(out), rows, J, d, J * d if target_ld is None else target_ld,
                                       stream_ptr()), "fvta_wsum_fwd")


def wsum_bwd(target, weights, d_out, d_weights, d_target, rows, J, d, target_ld=None):
    check(_lib.load().fvta_wsum_bwd(ptr(target), ptr(weights), ptr(d_out), ptr(d_weights), ptr(d_target), rows, J, d,
                                    J * d if target_ld is None else target_ld, stream_ptr()), "fvta_wsum_bwd")


def attn_qside_fwd(a_logits, hq, q_a, R, V, JQ, w):
    """q_a [R,w] = mean_v softsel(hq [R,JQ,w], a_logits [R,V,JQ])  (the bidirect branch, model.py:169-177)"""
    check(_lib.load().fvta_attn_qside_fwd(ptr(a_logits), ptr(hq), ptr(q_a), R, V, JQ, w, stream_ptr()), "fvta_attn_qside_fwd")


def attn_qside_bwd(a_logits, hq, d_q_a, dA, d_hq, R, V, JQ, w):
    """dA [R,V,JQ] overwritten, d_hq accumulated"""
    check(_lib.load().fvta_attn_qside_bwd(ptr(a_logits), ptr(hq), ptr(d_q_a), ptr(dA), ptr(d_hq), R, V, JQ, w, stream_ptr()),
          "fvta_attn_qside_bwd")


def rows_reduce(x, out, rows, J, d, out_ld=None, scale=1.0, accumulate=False):
    """out[r, :] (+)= scale * sum_j x[r, j, :]; `out` rows out_ld floats apart (fvta_rows_reduce)"""
    check(_lib.load().fvta_rows_reduce(ptr(x), ptr(out), rows, J, d, d if out_ld is None else out_ld, float(scale),
                                       int(accumulate), stream_ptr()), "fvta_rows_reduce")


def rows_broadcast(v, out, rows, J, d, v_ld=None, scale=1.0, accumulate=False):
    """out[r, j, :] (+)= scale * v[r, :]; `v` rows v_ld floats apart (fvta_rows_broadcast)"""
    check(_lib.load().fvta_rows_broadcast(ptr(v), ptr(out), rows, J, d, d if v_ld is None else v_ld, float(scale),
                                          int(accumulate), stream_ptr()), "fvta_rows_broadcast")


def _focal_logits_bwd(self, hinfo, hq, W, dA, d_hinfo, d_hq, dW, db):
    """dense logit gradient dA [N,T,JQ] of this K = 1 attention -> d_hinfo, d_hq (accumulated), dW, db (accumulated)"""
    if getattr(self, "dense_work", None) is None:
        self.dense_work = _bytes(self.lib.fvta_attn_logits_bwd_workspace_bytes(ctypes.byref(self.desc)), self.dev)
    check(self.lib.fvta_attn_logits_bwd(ctypes.byref(self.desc), ptr(hinfo), ptr(hq), ptr(W), ptr(dA), ptr(d_hinfo), ptr(d_hq),
                                        ptr(dW), ptr(db), ptr(self.dense_work), stream_ptr()), "fvta_attn_logits_bwd")


FocalAttention.logits_bwd = _focal_logits_bwd


def as_mask_u8(m):
    if m is None:
        return None
    return m.to(torch.uint8).contiguous()


# --------------------------------------------------------------------- scorer
def scorer_ce_fwd(gq, g1, gch, W, b, y=None, use_eu_output=False, add_tanh=False):
    lib = _lib.load()
    N, C, w = gch.shape
    desc = ScorerDesc(N, C, w, int(use_eu_output), int(add_tanh), 0)
    logits = torch.empty(N, C, device=gq.device, dtype=torch.float32)
    yp = torch.empty_like(logits)
    loss = torch.zeros(1, device=gq.device, dtype=torch.float32) if y is not None else None
    check(lib.fvta_scorer_ce_fwd(ctypes.byref(desc), ptr(_f32c(gq)), ptr(_f32c(g1)), ptr(_f32c(gch)), ptr(W), ptr(b),
                                 ptr(y), ptr(logits), ptr(yp), ptr(loss), stream_ptr()), "fvta_scorer_ce_fwd")
    return logits, yp, loss


def scorer_ce_bwd(gq, g1, gch, W, b, y, logits, yp, loss_scale, dW, db, use_eu_output=False, add_tanh=False,
                  tf_xent_grad=True):
    """tf_xent_grad: d logits = softmax - labels on every row (TF-1's kernel; all-False label rows still carry
    gradient); False: the gradient of -sum(y log softmax) proper."""
    lib = _lib.load()
    N, C, w = gch.shape
    desc = ScorerDesc(N, C, w, int(use_eu_output), int(add_tanh), 0 if tf_xent_grad else 1)
    dgq, dg1, dgch = torch.empty_like(gq), torch.empty_like(g1), torch.empty_like(gch)
    check(lib.fvta_scorer_ce_bwd(ctypes.byref(desc), ptr(gq), ptr(g1), ptr(gch), ptr(W), ptr(b), ptr(y), ptr(logits),
                                 ptr(yp), float(loss_scale), ptr(dgq), ptr(dg1), ptr(dgch), ptr(dW), ptr(db),
                                 stream_ptr()), "fvta_scorer_ce_bwd")
    return dgq, dg1, dgch


# ----------------------------------------------------------------- optimisers
def adadelta_step(var, grad, accum, accum_update, lr, rho=0.95, eps=1e-8, grad_scale=1.0):
    check(_lib.load().fvta_adadelta_step(ptr(var), ptr(grad), ptr(accum), ptr(accum_update), var.numel(), lr, rho, eps,
                                         grad_scale, stream_ptr()), "fvta_adadelta_step")


def adam_step(var, grad, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    check(_lib.load().fvta_adam_step(ptr(var), ptr(grad), ptr(m), ptr(v), var.numel(), lr, beta1, beta2, eps, int(t),
                                     grad_scale, stream_ptr()), "fvta_adam_step")


def weight_decay(var, grad, coef, loss):
    """add_wd (model_v2.py:347-354) for one variable: loss += coef/2 * sum(var^2), grad += coef * var (either may be None)."""
    check(_lib.load().fvta_weight_decay(ptr(var), ptr(grad) if grad is not None else None, var.numel(), float(coef),
                                        ptr(loss) if loss is not None else None, stream_ptr()), "fvta_weight_decay")


# ------------------------------------------------------------ AttentionGRUCell
def attgru_fwd(inputs, state, Wg, bg, Wc, Wi, bi):
    """attention_gru_cell.py:50-70, one step.  inputs [B,d+1] (last column = gate), state [B,d]."""
    lib = _lib.load()
    B, d = state.shape
    new_h = torch.empty_like(state)
    saved = torch.empty(B, 3 * d, device=state.device, dtype=torch.float32)
    check(lib.fvta_attgru_fwd(B, d, ptr(_f32c(inputs)), ptr(_f32c(state)), ptr(_f32c(Wg)), ptr(_f32c(bg)), ptr(_f32c(Wc)),
                              ptr(_f32c(Wi)), ptr(_f32c(bi)), ptr(new_h), ptr(saved), stream_ptr()), "fvta_attgru_fwd")
    return new_h, saved


def attgru_bwd(inputs, state, Wg, Wc, Wi, saved, d_new_h, dWg, dbg, dWc, dWi, dbi):
    """Gradients of one AttentionGRUCell step; parameter gradients are accumulated into."""
    lib = _lib.load()
    B, d = state.shape
    d_inputs = torch.empty_like(inputs)
    d_state = torch.empty_like(state)
    ws = torch.empty(B, 4 * d, device=state.device, dtype=torch.float32)
    check(lib.fvta_attgru_bwd(B, d, ptr(inputs), ptr(state), ptr(Wg), ptr(Wc), ptr(Wi), ptr(saved), ptr(_f32c(d_new_h)),
                              ptr(d_inputs), ptr(d_state), ptr(dWg), ptr(dbg), ptr(dWc), ptr(dWi), ptr(dbi), ptr(ws),
                              stream_ptr()), "fvta_attgru_bwd")
    return d_inputs, d_state


# ------------------------------------------------------------------ time warp
class TimeWarp:
    """model_v2.py:953-1009 (closed form): warp_h = hall * c[n,t] * cnt(t)."""

    def __init__(self, N, K, T, w, warp_type, window_t=3.0):
        self.lib = _lib.load()
        self.dev = require_gpu()
        self.desc = TimewarpDesc(N, K, T, w, int(warp_type), float(window_t))
        nb = self.lib.fvta_timewarp_workspace_bytes(ctypes.byref(self.desc))
        if nb == 0:
            raise Exception(self.lib.fvta_last_error().decode())     # "time warping type not implemented" (model_v2.py:341)
        self.work = _bytes(nb, self.dev)
        self.c = torch.empty(N, T, device=self.dev, dtype=torch.float32)
        self.scale = torch.empty(N, T, device=self.dev, dtype=torch.float32)

    def forward(self, hall, lq, WH_W, WH_b, WC_W, WC_b, warp_h):
        check(self.lib.fvta_timewarp_fwd(ctypes.byref(self.desc), ptr(_f32c(hall)), ptr(_f32c(lq)), ptr(WH_W), ptr(WH_b),
                                         ptr(WC_W), ptr(WC_b), ptr(_f32c(warp_h)), ptr(self.c), ptr(self.scale),
                                         ptr(self.work), stream_ptr()), "fvta_timewarp_fwd")

    def backward(self, hall, lq, WH_W, WH_b, WC_W, WC_b, d_warp, d_hall, d_lq, dWH_W, dWH_b, dWC_W, dWC_b,
                 d_scale_att=None):
        """d_scale_att [N,T]: the attention's gradient w.r.t. the per-position scale (use_time_warp_att)."""
        check(self.lib.fvta_timewarp_bwd_att(ctypes.byref(self.desc), ptr(hall), ptr(lq), ptr(WH_W), ptr(WH_b), ptr(WC_W),
                                             ptr(WC_b), ptr(self.c), ptr(_f32c(d_warp)), ptr(d_scale_att), ptr(d_hall),
                                             ptr(d_lq), ptr(dWH_W), ptr(dWH_b), ptr(dWC_W), ptr(dWC_b), ptr(self.work),
                                             stream_ptr()), "fvta_timewarp_bwd")


    def forward_shadow(self, table, lq, WH_W, WH_b, WC_W, WC_b, warp_rows):
        """forward() over the encoders' bf16 shadow rows: table int64 [2, N*K*T] (BiLstm.shadow_rows), warp_rows bf16
        [N*K*T, w] receives the warped rows (the focal attention reads them through a table of addresses into it)."""
        check(self.lib.fvta_timewarp_fwd_shadow(ctypes.byref(self.desc), ptr(table), ptr(_f32c(lq)), ptr(WH_W), ptr(WH_b),
                                                ptr(WC_W), ptr(WC_b), ptr(warp_rows), ptr(self.c), ptr(self.scale),
                                                ptr(self.work), stream_ptr()), "fvta_timewarp_fwd_shadow")

    def backward_shadow(self, table, lq, WH_W, WH_b, WC_W, WC_b, d_warp, d_hall, d_lq, dWH_W, dWH_b, dWC_W, dWC_b):
        check(self.lib.fvta_timewarp_bwd_shadow(ctypes.byref(self.desc), ptr(table), ptr(lq), ptr(WH_W), ptr(WH_b), ptr(WC_W),
                                                ptr(WC_b), ptr(self.c), ptr(_f32c(d_warp)), ptr(d_hall), ptr(d_lq),
                                                ptr(dWH_W), ptr(dWH_b), ptr(dWC_W), ptr(dWC_b), ptr(self.work),
                                                stream_ptr()), "fvta_timewarp_bwd_shadow")


# ------------------------------------------------------- embedding front-end
class TokenEmbed:
    """model_v2.py:524-620 for all text tokens of a batch: char-CNN + word lookup, rows written at tok_off.

    word_ids [ntok] i32, char_ids [ntok, W] i32 (None without the char-CNN), tok_off [ntok] i64 element offsets."""

    def __init__(self, ntok, W, cdim, cwdim, wdim, VW, VT, VC, height=5):
        self.lib = _lib.load()
        self.dev = require_gpu()
        self.desc = EmbedDesc(ntok, W, cdim, cwdim, wdim, VW, VT, VC, height)
        nb = self.lib.fvta_embed_workspace_bytes(ctypes.byref(self.desc))
        if nb == 0:
            raise _lib.FvtaError("embed: " + self.lib.fvta_last_error().decode())
        self.work = _bytes(nb, self.dev)
        self.argpos = torch.empty(max(ntok * cwdim, 1), dtype=torch.uint8, device=self.dev)
        self.cwdim = cwdim

    def set_dropout(self, keep_prob, seed):
        """conv1d's dropout of the gathered char embeddings for the NEXT forward / backward pair (model_v2.py:58-62,
        training only); keep_prob 1.0 switches it off."""
        self.desc.keep_prob = float(keep_prob)
        self.desc.dropout_seed = int(seed) & (2 ** 64 - 1)

    def forward(self, word_ids, char_ids, tok_off, word_emb, fixed_emb, char_emb, filt, bias, x):
        check(self.lib.fvta_embed_fwd(ctypes.byref(self.desc), ptr(word_ids), ptr(char_ids), ptr(tok_off), ptr(word_emb),
                                      ptr(fixed_emb), ptr(char_emb), ptr(filt), ptr(bias), ptr(x), ptr(self.argpos),
                                      stream_ptr()), "fvta_embed_fwd")

    def backward(self, word_ids, char_ids, tok_off, char_emb, filt, dx, d_word_emb, d_char_emb, d_filt, d_bias):
        check(self.lib.fvta_embed_bwd(ctypes.byref(self.desc), ptr(word_ids), ptr(char_ids), ptr(tok_off), ptr(char_emb),
                                      ptr(filt), ptr(self.argpos), ptr(dx), ptr(d_word_emb), ptr(d_char_emb),
                                      ptr(d_filt), ptr(d_bias), ptr(self.work), stream_ptr()), "fvta_embed_bwd")


class ImageTrans:
    """model_v2.py:634-645: photo feature lookup (+ image_trans_linear); rows written at row_off."""

    def __init__(self, M, idim, tdim, add_tanh):
        self.lib = _lib.load()
        self.dev = require_gpu()
        self.desc = ImgTransDesc(M, idim, tdim, int(add_tanh))
        self.work = torch.empty(M * tdim, dtype=torch.float32, device=self.dev)

    def forward(self, pidx, row_off, image_emb_mat, W, b, x):
        check(self.lib.fvta_image_trans_fwd(ctypes.byref(self.desc), ptr(pidx), ptr(row_off), ptr(_f32c(image_emb_mat)),
                                            ptr(W), ptr(b), ptr(x), stream_ptr()), "fvta_image_trans_fwd")

    def backward(self, pidx, row_off, image_emb_mat, x, dx, dW, db):
        check(self.lib.fvta_image_trans_bwd(ctypes.byref(self.desc), ptr(pidx), ptr(row_off), ptr(image_emb_mat), ptr(x),
                                            ptr(dx), ptr(dW), ptr(db), ptr(self.work), stream_ptr()),
              "fvta_image_trans_bwd")
