"""GPU parity (forward): HIP kernels through the C ABI vs the CPU oracle on seeded inputs.

Tolerance: BASELINE.json north_star -> 1e-4 relative fp32 (bit-exact answer argmax)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _close(a, b, rtol=RTOL, atol=1e-5, msg=""):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(128, 128, 16), (200, 260, 72), (64, 36, 100), (712, 2048, 256)])
def test_mfma_f32_gemm_layouts(layout, shape):
    """fragment maps of the fp32 MFMA tile engine, asymmetric operands."""
    from fvta_memexqa_amd import ops
    M, N, K = shape
    M, N, K = (M + 3) // 4 * 4, (N + 3) // 4 * 4, (K + 3) // 4 * 4
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + layout)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    ref = A.double() @ B.double()
    Ad = (A if layout != 2 else A.t().contiguous()).cuda()
    Bd = (B if layout != 1 else B.t().contiguous()).cuda()
    C = ops.test_gemm(Ad, Bd, layout)
    _close(C, ref, rtol=1e-5, atol=1e-4 * K ** 0.5)


@pytest.mark.parametrize("B,J,din,d,dense,share", [(5, 6, 8, 32, False, True), (300, 9, 12, 64, False, True),
                                                   (130, 5, 200, 128, True, False), (64, 30, 200, 512, False, True)])
def test_bilstm_forward_matches_oracle(B, J, din, d, dense, share):
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    g = torch.Generator().manual_seed(B + J + d)
    x = torch.randn(B, J, din, generator=g)
    lens = torch.full((B,), J) if dense else torch.randint(0, J + 1, (B,), generator=g)
    lim = (6.0 / (din + d + 4 * d)) ** 0.5
    k_fw = (torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim * 2
    b_fw = torch.randn(4 * d, generator=g) * 0.1
    k_bw = None if share else (torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim * 2
    b_bw = None if share else torch.randn(4 * d, generator=g) * 0.1
    mask = torch.arange(J)[None, :] < lens[:, None]
    ref_out, ref_last = F.encode_stream(x.double(), mask, k_fw.double(), b_fw.double(),
                                        None if share else k_bw.double(), None if share else b_bw.double())
    cu = lambda t: None if t is None else t.cuda()
    for training in (False, True):
        out, last, _ = ops.bilstm_simple(cu(x), lens, cu(k_fw), cu(b_fw), cu(k_bw), cu(b_bw), training=training)
        _close(out, ref_out, msg="out training=%s" % training)
        _close(last, ref_last, msg="last")


def _att_case(N, K, T, JQ, w, simi, tanh, masked, seed, p_valid=0.6):
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(N, K, T, w, generator=g) * 0.5
    q = torch.randn(N, JQ, w, generator=g) * 0.5
    F_ = {1: 3 * w, 2: 2 * w, 3: 4 * w, 4: 0}[simi]
    W = torch.randn(F_, 1, generator=g) * 0.1 if F_ else None
    b = torch.randn(1, generator=g) * 0.1 if F_ else None
    if masked:
        hm = torch.rand(N, K, T, generator=g) < p_valid
        qm = torch.rand(N, JQ, generator=g) < 0.8
        qm[:, 0] = True
        hm[0, 0] = False                      # a fully masked modality
        if N > 1:
            qm[N - 1] = False                 # a padded batch row
        if K > 1:
            hm[0, 1] = False
            hm[0, 1, T // 2] = True           # single valid row
    else:
        hm = qm = None
    return h, q, W, b, hm, qm


@pytest.mark.parametrize("N,K,T,JQ,w", [(2, 3, 50, 10, 64), (3, 2, 100, 30, 256), (2, 6, 333, 30, 1024),
                                        (1, 2, 70, 60, 2048), (2, 1, 40, 33, 128), (2, 2, 64, 7, 512)])
@pytest.mark.parametrize("simi,tanh", [(1, False), (2, True), (3, True), (4, False)])
@pytest.mark.parametrize("masked", [True, False])
def test_attention_3d_forward_matches_oracle(N, K, T, JQ, w, simi, tanh, masked):
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    h, q, W, b, hm, qm = _att_case(N, K, T, JQ, w, simi, tanh, masked, seed=N * 100 + T + w + simi)
    dd = lambda t: None if t is None else t.double()
    ref_ha, ref_a = F.attention_3d(dd(h), dd(q), dd(W), dd(b), hm, qm, simiMatrix=simi, add_tanh=tanh)
    op = ops.FocalAttention(N, K, T, JQ, w, simi, tanh)
    cu = lambda t: None if t is None else t.cuda().contiguous()
    ha, a = op.forward(cu(h), cu(q), cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm)),
                       None if W is None else cu(W.reshape(-1)), cu(b), want_logits=True)
    _close(ha, ref_ha, msg="h_a")
    _close(a, ref_a, rtol=1e-4, atol=2e-5, msg="a_logits")


def test_attention_1d_question_form():
    """model_v2.py:1044: attention(hq, g1[:,None,:], q_mask, ones) == K=1, 'JQ'=1."""
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    g = torch.Generator().manual_seed(11)
    N, JQ, w = 4, 9, 128
    hq = torch.randn(N, JQ, w, generator=g) * 0.5
    g1 = torch.randn(N, 1, w, generator=g) * 0.5
    W = torch.randn(2 * w, 1, generator=g) * 0.1
    b = torch.randn(1, generator=g) * 0.1
    qm = torch.arange(JQ)[None, :] < torch.tensor([9, 4, 1, 0])[:, None]
    ones = torch.ones(N, 1, dtype=torch.bool)
    ref, ref_a = F.attention(hq.double(), g1.double(), W.double(), b.double(), qm, ones, simiMatrix=2, add_tanh=True)
    op = ops.FocalAttention(N, 1, JQ, 1, w, 2, True)
    ha, a = op.forward(hq.cuda(), g1.cuda(), ops.as_mask_u8(qm).cuda(), ops.as_mask_u8(ones).cuda(),
                       W.reshape(-1).cuda(), b.cuda(), want_logits=True)
    _close(ha, ref)
    _close(a.reshape(N, JQ, 1), ref_a, atol=2e-5)


@pytest.mark.parametrize("eu,tanh,tf_grad", [(False, False, True), (True, True, True), (False, False, False)])
def test_scorer_ce_forward_and_backward(eu, tanh, tf_grad):
    """tf_grad: TF-1's kernel gradient softmax - labels (the padded last row, labels all False, still carries
    gradient; the default) vs the mathematical one."""
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    g = torch.Generator().manual_seed(5)
    N, C, w = 6, 4, 256
    gq = (torch.randn(N, w, generator=g) * 0.5).double().requires_grad_()
    g1 = (torch.randn(N, w, generator=g) * 0.5).double().requires_grad_()
    gch = (torch.randn(N, C, w, generator=g) * 0.5).double().requires_grad_()
    W = (torch.randn((7 if eu else 5) * w, 1, generator=g) * 0.1).double().requires_grad_()
    b = (torch.randn(1, generator=g) * 0.1).double().requires_grad_()
    y = torch.zeros(N, C, dtype=torch.bool)
    y[torch.arange(N - 1), torch.randint(0, C, (N - 1,), generator=g)] = True   # last row = padded (all False)
    logits, yp = F.scorer(gq, g1, gch, W, b, eu, tanh)
    loss = F.softmax_cross_entropy_mean(logits, y, tf_grad=tf_grad)
    loss.backward()
    f = lambda t: t.detach().float().cuda().contiguous()
    yd = ops.as_mask_u8(y).cuda()
    l2, yp2, loss2 = ops.scorer_ce_fwd(f(gq), f(g1), f(gch), f(W).reshape(-1), f(b), yd, eu, tanh)
    _close(l2, logits)
    _close(yp2, yp)
    _close(loss2, loss.reshape(1))
    assert (yp2.argmax(1).cpu() == yp.argmax(1)).all()
    dW = torch.zeros(W.numel(), device="cuda")
    db = torch.zeros(1, device="cuda")
    dgq, dg1, dgch = ops.scorer_ce_bwd(f(gq), f(g1), f(gch), f(W).reshape(-1), f(b), yd, l2, yp2, 1.0, dW, db, eu, tanh,
                                       tf_xent_grad=tf_grad)
    # the padded row: softmax/N through the scorer with TF's gradient, nothing with the mathematical one
    assert (dgch[N - 1].abs().max().item() > 0) == tf_grad
    _close(dgq, gq.grad, atol=1e-6)
    _close(dg1, g1.grad, atol=1e-6)
    _close(dgch, gch.grad, atol=1e-6)
    _close(dW, W.grad.reshape(-1), atol=1e-6)
    _close(db, b.grad, atol=1e-6)


def test_optimizer_steps_match_oracle():
    from fvta_memexqa_amd import ops
    from oracle import fvta_literal as L
    rng = np.random.default_rng(0)
    n = 1000
    var, grad = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    a, au = np.abs(rng.standard_normal(n)).astype(np.float32), np.abs(rng.standard_normal(n)).astype(np.float32) * 0.01
    ev, ea, eau = L.adadelta_step(var.astype(np.float64), grad.astype(np.float64), a.astype(np.float64), au.astype(np.float64), 0.5)
    tv, ta, tau = [torch.from_numpy(v.copy()).cuda() for v in (var, a, au)]
    ops.adadelta_step(tv, torch.from_numpy(grad).cuda(), ta, tau, 0.5)
    _close(tv, ev, atol=1e-6)
    _close(ta, ea, atol=1e-6)
    _close(tau, eau, atol=1e-6)
    m, v = rng.standard_normal(n).astype(np.float32) * 0.1, np.abs(rng.standard_normal(n)).astype(np.float32) * 0.1
    ev, em, evv = L.adam_step(var.astype(np.float64), grad.astype(np.float64), m.astype(np.float64), v.astype(np.float64), 3, 0.001)
    tv, tm, tvv = [torch.from_numpy(x.copy()).cuda() for x in (var, m, v)]
    ops.adam_step(tv, torch.from_numpy(grad).cuda(), tm, tvv, 3, 0.001)
    _close(tv, ev, atol=1e-6)
    _close(tm, em, atol=1e-6)
    _close(tvv, evv, atol=1e-6)


@pytest.mark.parametrize("w,JQ", [(1024, 30), (512, 17), (2048, 30), (1024, 40)])
def test_attention_forward_is_bitwise_reproducible(w, JQ):
    """Same inputs, 12 launches: bitwise equal outputs and saved state.  (Guards the asynchronous operand pipelines
    of the forward kernels -- a write-after-read race there shows up as run-to-run differences, not as a crash.)"""
    from fvta_memexqa_amd import ops
    N, K, T = 4, 6, 333
    h, q, W, b, hm, qm = _att_case(N, K, T, JQ, w, 2, True, True, seed=w + JQ)
    cu = lambda t: None if t is None else t.cuda().contiguous()
    op = ops.FocalAttention(N, K, T, JQ, w, 2, True)
    args = (cu(h), cu(q), cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm)), cu(W.reshape(-1)), cu(b))
    ref = None
    for _ in range(12):
        ha, _ = op.forward(*args)
        torch.cuda.synchronize()
        cur = (ha.clone(), op.saved.clone())
        if ref is None:
            ref = cur
        else:
            assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])


def test_attention_fast_path_matches_exact_path(attn_select):
    """JQ <= 32, 128 <= w <= 1024 runs the fp16 3-term-split kernel; fvta_attn_kernel_select(1, .) routes to the fp32-MFMA kernel.
    The two must agree far inside the 1e-4 parity tolerance."""
    from fvta_memexqa_amd import ops
    N, K, T, JQ, w = 3, 5, 200, 30, 1024
    h, q, W, b, hm, qm = _att_case(N, K, T, JQ, w, 2, True, True, seed=77)
    cu = lambda t: None if t is None else t.cuda().contiguous()
    op = ops.FocalAttention(N, K, T, JQ, w, 2, True)
    args = (cu(h), cu(q), cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm)), cu(W.reshape(-1)), cu(b))
    fast, _ = op.forward(*args)
    attn_select.exact()
    exact, _ = op.forward(*args)
    _close(fast, exact.cpu(), rtol=2e-5, atol=2e-6, msg="fast vs exact")


def test_attention_fast_path_randomised_differential(attn_select):
    """The 16-row kernel loads its rows through inline asm into pinned registers with hand-counted waits (a misplaced
    wait or a moved register shows as garbage in SOME stream position): 24 random shapes / maskings / stream lengths
    (1 to many tiles per workgroup, several items per workgroup, fully masked and single-row modalities, n changing inside
    a workgroup's stream), each against the exact-fp32 kernel, values and arg-max positions."""
    from fvta_memexqa_amd import ops
    rng = np.random.RandomState(2024)
    cu = lambda t: None if t is None else t.cuda().contiguous()
    for case in range(24):
        w = int(rng.choice([128, 256, 512, 1024]))
        N, K = int(rng.randint(1, 9)), int(rng.randint(1, 8))
        T = int(rng.choice([5, 16, 17, 48, 150, 333, 700, 1300]))
        JQ = int(rng.randint(1, 33))
        simi = int(rng.choice([1, 2, 3, 4]))
        masked = bool(rng.rand() < 0.8)
        h, q, W, b, hm, qm = _att_case(N, K, T, JQ, w, simi, True, masked, seed=1000 + case, p_valid=float(rng.choice([0.1, 0.6, 0.95])))
        op = ops.FocalAttention(N, K, T, JQ, w, simi, True)
        args = (cu(h), cu(q), cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm)), cu(W.reshape(-1)) if W is not None else None, cu(b))
        attn_select.fast()
        fast, _ = op.forward(*args)
        fast, saved_fast = fast.cpu(), op.saved.clone()
        again, _ = op.forward(*args)
        assert torch.equal(again.cpu(), fast), "case %d: not reproducible" % case
        attn_select.exact()
        exact, _ = op.forward(*args)
        tag = "case %d (N %d K %d T %d JQ %d w %d simi %d masked %s)" % (case, N, K, T, JQ, w, simi, masked)
        assert torch.isfinite(fast).all(), tag
        _close(fast, exact.cpu(), rtol=5e-5, atol=5e-6, msg=tag)
        nkt = N * K * T
        amax_f = saved_fast[:4 * nkt].view(torch.float32).cpu()
        amax_e = op.saved[:4 * nkt].view(torch.float32).cpu()
        ok = torch.isfinite(amax_e) & (amax_e > -1e29)
        np.testing.assert_allclose(amax_f[ok].numpy(), amax_e[ok].numpy(), rtol=5e-5, atol=5e-6, err_msg=tag + " amax")


@pytest.mark.parametrize("skew", ["one_long", "all_but_one_empty", "two_sizes"])
def test_attention_pair_kernel_balance_on_skewed_batches(attn_select, skew):
    """With masks the pair kernel deals workgroups to albums in proportion to their valid row tiles (attn_balance_kernel),
    so the split points of an album's partial sums -- the fp32 rounding of its h_a, nothing else -- depend on the OTHER
    albums of the batch.  The table's edge cases at N = 64: one album holding nearly every tile (its share clamps at the
    per-album maximum), every other album empty or fully masked, and two sizes; each against the exact-fp32 kernel, and
    an album's result against the same album in a batch of different neighbours (equal up to that rounding)."""
    from fvta_memexqa_amd import ops
    N, K, T, JQ, w = 64, 6, 1200, 30, 1024
    g = torch.Generator().manual_seed(4242)
    h = torch.randn(N, K, T, w, generator=g) * 0.5
    q = torch.randn(N, JQ, w, generator=g) * 0.5
    W = torch.randn(2 * w, 1, generator=g) * 0.1
    b = torch.randn(1, generator=g) * 0.1
    hm = torch.zeros(N, K, T, dtype=torch.bool)
    qm = torch.ones(N, JQ, dtype=torch.bool)
    if skew == "one_long":
        hm[5] = True                                   # every row of one album
        hm[:, :, :3] |= torch.rand(N, K, 3, generator=g) < 0.5          # the others: at most three rows per modality
    elif skew == "all_but_one_empty":
        hm[17, :, :700] = True
        qm[40:] = False                                # padded batch rows
    else:
        hm[::2, :, :1000] = True
        hm[1::2, :2, :40] = True
    cu = lambda t: t.cuda().contiguous()
    op = ops.FocalAttention(N, K, T, JQ, w, 2, True)
    args = (cu(h), cu(q), cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm)), cu(W.reshape(-1)), cu(b))
    attn_select.fast()
    fast, _ = op.forward(*args)
    fast = fast.cpu()
    again, _ = op.forward(*args)
    assert torch.equal(again.cpu(), fast)              # the same batch: bitwise
    attn_select.exact()
    exact, _ = op.forward(*args)
    assert torch.isfinite(fast).all()
    _close(fast, exact.cpu(), rtol=5e-5, atol=5e-6, msg=skew)
    # the long album among other neighbours: only the rounding of its partial sums may move
    attn_select.fast()
    n0 = {"one_long": 5, "all_but_one_empty": 17, "two_sizes": 0}[skew]
    hm2 = hm.clone()
    hm2[(n0 + 1) % N] = True
    other, _ = op.forward(args[0], args[1], cu(ops.as_mask_u8(hm2)), *args[3:])
    _close(other.cpu()[n0], fast[n0], rtol=2e-5, atol=2e-6, msg=skew + ": album among other neighbours")


@pytest.mark.parametrize("mode", ["1", "2", "3"])
def test_attention_wave16_kernel_randomised_differential(attn_select, mode):
    """fvta_attn_kernel_select(0, 1 / 2 / 3): the one-wave-per-tile (attn_fwd_wave16) and two-waves-per-tile (attn_fwd_pair16: w >= 512,
    else the former) forward kernels on random shapes / maskings / stream
    lengths against the exact-fp32 kernel -- values, saved max-pooled logits and arg-max positions -- and against itself
    (bitwise reproducible); then the backward pass on ITS saved state against the backward on the exact kernel's."""
    from fvta_memexqa_amd import ops
    rng = np.random.RandomState(4242)
    cu = lambda t: None if t is None else t.cuda().contiguous()
    for case in range(20):
        w = int(rng.choice([256, 512, 1024]))
        N, K = int(rng.randint(1, 9)), int(rng.randint(1, 8))
        T = int(rng.choice([5, 16, 17, 48, 150, 333, 700, 1300]))
        JQ = int(rng.randint(1, 33))
        simi = int(rng.choice([1, 2, 3]))
        tanh = bool(rng.rand() < 0.5)
        masked = bool(rng.rand() < 0.8)
        h, q, W, b, hm, qm = _att_case(N, K, T, JQ, w, simi, tanh, masked, seed=3000 + case, p_valid=float(rng.choice([0.1, 0.6, 0.95])))
        op = ops.FocalAttention(N, K, T, JQ, w, simi, tanh)
        args = (cu(h), cu(q), cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm)), cu(W.reshape(-1)), cu(b))
        attn_select.fast(mode)
        fast, _ = op.forward(*args)
        fast, saved_fast = fast.cpu(), op.saved.clone()
        again, _ = op.forward(*args)
        tag = "case %d (N %d K %d T %d JQ %d w %d simi %d tanh %s masked %s)" % (case, N, K, T, JQ, w, simi, tanh, masked)
        assert torch.equal(again.cpu(), fast) and torch.equal(op.saved, saved_fast), tag + ": not reproducible"
        g = torch.randn(N, w, generator=torch.Generator().manual_seed(case)).cuda()
        # (accumulate = 0 OVERWRITES d_hinfo / d_hq: NaN sentinels -- a valid row the kernel missed would stay NaN, a masked
        #  row must come back exactly 0)
        grads_fast = [torch.full((N, K, T, w), float("nan"), device="cuda"), torch.full((N, JQ, w), float("nan"), device="cuda"),
                      torch.zeros_like(args[4]), torch.zeros(1, device="cuda")]
        op.backward(*args, g, *grads_fast, accumulate=0)
        assert torch.isfinite(grads_fast[0]).all() and torch.isfinite(grads_fast[1]).all(), tag + ": rows left unwritten"
        if args[2] is not None:
            dead = (args[2].view(N, K, T) == 0) & (args[2].view(N, K, T) != 0).any(2, keepdim=True) & (args[3].view(N, 1, JQ) != 0).any(2, keepdim=True)
            assert float(grads_fast[0][dead].abs().max() if dead.any() else 0.0) == 0.0, tag + ": masked rows of a live stream not zero"
        attn_select.exact()
        exact, _ = op.forward(*args)
        assert torch.isfinite(fast).all(), tag
        _close(fast, exact.cpu(), rtol=5e-5, atol=5e-6, msg=tag)
        nkt = N * K * T
        amax_f = saved_fast[:4 * nkt].view(torch.float32).cpu()
        amax_e = op.saved[:4 * nkt].view(torch.float32).cpu()
        ok = torch.isfinite(amax_e) & (amax_e > -1e29)
        np.testing.assert_allclose(amax_f[ok].numpy(), amax_e[ok].numpy(), rtol=5e-5, atol=5e-6, err_msg=tag + " amax")
        grads_exact = [torch.full((N, K, T, w), float("nan"), device="cuda"), torch.full((N, JQ, w), float("nan"), device="cuda"),
                       torch.zeros_like(args[4]), torch.zeros(1, device="cuda")]
        op.backward(*args, g, *grads_exact, accumulate=0)
        for name, a_, b_ in zip(("d_hinfo", "d_hq", "dW", "db"), grads_fast, grads_exact):
            _close(a_, b_.cpu(), rtol=2e-4, atol=2e-5, msg=tag + " " + name)


def test_attention_wide_kernel_randomised_differential(attn_select):
    """attn_fwd_wide (w = 2048: BASELINE.json configs[4]'s rows; the rows-stationary, question-streaming kernel) on random
    shapes / maskings / stream lengths -- one- and two-pass question lengths, runs crossing stream boundaries, fully masked
    and single-row streams, ragged last tiles -- against the exact-fp32 kernel: values, saved max-pooled logits and arg-max
    positions; bitwise reproducible; the backward on ITS saved state against the backward on the exact kernel's."""
    from fvta_memexqa_amd import ops
    rng = np.random.RandomState(777)
    cu = lambda t: None if t is None else t.cuda().contiguous()
    w = 2048
    for case in range(14):
        N, K = int(rng.randint(1, 7)), int(rng.randint(1, 8))
        T = int(rng.choice([5, 32, 33, 96, 150, 333, 700, 1300]))
        JQ = int(rng.choice([1, 17, 32, 33, 47, 60, 64]))
        simi = int(rng.choice([1, 2, 3]))
        tanh = bool(rng.rand() < 0.5)
        masked = bool(rng.rand() < 0.8)
        h, q, W, b, hm, qm = _att_case(N, K, T, JQ, w, simi, tanh, masked, seed=5000 + case, p_valid=float(rng.choice([0.1, 0.6, 0.95])))
        op = ops.FocalAttention(N, K, T, JQ, w, simi, tanh)
        args = (cu(h), cu(q), cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm)), cu(W.reshape(-1)), cu(b))
        attn_select.fast()
        fast, _ = op.forward(*args)
        fast, saved_fast = fast.cpu(), op.saved.clone()
        again, _ = op.forward(*args)
        tag = "case %d (N %d K %d T %d JQ %d simi %d tanh %s masked %s)" % (case, N, K, T, JQ, simi, tanh, masked)
        assert torch.equal(again.cpu(), fast) and torch.equal(op.saved, saved_fast), tag + ": not reproducible"
        g = torch.randn(N, w, generator=torch.Generator().manual_seed(case)).cuda()
        # (accumulate = 0 OVERWRITES d_hinfo / d_hq: NaN sentinels -- a valid row the kernel missed would stay NaN, a masked
        #  row must come back exactly 0)
        grads_fast = [torch.full((N, K, T, w), float("nan"), device="cuda"), torch.full((N, JQ, w), float("nan"), device="cuda"),
                      torch.zeros_like(args[4]), torch.zeros(1, device="cuda")]
        op.backward(*args, g, *grads_fast, accumulate=0)
        assert torch.isfinite(grads_fast[0]).all() and torch.isfinite(grads_fast[1]).all(), tag + ": rows left unwritten"
        if args[2] is not None:
            dead = (args[2].view(N, K, T) == 0) & (args[2].view(N, K, T) != 0).any(2, keepdim=True) & (args[3].view(N, 1, JQ) != 0).any(2, keepdim=True)
            assert float(grads_fast[0][dead].abs().max() if dead.any() else 0.0) == 0.0, tag + ": masked rows of a live stream not zero"
        attn_select.exact()
        exact, _ = op.forward(*args)
        assert torch.isfinite(fast).all(), tag
        _close(fast, exact.cpu(), rtol=5e-5, atol=5e-6, msg=tag)
        nkt = N * K * T
        amax_f = saved_fast[:4 * nkt].view(torch.float32).cpu()
        amax_e = op.saved[:4 * nkt].view(torch.float32).cpu()
        ok = torch.isfinite(amax_e) & (amax_e > -1e29)
        np.testing.assert_allclose(amax_f[ok].numpy(), amax_e[ok].numpy(), rtol=5e-5, atol=5e-6, err_msg=tag + " amax")
        grads_exact = [torch.full((N, K, T, w), float("nan"), device="cuda"), torch.full((N, JQ, w), float("nan"), device="cuda"),
                       torch.zeros_like(args[4]), torch.zeros(1, device="cuda")]
        op.backward(*args, g, *grads_exact, accumulate=0)
        for name, a_, b_ in zip(("d_hinfo", "d_hq", "dW", "db"), grads_fast, grads_exact):
            # (absolute tolerance relative to the tensor's scale: dW sums N K T rows of 2048 channels)
            _close(a_, b_.cpu(), rtol=2e-4, atol=2e-5 * max(1.0, float(b_.abs().max())), msg=tag + " " + name)
