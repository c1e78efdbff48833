"""GPU parity (backward): HIP gradients vs torch autograd on the fp64 fused oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-4, atol=1e-5, msg=""):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale, err_msg=msg)


@pytest.mark.parametrize("N,K,T,JQ,w", [(2, 3, 50, 10, 64), (3, 2, 100, 30, 256), (2, 6, 333, 30, 1024),
                                        (1, 2, 70, 60, 2048), (2, 1, 1100, 5, 128), (2, 2, 64, 7, 512)])
@pytest.mark.parametrize("simi,tanh", [(1, False), (2, True), (3, True), (4, False)])
@pytest.mark.parametrize("masked", [True, False])
def test_attention_3d_backward_matches_autograd(N, K, T, JQ, w, simi, tanh, masked):
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    from tests.test_gpu_forward import _att_case
    h, q, W, b, hm, qm = _att_case(N, K, T, JQ, w, simi, tanh, masked, seed=N * 100 + T + w + simi + 1)
    if masked:
        hm[0, 0, :3] = True   # keep every (n,k) with at least one valid row: the fully masked case is
        if N > 1:              # covered separately (documented deviation for the masked-logit gradient)
            qm[N - 1, :2] = True
    g = torch.Generator().manual_seed(99)
    gout = torch.randn(N, w, generator=g)
    hd, qd = h.double().requires_grad_(), q.double().requires_grad_()
    Wd, bd = (None, None) if W is None else (W.double().requires_grad_(), b.double().requires_grad_())
    ref_ha, _ = F.attention_3d(hd, qd, Wd, bd, hm, qm, simiMatrix=simi, add_tanh=tanh)
    (ref_ha * gout.double()).sum().backward()

    op = ops.FocalAttention(N, K, T, JQ, w, simi, tanh)
    cu = lambda t: None if t is None else t.cuda().contiguous()
    hc, qc, Wc, bc = cu(h), cu(q), (None if W is None else cu(W.reshape(-1))), cu(b)
    hmc, qmc = cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm))
    ha, _ = op.forward(hc, qc, hmc, qmc, Wc, bc)
    dh = torch.full_like(hc, 7.0)           # overwritten (accumulate=0)
    dq = torch.full_like(qc, 7.0)
    dW = None if Wc is None else torch.zeros_like(Wc)
    db = None if Wc is None else torch.zeros(1, device="cuda")
    op.backward(hc, qc, hmc, qmc, Wc, bc, cu(gout), dh, dq, dW, db, accumulate=False)
    _close(dh, hd.grad, msg="d_hinfo")
    _close(dq, qd.grad, msg="d_hq")
    if Wd is not None:
        _close(dW, Wd.grad.reshape(-1), msg="dW")
        _close(db, bd.grad, msg="db")
    # accumulate mode adds on top
    dh2 = torch.ones_like(hc)
    dq2 = torch.ones_like(qc)
    op.backward(hc, qc, hmc, qmc, Wc, bc, cu(gout), dh2, dq2, dW, db, accumulate=True)
    _close(dh2 - 1.0, hd.grad, atol=2e-5, msg="d_hinfo accumulate")
    _close(dq2 - 1.0, qd.grad, atol=2e-5, msg="d_hq accumulate")
    if Wd is not None:
        _close(dW, 2 * Wd.grad.reshape(-1), atol=3e-5, msg="dW accumulates")


@pytest.mark.parametrize("N,K,T,JQ,w,simi,tanh", [(2, 3, 48, 10, 64, 2, True), (2, 2, 96, 30, 256, 1, False),
                                                  (2, 6, 400, 30, 1024, 2, True), (1, 2, 64, 60, 2048, 3, True)])
@pytest.mark.parametrize("exact", [False, True])
def test_attention_3d_ties_among_valid_entries(N, K, T, JQ, w, simi, tanh, exact, attn_select):
    """Genuine ties among VALID entries (model_v2.py:268, 278): every question position has an identical twin
    (q[j + JQ/2] = q[j]: a tie in every row's max over j) and every context row has an identical twin in its modality
    (h[t + T/2] = h[t]: the max over t is always a tie).  The kernels send the max-over-j gradient to the FIRST arg-max and
    split the max-over-t gradient; TensorFlow's reduce_max (and the oracle's default, torch.amax) splits both.  Asserted:
      * forward values are the same in both conventions, and the kernels' match;
      * the kernels' gradients equal the oracle's under max_grad="first", element by element;
      * against TF's split convention ONLY d_hq differs, and only by moving gradient between twins: the sum over each twin
        pair is the split convention's (d_hinfo, dW, db are identical because twins have identical rows)."""
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    from tests.test_gpu_forward import _att_case
    (attn_select.exact if exact else attn_select.fast)()
    h, q, W, b, _, _ = _att_case(N, K, T, JQ, w, simi, tanh, False, seed=T + w + simi + 31)
    h[:, :, T // 2:] = h[:, :, :T - T // 2].clone()
    q[:, JQ // 2:] = q[:, :JQ - JQ // 2].clone()
    hm = torch.ones(N, K, T, dtype=torch.bool)
    qm = torch.ones(N, JQ, dtype=torch.bool)
    hm[:, :, T - 3:] = False                 # (a few padded rows / positions, so that the masked path runs too)
    qm[0, JQ - 1] = False
    gout = torch.randn(N, w, generator=torch.Generator().manual_seed(7))
    grads = {}
    for mode in ("first", "split"):
        leaves = [t.double().requires_grad_() for t in (h, q, W, b)]
        ha, _ = F.attention_3d(*leaves, hm, qm, simiMatrix=simi, add_tanh=tanh, max_grad=mode)
        (ha * gout.double()).sum().backward()
        grads[mode] = (ha.detach(), [t.grad for t in leaves])
    ref_ha, (rdh, rdq, rdW, rdb) = grads["first"]
    _, (sdh, sdq, sdW, sdb) = grads["split"]
    # (same values in both conventions -- up to the last fp64 bit: "first" picks ONE of two twins whose logits a blocked CPU
    #  GEMM may round differently)
    _close(grads["first"][0], grads["split"][0], rtol=1e-12, atol=1e-14, msg="h_a does not depend on the convention")
    op = ops.FocalAttention(N, K, T, JQ, w, simi, tanh)
    cu = lambda t: t.cuda().contiguous()
    hc, qc, Wc, bc = cu(h), cu(q), cu(W.reshape(-1)), cu(b)
    hmc, qmc = cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm))
    ha, _ = op.forward(hc, qc, hmc, qmc, Wc, bc)
    _close(ha, ref_ha, msg="h_a")
    dh, dq = torch.full_like(hc, float("nan")), torch.full_like(qc, float("nan"))
    dW, db = torch.zeros_like(Wc), torch.zeros(1, device="cuda")
    op.backward(hc, qc, hmc, qmc, Wc, bc, cu(gout), dh, dq, dW, db, accumulate=False)
    _close(dh, rdh, msg="d_hinfo (first arg-max)")
    _close(dq, rdq, msg="d_hq (first arg-max)")
    _close(dW, rdW.reshape(-1), msg="dW")
    _close(db, rdb, msg="db")
    assert float(dh[:, :, T - 3:].abs().max()) == 0.0            # masked rows: exactly zero, whatever was in the buffer
    # the difference to TF's convention, in numbers: the second twin of a pair gets nothing here and half there
    J2 = JQ - JQ // 2
    pair = lambda t: t[:, :JQ // 2] + t[:, J2:J2 + JQ // 2] if JQ % 2 == 0 else None
    assert float((rdq - sdq).abs().max()) > 1e-3 * float(sdq.abs().max())       # the test DOES exercise ties
    _close(rdh, sdh, rtol=1e-9, atol=1e-12, msg="d_hinfo does not depend on the convention")
    _close(rdW, sdW, rtol=1e-9, atol=1e-12, msg="dW does not depend on the convention")
    if JQ % 2 == 0:     # (album 0's last position is masked: that pair is no tie, and its sum is the first twin's in both conventions)
        _close(pair(dq.cpu().double()), pair(sdq), msg="twin-pair sums of d_hq = TF's split convention")


def test_attention_backward_fully_masked_rows_direct_term():
    """Fully masked (n,k): p = 1/T over all T; the direct term p*r*g still reaches d_hinfo."""
    from fvta_memexqa_amd import ops
    N, K, T, JQ, w = 1, 2, 40, 5, 64
    g = torch.Generator().manual_seed(3)
    h = torch.randn(N, K, T, w, generator=g)
    q = torch.randn(N, JQ, w, generator=g)
    W = torch.randn(2 * w, generator=g) * 0.1
    b = torch.zeros(1)
    hm = torch.zeros(N, K, T, dtype=torch.bool)        # both modalities empty -> r = 1/K
    qm = torch.ones(N, JQ, dtype=torch.bool)
    gout = torch.randn(N, w, generator=g)
    op = ops.FocalAttention(N, K, T, JQ, w, 2, True)
    cu = lambda t: t.cuda().contiguous()
    op.forward(cu(h), cu(q), cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm)), cu(W), cu(b))
    dh = torch.zeros(N, K, T, w, device="cuda")
    dq = torch.zeros(N, JQ, w, device="cuda")
    dW = torch.zeros(2 * w, device="cuda")
    db = torch.zeros(1, device="cuda")
    op.backward(cu(h), cu(q), cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm)), cu(W), cu(b), cu(gout), dh, dq, dW, db, False)
    exp = (gout[:, None, None, :] / (K * T)).expand(N, K, T, w)
    _close(dh, exp)
    assert float(dW.abs().max()) == 0.0 and float(dq.abs().max()) == 0.0


@pytest.mark.parametrize("B,J,din,d,dense,share,need_dx", [(5, 6, 8, 32, False, True, True),
                                                           (300, 9, 12, 64, False, True, True),
                                                           (130, 5, 200, 128, True, False, True),
                                                           (70, 17, 100, 64, False, False, False),
                                                           (64, 30, 200, 512, False, True, True)])
def test_bilstm_backward_matches_autograd(B, J, din, d, dense, share, need_dx):
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    g = torch.Generator().manual_seed(B + J + d + 1)
    x = torch.randn(B, J, din, generator=g)
    lens = torch.full((B,), J) if dense else torch.randint(0, J + 1, (B,), generator=g)
    lim = (6.0 / (din + d + 4 * d)) ** 0.5
    mk = lambda: ((torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim * 2, torch.randn(4 * d, generator=g) * 0.1)
    k_fw, b_fw = mk()
    k_bw, b_bw = (None, None) if share else mk()
    mask = torch.arange(J)[None, :] < lens[:, None]
    g_out = torch.randn(B, J, 2 * d, generator=g) * mask[:, :, None]
    g_last = torch.randn(B, 2 * d, generator=g)
    leaves = [t.double().requires_grad_() for t in (x, k_fw, b_fw)]
    if not share:
        leaves += [k_bw.double().requires_grad_(), b_bw.double().requires_grad_()]
    ref_out, ref_last = F.encode_stream(leaves[0], mask, leaves[1], leaves[2], *(leaves[3:] if not share else []))
    ((ref_out * g_out.double()).sum() + (ref_last * g_last.double()).sum()).backward()

    cu = lambda t: None if t is None else t.cuda().contiguous()
    xc, kf, bf, kb, bb = cu(x), cu(k_fw), cu(b_fw), cu(k_bw), cu(b_bw)
    out, last, op = ops.bilstm_simple(xc, lens, kf, bf, kb, bb, training=True)
    d_out = cu(g_out).clone()
    op.last_state_bwd(cu(g_last), 0, B, d_out)
    dx = torch.zeros_like(xc) if need_dx else None
    dkf, dbf = torch.zeros_like(kf), torch.zeros_like(bf)
    dkb, dbb = (None, None) if share else (torch.zeros_like(kb), torch.zeros_like(bb))
    op.backward(xc, out, d_out, kf, kb, dx, dkf, dbf, dkb, dbb)
    if need_dx:
        _close(dx, leaves[0].grad, msg="dx")
    _close(dkf, leaves[1].grad, msg="dkernel_fw")
    _close(dbf, leaves[2].grad, msg="dbias_fw")
    if not share:
        _close(dkb, leaves[3].grad, msg="dkernel_bw")
        _close(dbb, leaves[4].grad, msg="dbias_bw")


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("share", [True, False])
def test_bilstm_input_dropout(precision, share):
    """DropoutWrapper(cell, input_keep_prob) (model_v2.py:657-661): each direction of the bi-LSTM runs on its own dropped
    copy of the inputs.  The two copies (fvta_dropout_pair_fwd) bit for bit against the oracle's masks; outputs, last
    states and every gradient -- dx folded back through the masks (fvta_dropout_pair_bwd) -- against autograd of the
    restatement run with those masks."""
    from fvta_memexqa_amd import ops
    from fvta_memexqa_amd._lib import BF16, F32
    from oracle import fvta_fused as F
    B, J, din, d, keep, seed = 37, 7, 12, 32, 0.7, 0xC0FFEE123456789
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, J, din, generator=g)
    lens = torch.randint(0, J + 1, (B,), generator=g)
    lim = (6.0 / (din + d + 4 * d)) ** 0.5
    mk = lambda: ((torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim * 2, torch.randn(4 * d, generator=g) * 0.1)
    k_fw, b_fw = mk()
    k_bw, b_bw = (None, None) if share else mk()
    mask = torch.arange(J)[None, :] < lens[:, None]
    g_out = torch.randn(B, J, 2 * d, generator=g) * mask[:, :, None]
    g_last = torch.randn(B, 2 * d, generator=g)
    n = x.numel()
    keepm = F.dropout_keep_masks(n, keep, seed)
    assert 0.6 < keepm.float().mean() < 0.8 and not torch.equal(keepm[0], keepm[1])

    cu = lambda t: None if t is None else t.cuda().contiguous()
    xc = cu(x)
    x2 = torch.empty(2, n, device="cuda")
    ops.dropout_pair_fwd(xc, x2, keep, seed)
    scale = (torch.tensor(1.0) / torch.tensor(keep)).item()
    expect = torch.where(keepm, (x.reshape(1, n) * torch.tensor(scale, dtype=torch.float32)).expand(2, n), torch.zeros(2, n))
    assert torch.equal(x2.cpu(), expect)

    leaves = [t.double().requires_grad_() for t in (x, k_fw, b_fw)]
    if not share:
        leaves += [k_bw.double().requires_grad_(), b_bw.double().requires_grad_()]
    ref_out, ref_last = F.encode_stream(leaves[0], mask, leaves[1], leaves[2], *(leaves[3:] if not share else [None, None]),
                                        input_keep=keepm.reshape(2, B, J, din), keep_prob=keep)
    ((ref_out * g_out.double()).sum() + (ref_last * g_last.double()).sum()).backward()

    kf, bf, kb, bb = cu(k_fw), cu(b_fw), cu(k_bw), cu(b_bw)
    ar = torch.arange(B, dtype=torch.int64)
    op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                    share_fw_bw=share, precision=BF16 if precision == "bf16" else F32, training=True, x_bw_delta=n)
    op.make_plan(lens)
    out = torch.empty(B, J, 2 * d, device="cuda")
    op.forward(x2, out, kf, bf, kb, bb)
    last = torch.empty(B, 2 * d, device="cuda")
    op.last_state(out, 0, B, last)
    tol = dict(rtol=3e-2, atol=3e-2) if precision == "bf16" else {}
    _close(out, ref_out.detach(), msg="out", **tol)
    _close(last, ref_last.detach(), msg="last", **tol)
    d_out = cu(g_out).clone()
    op.last_state_bwd(cu(g_last), 0, B, d_out)
    dx2 = torch.zeros(2, n, device="cuda")
    dkf, dbf = torch.zeros_like(kf), torch.zeros_like(bf)
    dkb, dbb = (None, None) if share else (torch.zeros_like(kb), torch.zeros_like(bb))
    op.backward(x2, out, d_out, kf, kb, dx2, dkf, dbf, dkb, dbb)
    dx = torch.empty(n, device="cuda")
    ops.dropout_pair_bwd(dx2, dx, keep, seed)

    def close(a, b, tag):
        a, b = a.cpu().double().flatten(), b.double().flatten()
        err, ref = (a - b).norm().item(), b.norm().item()
        lim = 4e-2 if precision == "bf16" else 2e-4
        assert err <= lim * ref + 1e-6, "%s: |a-b| %.3g, |b| %.3g" % (tag, err, ref)

    close(dx, leaves[0].grad, "dx")
    close(dkf, leaves[1].grad, "dkernel_fw")
    close(dbf, leaves[2].grad, "dbias_fw")
    if not share:
        close(dkb, leaves[3].grad, "dkernel_bw")
        close(dbb, leaves[4].grad, "dbias_bw")
    # and the accumulate form
    base = torch.full((n,), 0.5, device="cuda")
    ops.dropout_pair_bwd(dx2, base, keep, seed, accumulate=True)
    assert torch.allclose(base, dx + 0.5, rtol=1e-6, atol=1e-6)
