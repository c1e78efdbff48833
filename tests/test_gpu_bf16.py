"""GPU checks of the bf16 MFMA engine (BASELINE.json configs[2]: bf16 compute / fp32 accumulate).

The fragment maps are checked EXACTLY (operands pre-rounded to bf16, so only fp32 accumulation
order differs).  The LSTM / whole-model results are compared with the fp64 oracle at a bf16
tolerance (operands carry 8 significant bits): 3e-2 absolute on activations in [-1,1],
and the fp32 engine remains the 1e-4 parity path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F32, BF16 = 0, 1


def _close(a, b, rtol, atol, msg=""):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("shape", [(128, 128, 32), (200, 264, 96), (64, 40, 160), (712, 2048, 256), (300, 136, 1000)])
def test_mfma_bf16_gemm_layouts(layout, shape):
    from fvta_memexqa_amd import ops
    M, N, K = [(v + 7) // 8 * 8 for v in shape]
    if layout == 1 and K % 32:
        pytest.skip("row images need K % 32 == 0 (the LSTM's internal layouts guarantee it)")
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + layout)
    A = torch.randn(M, K, generator=g).bfloat16().float()     # asymmetric, exactly representable
    B = torch.randn(K, N, generator=g).bfloat16().float()
    ref = A.double() @ B.double()
    Ad = (A if layout == 1 else A.t().contiguous()).cuda()
    Bd = (B.t().contiguous() if layout == 1 else B).cuda()
    C = ops.test_gemm(Ad, Bd, layout, precision=BF16)
    _close(C, ref, rtol=1e-5, atol=1e-4 * K ** 0.5)


@pytest.mark.parametrize("layout", [3, 4])
@pytest.mark.parametrize("shape", [(128, 128, 32), (200, 264, 96), (64, 40, 160), (712, 2048, 256), (300, 136, 1000), (448, 512, 12864)])
def test_mfma_split_engine_gemm_layouts(layout, shape):
    """The split engine's tiles (precision bf16x3: two stored bf16 terms per value in the il32 layout, three MFMA products
    hi hi + hi lo + lo hi per k-step): full fp32 operands, result within 2^-16 of sum |a||b| -- layout 3 the row tile
    (forward / backward / dx), layout 4 the k-major tile of the weight gradient."""
    from fvta_memexqa_amd import ops
    M, N, K = [(v + 31) // 32 * 32 for v in shape] if layout == 4 else [(v + 7) // 8 * 8 for v in shape]
    if layout == 3:
        K = (K + 31) // 32 * 32
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + layout)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    ref = A.double() @ B.double()
    bound = (A.double().abs() @ B.double().abs()) * 2.0 ** -16 + 1e-6
    Ad = (A if layout == 3 else A.t().contiguous()).cuda()
    Bd = (B.t().contiguous() if layout == 3 else B).cuda()
    C = ops.test_gemm(Ad, Bd, layout, precision=BF16).cpu().double()
    assert bool(((C - ref).abs() <= bound).all()), "max error %.3g against a bound of %.3g" % (
        float((C - ref).abs().max()), float(bound.max()))
    # ... and far inside it in the mean: a dropped term (2^-9 per product) would show
    assert float((C - ref).abs().mean() / ref.abs().mean()) < 2e-5


# (shapes 3-6 run the weights-in-registers forward step kernel, csrc/lstm_wreg.hip: in_i / d = 128 / 128,
#  224 / 512, 224 / 512 with separate fw / bw kernels, 32 / 128 with a ragged tail tile; the first two the tiled one.
#  Every d = 512 shape runs the pipelined backward step lstm_bwd_ring_bf16; the last two give its workgroups two and
#  three row tiles each -- both exits of the loop unrolled by two -- with a ragged / a dense tail)
@pytest.mark.parametrize("B,J,din,d,dense,share", [(5, 6, 8, 32, False, True), (300, 9, 16, 64, False, False),
                                                   (130, 7, 104, 128, True, True), (64, 30, 200, 512, False, True),
                                                   (70, 6, 200, 512, False, False), (33, 5, 12, 128, False, True),
                                                   (1500, 4, 200, 512, False, True), (1030, 3, 100, 512, True, False),
                                                   # more than 8192 active rows per step: the regime bench.py times, i.e.
                                                   # the tiled backward step kernel lstm_bwd_fused_bf16 (dense steps) --
                                                   # and, ragged, its hand-over to the pipelined one below 8192 rows
                                                   (9000, 3, 200, 512, True, True), (9100, 4, 200, 512, False, True)])
def test_bilstm_bf16_forward_backward(B, J, din, d, dense, share):
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    g = torch.Generator().manual_seed(B + J + d + 5)
    x = torch.randn(B, J, din, generator=g)
    lens = torch.full((B,), J) if dense else torch.randint(0, J + 1, (B,), generator=g)
    lim = (6.0 / (din + d + 4 * d)) ** 0.5
    mk = lambda: ((torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim, torch.randn(4 * d, generator=g) * 0.1)
    k_fw, b_fw = mk()
    k_bw, b_bw = (None, None) if share else mk()
    mask = torch.arange(J)[None, :] < lens[:, None]
    g_out = torch.randn(B, J, 2 * d, generator=g) * mask[:, :, None]
    leaves = [t.double().requires_grad_() for t in (x, k_fw, b_fw)]
    if not share:
        leaves += [k_bw.double().requires_grad_(), b_bw.double().requires_grad_()]
    ref_out, ref_last = F.encode_stream(leaves[0], mask, leaves[1], leaves[2], *(leaves[3:] if not share else []))
    (ref_out * g_out.double()).sum().backward()
    cu = lambda t: None if t is None else t.cuda().contiguous()
    xc, kf, bf, kb, bb = cu(x), cu(k_fw), cu(b_fw), cu(k_bw), cu(b_bw)
    out, last, op = ops.bilstm_simple(xc, lens, kf, bf, kb, bb, training=True, precision=BF16)
    _close(out, ref_out, rtol=0, atol=3e-2, msg="out")
    _close(last, ref_last, rtol=0, atol=3e-2, msg="last")
    dx = torch.zeros_like(xc)
    dkf, dbf = torch.zeros_like(kf), torch.zeros_like(bf)
    dkb, dbb = (None, None) if share else (torch.zeros_like(kb), torch.zeros_like(bb))
    op.backward(xc, out, cu(g_out), kf, kb, dx, dkf, dbf, dkb, dbb)

    def rel(a, b, name, tol=4e-2):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        err = (a - b).norm() / (b.norm() + 1e-30)
        assert err < tol, "%s: relative L2 error %.4f" % (name, err)

    rel(dx, leaves[0].grad, "dx")
    rel(dkf, leaves[1].grad, "dkernel_fw")
    rel(dbf, leaves[2].grad, "dbias_fw")
    if not share:
        rel(dkb, leaves[3].grad, "dkernel_bw")


def test_model_bf16_matches_oracle_loosely_and_fp32_engine_closely():
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs, make_params
    from oracle import fvta_fused as F
    spec = SynthSpec(dense=False, **CONFIGS["plumbing"])
    params, inputs = make_params(spec), make_inputs(spec)
    with torch.no_grad():
        ref = F.fvta_forward({k: v.double() for k, v in params.items()},
                             {k: v for k, v in __import__("fvta_memexqa_amd.synth", fromlist=["to_dtype"]).to_dtype(inputs, torch.float64).items()},
                             spec.cfg())
    outs = {}
    for prec in ("f32", "bf16"):
        model = Model(dict(spec.cfg(), batch_size=spec.N, precision=prec), text_in=spec.text_in, img_in=spec.img_in)
        model.set_oracle_params(params)
        L = model.load_inputs(inputs, training=True)
        model.zero_grad()
        outs[prec] = model.forward(L).cpu().double()
        model.backward(L)
        outs[prec + "_g"] = torch.from_numpy(model.get_oracle_grads()["text_kernel"]).double()
    _close(outs["f32"], ref["yp"], rtol=1e-4, atol=1e-5, msg="fp32 engine")
    _close(outs["bf16"], ref["yp"], rtol=0, atol=2e-2, msg="bf16 engine")
    err = (outs["bf16_g"] - outs["f32_g"]).norm() / outs["f32_g"].norm()
    assert err < 5e-2, "bf16 text_kernel gradient off by %.4f (relative L2)" % err


@pytest.mark.parametrize("B,J,din,d,dense", [(300, 9, 16, 64, False), (1100, 30, 200, 256, True), (700, 33, 40, 128, False)])
def test_bilstm_bf16_backward_overlapped_equals_serial(B, J, din, d, dense):
    """fvta_bilstm_bwd_overlap (dx and the per-step-group weight gradients on a lent side stream, beside the
    recurrence) against the serial fvta_bilstm_bwd: weight and bias gradients bitwise, dx to the order of its two
    float-atomic addends.  Repeated, so that a missing event dependency between the two streams would show."""
    from fvta_memexqa_amd import ops
    g = torch.Generator().manual_seed(B + J)
    x = torch.randn(B, J, din, generator=g).cuda()
    lens = torch.full((B,), J) if dense else torch.randint(0, J + 1, (B,), generator=g)
    lim = (6.0 / (din + 5 * d)) ** 0.5
    kf = ((torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim).cuda()
    bf = (torch.randn(4 * d, generator=g) * 0.1).cuda()
    mask = torch.arange(J)[None, :] < lens[:, None]
    g_out = (torch.randn(B, J, 2 * d, generator=g) * mask[:, :, None]).cuda()
    out, last, op = ops.bilstm_simple(x, lens, kf, bf, training=True, precision=BF16)
    side = torch.cuda.Stream()

    def run(side_stream):
        dx, dk, db = torch.zeros_like(x), torch.zeros_like(kf), torch.zeros_like(bf)
        op.backward(x, out, g_out, kf, None, dx, dk, db, side_stream=side_stream)
        torch.cuda.synchronize()
        return dx, dk, db

    ref = run(None)
    assert float(ref[1].abs().max()) > 0
    for _ in range(3):
        cur = run(side)
        assert torch.equal(cur[1], ref[1]), "dkernel"
        assert torch.equal(cur[2], ref[2]), "dbias"
        _close(cur[0], ref[0], rtol=1e-5, atol=1e-6, msg="dx")


def test_backward_with_host_lengths_hint_is_bitwise_the_same(monkeypatch):
    """fvta_bilstm_bwd_hint: the host's knowledge of the lengths only picks each step's block tile (few active rows: the
    small one) -- gradients are bitwise those of the unhinted call, also under a WRONG hint (costs time, never
    correctness)."""
    from fvta_memexqa_amd import ops
    from fvta_memexqa_amd._lib import BF16
    g = torch.Generator().manual_seed(31)
    B, J, din, d = 700, 9, 16, 256
    x = torch.randn(B, J, din, generator=g).cuda()
    lens = torch.randint(0, J + 1, (B,), generator=g)
    lim = (6.0 / (din + d + 4 * d)) ** 0.5
    k = ((torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim * 2).cuda()
    b = (torch.randn(4 * d, generator=g) * 0.1).cuda()
    mask = (torch.arange(J)[None, :] < lens[:, None])
    g_out = (torch.randn(B, J, 2 * d, generator=g) * mask[:, :, None]).cuda()
    out, last, op = ops.bilstm_simple(x, lens, k, b, None, None, training=True, precision=BF16)

    def grads(hint):
        op.set_active_hint(hint)
        dx, dk, db = torch.zeros_like(x), torch.zeros_like(k), torch.zeros_like(b)
        op.backward(x, out, g_out.clone(), k, None, dx, dk, db)
        torch.cuda.synchronize()
        return dx, dk, db

    base = grads(None)
    assert float(base[1].abs().max()) > 0
    for hint in (lens.numpy(), np.zeros(B, np.int64), np.full(B, J)):
        got = grads(hint)
        for a, e, tag in zip(got, base, ("dx", "dkernel", "dbias")):
            assert torch.equal(a, e), tag
    op.set_active_hint(lens.numpy())
    assert op.active_hint.tolist() == [int((lens > t).sum()) for t in range(J)]


# ------------------------------------------------------------------ the split-bf16 engine (FVTA_BF16X3)
@pytest.mark.parametrize("B,J,din,d,dense,share", [(5, 6, 8, 32, False, True), (300, 9, 12, 64, False, True),
                                                   (130, 5, 200, 128, True, False), (70, 17, 100, 64, False, False),
                                                   (64, 30, 200, 512, False, True), (333, 12, 200, 1024, False, True),
                                                   # the photo cell's regime (<= 64 sequences, input <= 128 wide, d = 512): the
                                                   # backward step and the input gradient on the four-wave 64 x 128 tiles
                                                   (40, 7, 100, 512, False, True), (64, 40, 100, 512, True, False)])
def test_bilstm_bf16x3_meets_the_fp32_tolerances(B, J, din, d, dense, share):
    """precision = bf16x3: every MFMA operand split in two bf16 terms, three products per GEMM (hi hi + hi lo + lo hi),
    fp32 saved gates -- the bi-LSTM forward and every gradient against autograd of the fp64 oracle at north_star's
    tolerance, 1e-4 relative (rtol 1e-4 plus 3e-5 x max|ref| absolute: a two-term bf16 split carries 16-17 significant
    bits per operand, so a sum of K products is good to ~2e-5 of its scale -- the exact-fp32 engine's tests hold 1e-5)."""
    from fvta_memexqa_amd import ops
    from fvta_memexqa_amd._lib import BF16X3
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
    out, last, op = ops.bilstm_simple(xc, lens, kf, bf, kb, bb, training=True, precision=BF16X3)

    def close(a, b, msg, rtol=1e-4, atol=3e-5):
        a, b = a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * max(1.0, float(np.abs(b).max())), err_msg=msg)

    close(out, ref_out, "out")
    close(last, ref_last, "last")
    d_out = cu(g_out).clone()
    op.last_state_bwd(cu(g_last), 0, B, d_out)
    dx = torch.zeros_like(xc)
    dkf, dbf = torch.zeros_like(kf), torch.zeros_like(bf)
    dkb, dbb = (None, None) if share else (torch.zeros_like(kb), torch.zeros_like(bb))
    op.backward(xc, out, d_out, kf, kb, dx, dkf, dbf, dkb, dbb)
    close(dx, leaves[0].grad, "dx")
    close(dkf, leaves[1].grad, "dkernel_fw")
    close(dbf, leaves[2].grad, "dbias_fw")
    if not share:
        close(dkb, leaves[3].grad, "dkernel_bw")
        close(dbb, leaves[4].grad, "dbias_bw")


def test_model_bf16x3_train_step_at_the_metric_shape_meets_the_fp32_tolerances():
    """BASELINE.json configs[2]'s shape at N = 2, precision = bf16x3: yp, loss (1e-4 relative, arg-max exact) and every
    parameter gradient (rtol 2e-4, atol 5e-5 x max|ref|) vs the fp64 oracle."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs, make_params, to_dtype
    from oracle import fvta_fused as F
    spec = SynthSpec(dense=False, **dict(CONFIGS["metric"], N=2))
    params, inputs = make_params(spec), make_inputs(spec)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    ref = F.fvta_forward(p64, to_dtype(inputs, torch.float64), spec.cfg())
    ref["loss"].backward()
    model = Model(dict(spec.cfg(), batch_size=spec.N, precision="bf16x3"), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L)
    model.backward(L, need_dx=True)

    def close(a, b, msg, rtol, atol):
        a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
        b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * max(1.0, float(np.abs(b).max())), err_msg=msg)

    close(yp, ref["yp"], "yp", 1e-4, 3e-5)
    assert (yp.argmax(1).cpu() == ref["yp"].argmax(1)).all()
    close(model.loss, ref["loss"].reshape(1), "loss", 1e-4, 3e-5)
    grads = model.get_oracle_grads()
    for k, v in p64.items():
        if v.grad is not None:
            close(grads[k].reshape(v.grad.shape), v.grad, "grad " + k, 2e-4, 5e-5)


@pytest.mark.parametrize("B,J,din,dense", [(2100, 5, 200, False), (1500, 3, 100, True), (40, 7, 200, False)])
def test_bilstm_bf16_backward_step_kernels_agree(B, J, din, dense):
    """The pipelined weights-stationary backward step (lstm_bwd_ring_bf16, d = 512) against the tiled one
    (lstm_bwd_fused_bf16) on the same saved state: the two differ in the summation order of dh_rec only (four partial
    sums per output), so dz agrees up to rounding flips of single bf16 values -- gradients within 2e-3 relative L2, far
    inside what a stale operand tile (a mis-counted hand-over) would do.  The pipelined kernel twice: bitwise."""
    from fvta_memexqa_amd import _lib, ops
    lib = _lib.load()
    d = 512
    g = torch.Generator().manual_seed(B + J)
    x = torch.randn(B, J, din, generator=g).cuda()
    lens = torch.full((B,), J) if dense else torch.randint(0, J + 1, (B,), generator=g)
    lim = (6.0 / (din + 5 * d)) ** 0.5
    kf = ((torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim).cuda()
    bf = (torch.randn(4 * d, generator=g) * 0.1).cuda()
    mask = torch.arange(J)[None, :] < lens[:, None]
    g_out = (torch.randn(B, J, 2 * d, generator=g) * mask[:, :, None]).cuda()
    out, last, op = ops.bilstm_simple(x, lens, kf, bf, None, None, training=True, precision=BF16)
    res = {}
    prev = lib.fvta_lstm_kernel_select(-1)
    try:
        for name, mode in (("ring", 7), ("tiled", 1), ("ring2", 7)):
            lib.fvta_lstm_kernel_select(mode)
            dx, dk, db = torch.zeros_like(x), torch.zeros_like(kf), torch.zeros_like(bf)
            op.backward(x, out, g_out, kf, None, dx, dk, db, None, None)
            torch.cuda.synchronize()
            res[name] = (dx, dk, db)
    finally:
        lib.fvta_lstm_kernel_select(-1 if prev == 7 else prev)
    for a, b in zip(res["ring"], res["ring2"]):
        if a is not res["ring"][0]:        # (dx meets its two directions with float atomics: order-dependent last bits)
            assert torch.equal(a, b)
    for name, a, b in zip(("dx", "dkernel", "dbias"), res["ring"], res["tiled"]):
        err = ((a - b).double().norm() / (b.double().norm() + 1e-30)).item()
        assert err < 2e-3, "%s: pipelined vs tiled backward step differ by %.5f (relative L2)" % (name, err)


@pytest.mark.parametrize("precision", [F32, BF16, 2])   # 2: the split-bf16 engine (its dx kernels add: rows zeroed by the library)
@pytest.mark.parametrize("B,J,din,d,dense,xdir", [(700, 5, 200, 512, False, False), (300, 4, 200, 512, True, False),
                                                   (300, 6, 100, 128, False, False), (260, 5, 200, 512, False, True)])
def test_bilstm_dx_overwrite_equals_zero_fill_and_accumulate(precision, B, J, din, d, dense, xdir):
    """fvta_lstm_desc.dx_overwrite: backward() WRITES dx -- the input gradient at t < len, zeros at len <= t < seq_J,
    whatever the buffer held -- bitwise what the accumulate contract gives on a zeroed buffer.  Wide input (both directions in
    one lstm_dx_bf16 launch, dx never read), narrow input and the fp32 engine (rows zeroed by the library, then added to),
    and a separate input per direction (fvta_lstm_plan_xdir: both copies)."""
    from fvta_memexqa_amd import ops
    g = torch.Generator().manual_seed(B + J + din)
    lens = torch.full((B,), J) if dense else torch.randint(0, J + 1, (B,), generator=g)
    n = B * J * din
    x = torch.randn((2 if xdir else 1) * n, generator=g).cuda()
    lim = (6.0 / (din + 5 * d)) ** 0.5
    kf = ((torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim).cuda()
    bf = (torch.randn(4 * d, generator=g) * 0.1).cuda()
    mask = torch.arange(J)[None, :] < lens[:, None]
    g_out = (torch.randn(B, J, 2 * d, generator=g) * mask[:, :, None]).cuda()
    ar = torch.arange(B, dtype=torch.int64)
    res = []
    for overwrite in (False, True):
        op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                        share_fw_bw=True, precision=precision, training=True, x_bw_delta=n if xdir else 0,
                        dx_overwrite=overwrite)
        op.make_plan(lens)
        out = torch.zeros(B, J, 2 * d, device="cuda")
        op.forward(x, out, kf, bf)
        dx = torch.full_like(x, float("nan")) if overwrite else torch.zeros_like(x)
        dk, db = torch.zeros_like(kf), torch.zeros_like(bf)
        op.backward(x, out, g_out, kf, None, dx, dk, db, None, None)
        torch.cuda.synchronize()
        res.append((dx, dk, db))
    assert torch.isfinite(res[1][0]).all(), "dx_overwrite left rows unwritten"
    pad = (~mask).cuda()
    for half in range(2 if xdir else 1):
        assert (res[1][0][half * n:(half + 1) * n].view(B, J, din)[pad] == 0).all()
    assert res[0][0].abs().max() > 0
    for name, a, b in zip(("dx", "dkernel", "dbias"), res[0], res[1]):
        assert torch.equal(a, b), "%s differs between the accumulate and the overwrite contract" % name


@pytest.mark.parametrize("precision", [F32, BF16])
def test_bilstm_out_pads_persist_zeroes_what_turns_into_padding(precision):
    """fvta_lstm_desc.out_pads_persist: the plan remembers how far the last forward wrote into which buffer and zeroes only
    [len, that) -- the output must equal a fresh op's (which zeroes every padded row) for any sequence of batches: shrinking
    and growing lengths, a plan that is replaced before any forward ran on it, another output buffer in between."""
    from fvta_memexqa_amd import ops
    B, J, din, d = 300, 7, 40, 128
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, J, din, generator=g).cuda()
    kf = ((torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * 0.1).cuda()
    bf = (torch.randn(4 * d, generator=g) * 0.1).cuda()
    ar = torch.arange(B, dtype=torch.int64)
    mk = lambda persist: ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                                    share_fw_bw=True, precision=precision, training=False, out_pads_persist=persist)
    op = mk(True)
    outs = [torch.full((B, J, 2 * d), 7.0, device="cuda") for _ in range(2)]   # garbage the first call must clear
    script = [("fwd", 0), ("fwd", 0), ("plan", 0), ("fwd", 0), ("fwd", 1), ("fwd", 0), ("fwd", 0), ("fwd", 1)]
    for step, (what, which) in enumerate(script):
        lens = torch.randint(0, J + 1, (B,), generator=g)
        if step == 1:
            lens = torch.full((B,), J)
        op.make_plan(lens)
        if what == "plan":
            continue
        op.forward(x, outs[which], kf, bf)
        ref_op = mk(False)
        ref_op.make_plan(lens)
        ref = torch.full((B, J, 2 * d), -3.0, device="cuda")
        ref_op.forward(x, ref, kf, bf)
        torch.cuda.synchronize()
        assert torch.equal(outs[which], ref), "step %d: stale rows survived in the padded part of the output" % step
