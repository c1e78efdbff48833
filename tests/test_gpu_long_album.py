"""GPU gradient parity at BASELINE.json configs[4] (long album: 120 photos x 6 streams x 60 tokens, hidden 1024 ->
T = 7200, w = 2048, JQ = 60, K = 7) and at hidden 1024 generally -- the kernel instantiations that configuration
selects (`attn_fwd_main` / `attn_bwd_main` at w = 2048 / JQ = 60, the LSTM engines at d = 1024, J = 60):

  * one ragged long-album QA pair, forward + backward: every parameter gradient and the encoder-input gradients
    against autograd of the fp64 oracle (reference: model_v2.py:210-298 attention_3d, 652-833 encoders, 1029-1096
    scorer / loss) -- fp32 and split-bf16 (bf16x3) engines at the 1e-4 class, bf16 engine at relative L2 4e-2;
  * op level: the bi-LSTM backward at (B >= 300, J = 60, in = 200, d = 1024), both engines; attention_3d backward at
    (N = 1, K = 7, T = 7200, JQ = 60, w = 2048), masked;
  * the N = 32 dense long-album train step (the configuration's own batch): finite, bitwise reproducible, bf16 answer
    arg-max equal to the fp32 engine's wherever the fp32 margin exceeds the bf16 tolerance.

Tolerances: fp32 engine rtol 2e-4 / atol 2e-5 x max|ref| (1e-4 class: sums of ~1e5 fp32 products); bf16 engine 3e-2
absolute on activations, 4e-2 relative L2 on gradient tensors, 1e-1 relative on one-element gradients (operands carry 8
significant bits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-4, atol=1e-5, msg=""):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale, err_msg=msg)


def _rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a)).double().reshape(-1)
    b = torch.as_tensor(np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b)).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _long_spec(N, dense):
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec
    return SynthSpec(dense=dense, **dict(CONFIGS["long_album"], N=N))


# ------------------------------------------------------------------ (i) one ragged long-album pair, every gradient
_ONE_PAIR = {}


def _one_pair_oracle():
    """the fp64 oracle's forward + autograd on the ragged long-album pair (minutes of CPU work: computed once, shared by
    the two engines' tests)"""
    if not _ONE_PAIR:
        from fvta_memexqa_amd.synth import make_inputs, make_params, to_dtype
        from oracle import fvta_fused as F
        spec = _long_spec(1, dense=False)
        params, inputs = make_params(spec), make_inputs(spec)
        p64 = {k: v.double().requires_grad_() for k, v in params.items()}
        in64 = to_dtype(inputs, torch.float64)
        for st in in64["ctx"] + [in64["q"], in64["choices"]]:
            st["x"].requires_grad_()
        ref = F.fvta_forward(p64, in64, spec.cfg())
        ref["loss"].backward()
        _ONE_PAIR.update(spec=spec, params=params, inputs=inputs, yp=ref["yp"].detach(), loss=ref["loss"].detach(),
                         grads={k: v.grad for k, v in p64.items()},
                         dx=[st["x"].grad for st in in64["ctx"]] + [in64["q"]["x"].grad, in64["choices"]["x"].grad])
    return _ONE_PAIR


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16"])
def test_long_album_one_pair_gradients_vs_oracle(precision):
    from fvta_memexqa_amd.model_v2 import Model
    R = _one_pair_oracle()
    spec, params, inputs = R["spec"], R["params"], R["inputs"]
    model = Model(dict(spec.cfg(), batch_size=spec.N, precision=precision), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    assert (L.T, L.JQ, model.wp, L.K) == (7200, 60, 2048, 7)
    model.zero_grad()
    yp = model.forward(L)
    model.backward(L, need_dx=True)
    grads = model.get_oracle_grads()
    dxs = model.get_input_grads(L)
    assert len(dxs) == len(R["dx"])
    if precision in ("f32", "bf16x3"):
        # (the split-bf16 engine carries 16-17 significant bits per MFMA operand: a little wider absolute bounds, same class)
        wa, wg = (1e-5, 2e-5) if precision == "f32" else (3e-5, 5e-5)
        _close(yp, R["yp"], rtol=1e-4, atol=wa, msg="yp")
        assert (yp.argmax(1).cpu() == R["yp"].argmax(1)).all()
        _close(model.loss, R["loss"].reshape(1), rtol=1e-4, atol=wa, msg="loss")
        for k, g in R["grads"].items():
            if g is not None:
                _close(grads[k].reshape(g.shape), g, rtol=2e-4, atol=wg, msg="grad " + k)
        for i, (a, b) in enumerate(zip(dxs, R["dx"])):
            _close(a.reshape(b.shape), b, rtol=2e-4, atol=wg, msg="dx stream %d" % i)
    else:
        _close(yp, R["yp"], rtol=0, atol=3e-2, msg="yp (bf16 engine)")
        worst, worst_scalar = {}, {}
        for k, g in R["grads"].items():
            if g is None or float(g.norm()) < 1e-9:
                continue
            # a one-element gradient (the attention logit biases) is a single sum of signed terms: its relative error
            # has no averaging over elements -- bounded at 1e-1, the tensors at 4e-2 relative L2
            (worst_scalar if g.numel() == 1 else worst)[k] = _rel_l2(grads[k].reshape(g.shape), g)
        for i, (a, b) in enumerate(zip(dxs, R["dx"])):
            if float(b.norm()) > 1e-9:
                worst["dx%d" % i] = _rel_l2(a.reshape(b.shape), b)
        assert worst and max(worst.values()) < 4e-2, "relative L2 gradient error: %r" % worst
        assert all(v < 1e-1 for v in worst_scalar.values()), "scalar gradients: %r" % worst_scalar


# ------------------------------------------------------------------ (ii) bi-LSTM backward at d = 1024, J = 60
@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("B,dense", [(320, True), (333, False)])
def test_bilstm_backward_hidden_1024(precision, B, dense):
    from fvta_memexqa_amd import ops
    from fvta_memexqa_amd._lib import BF16, F32
    from oracle import fvta_fused as F
    J, din, d = 60, 200, 1024
    g = torch.Generator().manual_seed(B + J + d + 3)
    x = torch.randn(B, J, din, generator=g)
    lens = torch.full((B,), J) if dense else torch.randint(0, J + 1, (B,), generator=g)
    lim = (6.0 / (din + d + 4 * d)) ** 0.5
    k_fw = (torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim * 2
    b_fw = torch.randn(4 * d, generator=g) * 0.1
    mask = torch.arange(J)[None, :] < lens[:, None]
    g_out = torch.randn(B, J, 2 * d, generator=g) * mask[:, :, None]
    g_last = torch.randn(B, 2 * d, generator=g)
    leaves = [t.double().requires_grad_() for t in (x, k_fw, b_fw)]
    ref_out, ref_last = F.encode_stream(leaves[0], mask, leaves[1], leaves[2])
    ((ref_out * g_out.double()).sum() + (ref_last * g_last.double()).sum()).backward()
    cu = lambda t: t.cuda().contiguous()
    xc, kf, bf = cu(x), cu(k_fw), cu(b_fw)
    out, last, op = ops.bilstm_simple(xc, lens, kf, bf, None, None, training=True,
                                      precision=BF16 if precision == "bf16" else F32)
    d_out = cu(g_out).clone()
    op.last_state_bwd(cu(g_last), 0, B, d_out)
    dx = torch.zeros_like(xc)
    dkf, dbf = torch.zeros_like(kf), torch.zeros_like(bf)
    op.backward(xc, out, d_out, kf, None, dx, dkf, dbf, None, None)
    if precision == "f32":
        _close(out, ref_out, rtol=1e-4, atol=1e-5, msg="out")
        _close(last, ref_last, rtol=1e-4, atol=1e-5, msg="last")
        _close(dx, leaves[0].grad, rtol=2e-4, atol=2e-5, msg="dx")
        _close(dkf, leaves[1].grad, rtol=2e-4, atol=2e-5, msg="dkernel")
        _close(dbf, leaves[2].grad, rtol=2e-4, atol=2e-5, msg="dbias")
    else:
        _close(out, ref_out, rtol=0, atol=3e-2, msg="out")
        _close(last, ref_last, rtol=0, atol=3e-2, msg="last")
        for name, a, b in (("dx", dx, leaves[0].grad), ("dkernel", dkf, leaves[1].grad), ("dbias", dbf, leaves[2].grad)):
            err = _rel_l2(a, b)
            assert err < 4e-2, "%s: relative L2 error %.4f" % (name, err)


# ------------------------------------------------------------------ (iii) attention_3d backward at the long-album shape
@pytest.mark.parametrize("simi,tanh", [(2, True), (1, False)])
def test_attention_3d_backward_long_album_shape(simi, tanh):
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    from tests.test_gpu_forward import _att_case
    N, K, T, JQ, w = 1, 7, 7200, 60, 2048
    h, q, W, b, hm, qm = _att_case(N, K, T, JQ, w, simi, tanh, True, seed=7200 + simi)
    hm[0, 0, :3] = True           # every (n,k) keeps a valid row (the fully masked deviation is covered separately)
    g = torch.Generator().manual_seed(98)
    gout = torch.randn(N, w, generator=g)
    hd, qd = h.double().requires_grad_(), q.double().requires_grad_()
    Wd, bd = W.double().requires_grad_(), b.double().requires_grad_()
    ref_ha, _ = F.attention_3d(hd, qd, Wd, bd, hm, qm, simiMatrix=simi, add_tanh=tanh)
    (ref_ha * gout.double()).sum().backward()
    op = ops.FocalAttention(N, K, T, JQ, w, simi, tanh)
    cu = lambda t: t.cuda().contiguous()
    hc, qc, Wc, bc = cu(h), cu(q), cu(W.reshape(-1)), cu(b)
    hmc, qmc = cu(ops.as_mask_u8(hm)), cu(ops.as_mask_u8(qm))
    ha, _ = op.forward(hc, qc, hmc, qmc, Wc, bc)
    _close(ha, ref_ha, msg="h_a")
    dh, dq = torch.full_like(hc, 7.0), torch.full_like(qc, 7.0)
    dW, db = torch.zeros_like(Wc), torch.zeros(1, device="cuda")
    op.backward(hc, qc, hmc, qmc, Wc, bc, cu(gout), dh, dq, dW, db, accumulate=False)
    _close(dh, hd.grad, msg="d_hinfo")
    _close(dq, qd.grad, msg="d_hq")
    _close(dW, Wd.grad.reshape(-1), msg="dW")
    _close(db, bd.grad, msg="db")


# ------------------------------------------------------------------ (iv) the configuration's own batch: N = 32 dense
def test_long_album_n32_dense_train_step_reproducible_and_argmax_matches_f32():
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import make_inputs, make_params
    spec = _long_spec(32, dense=True)
    params, inputs = make_params(spec), make_inputs(spec)
    out = {}
    for prec in ("bf16", "f32"):
        model = Model(dict(spec.cfg(), batch_size=spec.N, precision=prec), text_in=spec.text_in, img_in=spec.img_in)
        model.set_oracle_params(params)
        L = model.load_inputs(inputs, training=True)
        assert (L.T, L.JQ, model.wp, L.K) == (7200, 60, 2048, 7)
        runs = []
        for _ in range(2):
            model.zero_grad()
            yp = model.forward(L)
            model.backward(L, need_dx=False)
            torch.cuda.synchronize()
            runs.append((yp.clone(), model.loss.clone(), model.params.grad.clone(), L.g1.clone()))
        for a, b, name in zip(runs[0], runs[1], ("yp", "loss", "grad", "g1")):
            assert torch.equal(a, b), "%s engine: %s differs between two runs on the same batch" % (prec, name)
        assert torch.isfinite(runs[0][0]).all() and torch.isfinite(runs[0][1]).all() and torch.isfinite(runs[0][2]).all()
        assert float(runs[0][2].abs().max()) > 0
        slices = {n: (model.params.offsets[n], model.params.offsets[n] + int(np.prod(model.params.specs[n]))) for n in model.params.specs}
        out[prec] = (runs[0][0].cpu().double(), float(runs[0][1]), runs[0][2].cpu().double(), slices)
        del model, L, runs
        torch.cuda.empty_cache()
    yb, yf = out["bf16"][0], out["f32"][0]
    # the bf16 engine's flat gradient at the configuration's own batch (the step kernels of this regime: d = 1024,
    # 23,040 rows per step) against the exact-fp32 engine's, relative L2 per parameter slice
    gb, gf, slices = out["bf16"][2], out["f32"][2], out["f32"][3]
    worst = {}
    for name, (lo, hi) in slices.items():
        if float(gf[lo:hi].norm()) < 1e-9:
            assert float(gb[lo:hi].abs().max()) < 1e-5, name
            continue
        worst[name] = _rel_l2(gb[lo:hi], gf[lo:hi])
    # (the attention logits' own parameters hang off the arg-max positions -- max over j, max over t: model_v2.py:268, 278 --
    #  which move where the two engines' context rows, 3e-3 apart, meet a near-tie: a discontinuous routing, not an error
    #  of the kernels; they get the looser bound, the bi-LSTM and scorer slices the engine's 4e-2)
    loose = {k: v for k, v in worst.items() if "att_logits" in k}
    tight = {k: v for k, v in worst.items() if "att_logits" not in k}
    assert tight and max(tight.values()) < 4e-2, "bf16 vs f32 engine, relative L2 per parameter slice: %r" % worst
    assert not loose or max(loose.values()) < 0.2, "bf16 vs f32 engine, attention parameters: %r" % loose
    assert float((yb - yf).abs().max()) < 3e-2
    assert abs(out["bf16"][1] - out["f32"][1]) < 3e-2
    top2 = yf.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 6e-2
    assert decided.any()
    assert (yb.argmax(1)[decided] == yf.argmax(1)[decided]).all()
