"""GPU parity at BASELINE.json's named configurations, through the Model mirror and the C ABI:

  * configs[2] (the metric shape, bf16 engine): train-step GRADIENTS vs the fp64 oracle on a 2-QA-pair batch, and the
    N = 64 dense batch bench.py times -- finite loss, answer argmax equal to the exact-fp32 engine, bitwise
    reproducible;
  * configs[4] (long album: 120 photos x 6 streams x 60 tokens, hidden 1024 -> T = 7200, w = 2048, JQ = 60): forward
    vs the fp32 CPU oracle on one QA pair, forward+backward reproducibility at N = 2 in both precisions;
  * the short last batch of an epoch (model_v2.py:1270: padded rows, labels all False) with TF-1's softmax-xent
    gradient (softmax - labels on every row).

Tolerances: fp32 engine 1e-4 relative (north_star); bf16 engine 3e-2 absolute on activations / 4e-2 relative L2 on
gradients (operands carry 8 significant bits)."""
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
    a = torch.as_tensor(np.asarray(a)).double().reshape(-1)
    b = torch.as_tensor(np.asarray(b)).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _param_slices(model):
    """name -> (offset, shape) of every parameter in the model's flat buffers"""
    return {name: (model.params.offsets[name], model.params.specs[name]) for name in model.params.specs}


def _subset(inputs, n):
    return dict(ctx=[dict(x=s["x"][:n], mask=s["mask"][:n], cell=s["cell"]) for s in inputs["ctx"]],
                q=dict(x=inputs["q"]["x"][:n], mask=inputs["q"]["mask"][:n]),
                choices=dict(x=inputs["choices"]["x"][:n], mask=inputs["choices"]["mask"][:n]), y=inputs["y"][:n])


# ------------------------------------------------------------------ configs[2]: metric shape, bf16 train step
@pytest.mark.parametrize("dense", [True, False])
def test_metric_shape_bf16_train_step_gradients_vs_oracle(dense):
    """BASELINE.json configs[2] at N = 2 (the path is batch-independent; gradients sum over the batch): every
    parameter gradient of the bf16 engine against autograd of the fp64 oracle, relative L2 per parameter."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs, make_params, to_dtype
    from oracle import fvta_fused as F
    spec = SynthSpec(dense=dense, **dict(CONFIGS["metric"], N=2))
    params, inputs = make_params(spec), make_inputs(spec)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    ref = F.fvta_forward(p64, to_dtype(inputs, torch.float64), spec.cfg())
    ref["loss"].backward()
    model = Model(dict(spec.cfg(), batch_size=spec.N, precision="bf16"), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L)
    model.backward(L, need_dx=True)
    _close(yp, ref["yp"], rtol=0, atol=3e-2, msg="yp (bf16 engine)")
    _close(model.loss, ref["loss"].reshape(1), rtol=0, atol=3e-2, msg="loss (bf16 engine)")
    grads = model.get_oracle_grads()
    worst = {}
    for k, v in p64.items():
        if v.grad is None or float(v.grad.norm()) < 1e-9:     # out_b: sum_c (softmax - y) = 0 when every row has a label
            assert v.grad is None or float(np.abs(grads[k]).max()) < 1e-5, k
            continue
        worst[k] = _rel_l2(grads[k].reshape(v.grad.shape), v.grad)
    assert worst and max(worst.values()) < 4e-2, "relative L2 gradient error per parameter: %r" % worst


def test_metric_shape_n64_dense_is_finite_reproducible_and_matches_f32_engine_argmax():
    """The exact tensor bench.py times (N = 64 dense, train step): finite loss and gradients, bitwise equal across
    repeats, answer argmax of the bf16 engine equal to the exact-fp32 engine's wherever the fp32 margin between the
    two best answers exceeds the bf16 tolerance (the reference's bit-exactness claim is for its own fp32 path), and
    yp within the bf16 tolerance of the fp32 engine everywhere."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs, make_params
    spec = SynthSpec(dense=True, **CONFIGS["metric"])
    params, inputs = make_params(spec), make_inputs(spec)
    out = {}
    for prec in ("bf16", "f32"):
        model = Model(dict(spec.cfg(), batch_size=spec.N, precision=prec), text_in=spec.text_in, img_in=spec.img_in)
        model.set_oracle_params(params)
        L = model.load_inputs(inputs, training=True)
        runs = []
        for _ in range(2):
            model.zero_grad()
            yp = model.forward(L)
            model.backward(L, need_dx=True)
            torch.cuda.synchronize()
            runs.append((yp.clone(), model.loss.clone(), model.params.grad.clone(), L.g1.clone()))
        for a, b, name in zip(runs[0], runs[1], ("yp", "loss", "grad", "g1")):
            assert torch.equal(a, b), "%s engine: %s differs between two runs on the same batch" % (prec, name)
        assert torch.isfinite(runs[0][1]).all() and torch.isfinite(runs[0][2]).all() and torch.isfinite(runs[0][0]).all()
        assert float(runs[0][2].abs().max()) > 0
        slices = {name: (off, off + int(np.prod(shape))) for name, (off, shape) in _param_slices(model).items()}
        out[prec] = (runs[0][0].cpu().double(), float(runs[0][1]), runs[0][2].cpu().double(), slices)
        # the attention logits' own parameters, on THIS engine's rows and routing (tests/_attn_rows_check.py): tight
        from tests._attn_rows_check import attention_param_grads_on_engine_rows
        own = attention_param_grads_on_engine_rows(model, L)
        assert max(own.values()) < 2e-3, "%s engine: att_logits gradients vs fp64 autograd on the engine's own rows: %r" % (prec, own)
        del model, L, runs
        torch.cuda.empty_cache()
    yb, yf = out["bf16"][0], out["f32"][0]
    # the flat gradient of the timed regime -- more than 8192 active rows per step, i.e. the tiled backward step kernel
    # `lstm_bwd_fused_bf16`, which no oracle-level test reaches -- against the exact-fp32 engine, per parameter slice
    gb, gf, slices = out["bf16"][2], out["f32"][2], out["f32"][3]
    assert out["bf16"][3] == slices
    worst = {}
    for name, (lo, hi) in slices.items():
        if float(gf[lo:hi].norm()) < 1e-9:
            assert float(gb[lo:hi].abs().max()) < 1e-5, name
            continue
        worst[name] = _rel_l2(gb[lo:hi], gf[lo:hi])
    # (the attention logits' own parameters hang off the arg-max positions -- max over j, max over t: model_v2.py:268, 278 --
    #  which move where the two engines' context rows, 3e-3 apart, meet a near-tie: a discontinuous routing, not an error
    #  of the kernels; ACROSS engines they get the looser bound, the bi-LSTM and scorer slices the engine's 4e-2.  Each
    #  engine's att_logits gradients were held to 2e-3 above, against fp64 autograd on that engine's own rows and routing)
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


# ------------------------------------------------------------------ configs[4]: long album
def _long_spec(N, dense):
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec
    return SynthSpec(dense=dense, **dict(CONFIGS["long_album"], N=N))


def test_long_album_forward_vs_oracle_one_pair():
    """configs[4] at full length (T = 7200, w = 2048, JQ = 60 -- the shapes that leave the 16-row attention kernel),
    exact-fp32 engine, N = 2 on the GPU; the fp32 CPU oracle checks QA pair 0."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import make_inputs, make_params
    from oracle import fvta_fused as F
    spec = _long_spec(2, dense=False)
    params, inputs = make_params(spec), make_inputs(spec)
    model = Model(dict(spec.cfg(), batch_size=spec.N), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs)
    assert (L.T, L.JQ, model.wp, L.K) == (7200, 60, 2048, 7)
    yp = model.forward(L)
    with torch.no_grad():
        ref = F.fvta_forward(params, _subset(inputs, 1), spec.cfg())
    d, dp = model.d, model.dp
    unpad = lambda t: torch.cat([t[..., :d], t[..., dp:dp + d]], -1)
    _close(unpad(L.hall[:1]).reshape(ref["hall"].shape), ref["hall"], rtol=1e-4, atol=2e-5, msg="hall")
    _close(unpad(L.g1[:1]), ref["g1_all"], rtol=1e-4, atol=2e-5, msg="g1")
    _close(unpad(L.gq[:1]), ref["gq"], rtol=1e-4, atol=2e-5, msg="gq")
    _close(yp[:1], ref["yp"], rtol=1e-4, atol=1e-5, msg="yp")
    assert (yp[:1].argmax(1).cpu() == ref["yp"].argmax(1)).all()


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_long_album_forward_backward_reproducible(precision):
    """configs[4], N = 2, dense: forward + backward twice on the same batch -- bitwise equal context tensor, g1, yp and
    flat gradient; finite; the bf16 engine within its tolerance of the fp32 engine's yp is covered by the metric test."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import make_inputs, make_params
    spec = _long_spec(2, dense=True)
    params, inputs = make_params(spec), make_inputs(spec)
    model = Model(dict(spec.cfg(), batch_size=spec.N, precision=precision), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    ref = None
    for _ in range(2):
        model.zero_grad()
        yp = model.forward(L)
        model.backward(L, need_dx=True)
        torch.cuda.synchronize()
        cur = (L.arena.clone(), L.g1.clone(), yp.clone(), model.params.grad.clone())
        if ref is None:
            ref = cur
        else:
            for a, b, name in zip(cur, ref, ("arena", "g1", "yp", "grad")):
                assert torch.equal(a, b), name
    assert torch.isfinite(ref[3]).all() and torch.isfinite(ref[2]).all() and float(ref[3].abs().max()) > 0


# ------------------------------------------------------------------ short last batch (padded rows)
def _pad_rows(inputs, n_real):
    """what get_feed_dict leaves for rows >= num_examples (model_v2.py:1171-1270): zero ids -> here zero encoder
    inputs, all-False masks, all-False labels"""
    def z(st):
        st["x"][n_real:] = 0
        st["mask"][n_real:] = False
    for st in inputs["ctx"]:
        z(st)
    z(inputs["q"])
    z(inputs["choices"])
    inputs["y"][n_real:] = False
    return inputs


@pytest.mark.parametrize("tf_grad", [True, False])
def test_short_batch_training_step_with_tf_xent_gradient(tf_grad):
    """num_examples < N: the padded rows are all-masked everywhere and their labels are all False.  TF-1's
    softmax-xent gradient still sends softmax/N from them into the scorer (the bias gradient grows by
    (#padded)/N); every other path out of a padded row ends in zero context vectors or past-length LSTM outputs.
    All gradients vs autograd of the fp64 oracle with the matching custom gradient, and one Adadelta update."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params, to_dtype
    from fvta_memexqa_amd.trainer import Trainer
    from oracle import fvta_fused as F
    from oracle import fvta_literal as Lit
    spec = SynthSpec(N=5, A=2, P=3, S=2, L=5, d=32, SA=1, dense=False, text_in=12, img_in=8)
    n_real = 3
    params, inputs = make_params(spec), _pad_rows(make_inputs(spec), n_real)
    params["out_b"] = torch.tensor([0.3])
    cfg = dict(spec.cfg(), tf_xent_grad=tf_grad)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    ref = F.fvta_forward(p64, to_dtype(inputs, torch.float64), cfg)
    ref["loss"].backward()
    # hand value of the one gradient the padded rows change: d out_b = sum over rows of sum_c (softmax - y)_c / N
    yp_ref = ref["yp"].detach()
    db_rows = (yp_ref.sum(1) - inputs["y"].double().sum(1)) / spec.N
    exp_db = float(db_rows.sum()) if tf_grad else float(db_rows[:n_real].sum())
    np.testing.assert_allclose(float(p64["out_b"].grad), exp_db, rtol=1e-9, atol=1e-12)
    assert abs(float(db_rows[n_real:].sum()) - (spec.N - n_real) / spec.N) < 1e-9

    model = Model(dict(cfg, batch_size=spec.N, init_lr=0.5), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L)
    model.backward(L, need_dx=True)
    _close(yp, ref["yp"], msg="yp")
    _close(model.loss, ref["loss"].reshape(1), msg="loss")
    grads = model.get_oracle_grads()
    for k, v in p64.items():
        if v.grad is not None:
            _close(grads[k].reshape(v.grad.shape), v.grad, rtol=2e-4, atol=2e-5, msg="grad " + k)
    np.testing.assert_allclose(float(grads["out_b"].reshape(-1)[0]), exp_db, rtol=2e-4, atol=1e-6)
    # the same through Trainer.step (reference entry, trainer.py:30-40) and one Adadelta update
    model.set_oracle_params(params)
    loss, _, _ = Trainer(model, dict(init_lr=0.5)).step(None, (None, dict(inputs, num_examples=n_real)))
    _close(torch.tensor([loss]), ref["loss"].reshape(1))
    new = model.get_weights()
    for k, name in (("out_b", Model.N_OUT_B), ("out_W", Model.N_OUT_W), ("text_kernel", Model.N_TEXT_K % "fw")):
        v = p64[k]
        exp, _, _ = Lit.adadelta_step(v.detach().numpy(), v.grad.numpy(), np.zeros_like(v.detach().numpy()),
                                      np.zeros_like(v.detach().numpy()), 0.5)
        _close(new[name].reshape(exp.shape), exp, rtol=2e-4, atol=2e-5, msg="updated " + k)
