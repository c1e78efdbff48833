"""GPU parity of the model.py graph (the soft-attention baselines: multi-layer / multi-modal attention, direct links,
choices and question attention -- model.py:831-983) through fvta_memexqa_amd.model.Model, forward AND backward,
against the CPU oracle (oracle.fvta_fused.model_v1_forward, fp64 autograd).  Tolerance 1e-4 relative fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-4, atol=1e-5, msg=""):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale, err_msg=msg)


def _run(spec, flags, precision="f32", tol=1.0):
    from fvta_memexqa_amd.model import Model
    from fvta_memexqa_amd.synth import make_inputs, make_params_v1, to_dtype
    from oracle import fvta_fused as F
    inputs = make_inputs(spec)
    params = make_params_v1(spec, inputs, use_eu_output=bool(flags.get("use_eu_output")), concat=bool(flags.get("concat")))
    cfg = {**spec.cfg(), "use_question_att": False, "add_tanh": False, **flags}
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    ref = F.model_v1_forward(p64, to_dtype(inputs, torch.float64), cfg)
    ref["loss"].backward()
    model = Model(dict(cfg, batch_size=spec.N, ctx_streams=Model.streams_of(inputs), precision=precision),
                  text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L, want_logits=True)
    d, dp = model.d, model.dp
    unpad = lambda t: torch.cat([t[..., :d], t[..., dp:dp + d]], -1)
    def unpadk(t):                        # [..., k*2dp] feature vectors of k hidden-size pairs (the concat variant)
        return unpad(t.reshape(*t.shape[:-1], -1, 2 * dp)).reshape(*t.shape[:-1], -1)
    _close(unpad(L.hq), ref["hq"], rtol=1e-4 * tol, atol=1e-5 * tol, msg="hq")
    _close(unpad(L.g1s), ref["g1"], rtol=1e-4 * tol, atol=1e-5 * tol, msg="g1 (per-stream vectors)")
    _close(unpadk(L.g1), ref["g1_all"], rtol=1e-4 * tol, atol=1e-5 * tol, msg="g1_all")
    _close(unpadk(L.gq), ref["gq"], rtol=1e-4 * tol, atol=1e-5 * tol, msg="gq")
    _close(unpadk(L.lch), ref["gchoices"], rtol=1e-4 * tol, atol=1e-5 * tol, msg="gchoices")
    if flags.get("use_direct_links"):
        _close(model.att_logits.reshape(ref["att_logits"].shape), ref["att_logits"], atol=2e-5 * tol, msg="att_logits")
    if flags.get("use_mm_att"):
        _close(model.mm_att_logits.reshape(ref["mm_att_logits"].shape), ref["mm_att_logits"], atol=2e-5 * tol, msg="mm_att_logits")
    _close(model.logits, ref["logits"], rtol=1e-4 * tol, atol=1e-5 * tol, msg="logits")
    _close(yp, ref["yp"], rtol=1e-4 * tol, atol=1e-5 * tol, msg="yp")
    _close(model.loss, ref["loss"].reshape(1), rtol=1e-4 * tol, atol=1e-5 * tol, msg="loss")
    if tol == 1.0:
        assert (yp.argmax(1).cpu() == ref["yp"].argmax(1)).all(), "answer argmax must be bit-exact"
    model.backward(L, need_dx=True)
    grads = model.get_oracle_grads()
    seen = 0
    for k, v in p64.items():
        if v.grad is None:          # parameters of a switched-off block
            assert k not in grads or float(np.abs(grads[k]).max()) == 0.0, k
            continue
        if float(v.grad.abs().max()) < 1e-9:
            assert float(np.abs(grads[k]).max()) < 1e-5, k
            continue
        _close(grads[k].reshape(v.grad.shape), v.grad, rtol=2e-4 * tol, atol=2e-5 * tol, msg="grad " + k)
        seen += 1
    assert seen >= 3
    return model, L, ref


ALL = dict(use_ml_att=True, use_mm_att=True, use_direct_links=True, use_choices_att=True, use_question_att=True)


@pytest.mark.parametrize("flags", [
    {},                                                             # the plain LSTM baseline: last states, means
    dict(use_ml_att=True),
    dict(use_mm_att=True),
    dict(use_direct_links=True),
    dict(use_direct_links=True, direct_links_only=True),
    dict(use_direct_links=True, direct_links_only=True, use_question_att=True),
    dict(use_choices_att=True),
    dict(use_question_att=True),
    ALL,
    dict(ALL, use_eu_output=True),
    dict(use_bidirection=True),                                     # only the stack's squash linear (model.py:897)
    dict(use_bidirection=True, use_mm_att=True),
    dict(use_bidirection=True, use_choices_att=True),
    dict(use_bidirection=True, use_question_att=True),
    dict(use_bidirection=True, use_mm_att=True, use_direct_links=True, use_choices_att=True, use_question_att=True),
    dict(use_tgif_ml_att=True),
    dict(use_tgif_ml_att=True, use_mm_att=True, use_direct_links=True, use_question_att=True),
    dict(use_tgif_ml_att=True, use_bidirection=True, use_mm_att=True),
    dict(use_tgif_ml_att=True, concat=True),
    dict(concat=True),
    dict(concat=True, use_ml_att=True, use_choices_att=True, use_eu_output=True),
    dict(concat=True, use_bidirection=True, use_choices_att=True),
], ids=lambda f: "+".join(k[4:] if k.startswith("use_") else k for k in f) or "baseline")
@pytest.mark.parametrize("dense", [False, True])
def test_v1_model_forward_backward(flags, dense):
    from fvta_memexqa_amd.synth import SynthSpec
    spec = SynthSpec(N=3, A=2, P=3, S=1, L=5, d=32, SA=2, dense=dense, simiMatrix=1, text_in=12, img_in=8)
    _run(spec, flags)


@pytest.mark.parametrize("simi", [2, 3])
def test_v1_model_similarity_matrices(simi):
    """simiMatrix 2 in model.py's feature order [(h-q)^2, h*q] (model.py:149), 3; the photo streams keep similarity 1"""
    from fvta_memexqa_amd.synth import SynthSpec
    spec = SynthSpec(N=2, A=2, P=2, S=1, L=4, d=32, SA=1, dense=False, simiMatrix=simi, text_in=12, img_in=8)
    _run(spec, ALL)


def test_v1_model_six_streams_padded_hidden_and_wd():
    """the reference's own six streams (at, ad, when, where, pts, pis) under their checkpoint names, hidden 20 padded to
    32, weight decay on"""
    from fvta_memexqa_amd.model import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params_v1, to_dtype
    from oracle import fvta_fused as F
    spec = SynthSpec(N=2, A=2, P=2, S=1, L=4, d=20, SA=4, dense=False, simiMatrix=2, text_in=12, img_in=8)
    inputs = make_inputs(spec)
    params = make_params_v1(spec, inputs)
    cfg = {**spec.cfg(), "add_tanh": False, "wd": 1e-3, **ALL}
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    ref = F.model_v1_forward(p64, to_dtype(inputs, torch.float64), cfg)
    ref["loss"].backward()
    model = Model(dict(cfg, batch_size=spec.N), text_in=spec.text_in, img_in=spec.img_in)     # default streams
    assert Model.N_ML_W % "pts" in model.params.specs and Model.N_FULL_W in model.params.specs
    assert model.params.specs[Model.N_ML_W % "pis"] == (3 * model.wp,) and model.params.specs[Model.N_ML_W % "at"] == (2 * model.wp,)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L)
    _close(yp, ref["yp"], msg="yp")
    _close(model.loss, ref["loss"].reshape(1), msg="loss with l2 terms")
    model.backward(L)
    grads = model.get_oracle_grads()
    for k, v in p64.items():
        if v.grad is None:              # parameters of blocks that are switched off (tgif, squash, concat linears)
            continue
        _close(grads[k].reshape(v.grad.shape), v.grad, rtol=2e-4, atol=2e-5, msg="grad " + k)


def test_v1_model_unbuilt_switches_raise():
    from fvta_memexqa_amd.model import Model
    # flag sets the reference's own graph construction rejects (shape mismatches / an undefined name)
    for flags in (dict(use_bidirection=True, use_ml_att=True), dict(concat=True, use_question_att=True),
                  dict(concat=True, use_direct_links=True)):
        with pytest.raises(ValueError):
            Model(dict(flags, hidden_size=32))
    with pytest.raises(ValueError):
        Model({"simiMatrix": 4, "hidden_size": 32})


def test_v1_model_bf16_encoders():
    """the bf16 LSTM engine under the baselines' attention block: relative (Frobenius) error against the fp64 oracle"""
    from fvta_memexqa_amd.model import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params_v1, to_dtype
    from oracle import fvta_fused as F
    spec = SynthSpec(N=4, A=2, P=3, S=1, L=6, d=64, SA=2, dense=False, simiMatrix=1, text_in=16, img_in=8)
    inputs = make_inputs(spec)
    params = make_params_v1(spec, inputs)
    cfg = {**spec.cfg(), "add_tanh": False, **ALL}
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    ref = F.model_v1_forward(p64, to_dtype(inputs, torch.float64), cfg)
    ref["loss"].backward()
    model = Model(dict(cfg, batch_size=spec.N, ctx_streams=Model.streams_of(inputs), precision="bf16"),
                  text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L)
    rel = lambda a, b: float((torch.as_tensor(a).double().cpu().reshape(-1) - b.reshape(-1)).norm() / (b.norm() + 1e-30))
    assert rel(yp, ref["yp"].detach()) < 5e-3
    assert rel(model.loss, ref["loss"].detach()) < 5e-3
    model.backward(L)
    grads = model.get_oracle_grads()
    for k, v in p64.items():
        if v.grad is None or float(v.grad.abs().max()) < 1e-9:
            continue
        # (gradients that are differences of nearly cancelling terms -- norms of 1e-5 -- carry the bf16 rounding of the
        # encoder outputs at full size)
        assert rel(grads[k], v.grad) < (3e-2 if float(v.grad.norm()) > 1e-3 else 0.15), (k, rel(grads[k], v.grad))


def test_v1_trainer_step():
    from fvta_memexqa_amd.model import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params_v1
    from fvta_memexqa_amd.trainer import Trainer
    spec = SynthSpec(N=4, A=1, P=3, S=1, L=5, d=32, SA=2, dense=False, text_in=12, img_in=8)
    inputs = make_inputs(spec)
    model = Model({**spec.cfg(), "add_tanh": False, "batch_size": spec.N, "ctx_streams": Model.streams_of(inputs), **ALL},
                  text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(make_params_v1(spec, inputs))
    tr = Trainer(model, dict(init_lr=0.5))
    losses = [tr.step(None, (None, dict(inputs, num_examples=4)))[0] for _ in range(8)]
    assert losses[-1] < losses[0], losses
