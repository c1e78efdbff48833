"""The two independent CPU restatements (literal NumPy vs fused torch) agree."""
import numpy as np
import pytest
import torch

from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params, to_dtype, to_numpy
from oracle import fvta_fused as F
from oracle import fvta_literal as L


def _run_both(spec, cfg_over=None, extra_params=None):
    params = to_dtype(make_params(spec), torch.float64)
    inputs = to_dtype(make_inputs(spec), torch.float64)
    if extra_params:
        params.update(extra_params)
    cfg = spec.cfg()
    cfg.update(cfg_over or {})
    of = F.fvta_forward(params, inputs, cfg)
    ol = L.fvta_forward(to_numpy(params), to_numpy(inputs), cfg)
    return of, ol


@pytest.mark.parametrize("simi,tanh,qatt", [(1, False, False), (2, True, True), (3, True, True), (4, False, True)])
@pytest.mark.parametrize("dense", [True, False])
def test_forward_agree_fp64(simi, tanh, qatt, dense):
    spec = SynthSpec(N=3, A=2, P=3, S=2, L=5, d=8, SA=1, dense=dense, simiMatrix=simi, add_tanh=tanh,
                     use_question_att=qatt, text_in=12, img_in=8)
    of, ol = _run_both(spec)
    for k in ["hq", "lq", "lchoices", "hall", "g1_all", "att_logits", "gq", "logits", "yp", "loss"]:
        a = of[k].detach().numpy()
        b = np.asarray(ol[k])
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9, err_msg=k)
    assert (of["yp"].argmax(1).numpy() == ol["yp"].argmax(1)).all()


def test_forward_agree_unshared_bw():
    spec = SynthSpec(N=2, A=1, P=3, S=1, L=4, d=8, dense=False, share_fw_bw=False, text_in=12, img_in=8)
    of, ol = _run_both(spec)
    np.testing.assert_allclose(of["loss"].item(), ol["loss"], rtol=1e-10)
    np.testing.assert_allclose(of["hall"].numpy(), ol["hall"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("warp_type", [1, 2, 3, 4, 5])
def test_time_warp_closed_form_matches_literal(warp_type):
    spec = SynthSpec(N=2, A=1, P=2, S=1, L=3, d=4, dense=False, text_in=8, img_in=8)
    g = torch.Generator().manual_seed(7)
    w = spec.w
    extra = dict(WH_W=torch.randn(2 * w, w, generator=g, dtype=torch.float64) * 0.1,
                 WH_b=torch.randn(w, generator=g, dtype=torch.float64) * 0.1,
                 WC_W=torch.randn(w, 1, generator=g, dtype=torch.float64) * 0.1,
                 WC_b=torch.randn(1, generator=g, dtype=torch.float64) * 0.1,
                 window_t=1.3)
    of, ol = _run_both(spec, dict(use_time_warp=True, warp_type=warp_type), extra)
    np.testing.assert_allclose(of["hall"].numpy(), ol["hall"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(of["yp"].numpy(), ol["yp"], rtol=1e-9)


@pytest.mark.parametrize("warp_type", [1, 3, 5])
def test_time_warp_att_literal_matches_fused(warp_type):
    """use_time_warp_att (model_v2.py:269-275, 1020) through both whole-path restatements, ragged lengths (masked rows
    whose scale is negative take the softmax over t: the scale is applied AFTER exp_mask)."""
    spec = SynthSpec(N=3, A=1, P=3, S=2, L=3, d=4, dense=False, text_in=8, img_in=8)
    g = torch.Generator().manual_seed(11)
    w = spec.w
    extra = dict(WH_W=torch.randn(2 * w, w, generator=g, dtype=torch.float64) * 0.3,
                 WH_b=torch.randn(w, generator=g, dtype=torch.float64) * 0.3,
                 WC_W=torch.randn(w, 1, generator=g, dtype=torch.float64) * 0.5,
                 WC_b=torch.randn(1, generator=g, dtype=torch.float64) * 0.1,
                 window_t=1.3)
    of, ol = _run_both(spec, dict(use_time_warp=True, use_time_warp_att=True, warp_type=warp_type), extra)
    assert (of["c_warp"] < 0).any() and (of["c_warp"] > 0).any()         # both signs of the scale occur
    np.testing.assert_allclose(of["g1_all"].numpy(), ol["g1_all"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(of["yp"].numpy(), ol["yp"], rtol=1e-9)
    # and it differs from the run without it
    of0, _ = _run_both(spec, dict(use_time_warp=True, warp_type=warp_type), extra)
    assert np.abs(of0["g1_all"].numpy() - of["g1_all"].numpy()).max() > 1e-6


def test_plumbing_config_fp32_agree():
    """BASELINE.json configs[0] shape, fp32 both sides: <=1e-5."""
    from fvta_memexqa_amd.synth import CONFIGS
    spec = SynthSpec(dense=False, **CONFIGS["plumbing"])
    params, inputs = make_params(spec), make_inputs(spec)
    of = F.fvta_forward(params, inputs, spec.cfg())
    ol = L.fvta_forward(to_numpy(params), to_numpy(inputs), spec.cfg())
    np.testing.assert_allclose(of["yp"].numpy(), ol["yp"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(of["g1_all"].numpy(), ol["g1_all"], rtol=1e-4, atol=1e-5)
    assert (of["yp"].argmax(1).numpy() == ol["yp"].argmax(1)).all()


def test_attention_v1_feature_order():
    """model.py:149 orders simiMatrix-2 features [(h-q)^2, h*q]; model_v2.py:245 the reverse."""
    g = torch.Generator().manual_seed(3)
    h = torch.randn(2, 5, 6, generator=g, dtype=torch.float64)
    q = torch.randn(2, 3, 6, generator=g, dtype=torch.float64)
    W = torch.randn(12, 1, generator=g, dtype=torch.float64)
    b = torch.randn(1, generator=g, dtype=torch.float64)
    hm = torch.rand(2, 5, generator=g) > 0.3
    qm = torch.rand(2, 3, generator=g) > 0.3
    for order in ("v1", "v2"):
        ha, a = F.attention(h, q, W, b, hm, qm, simiMatrix=2, feat_order=order, bidirect=True)
        hb, bl = L.attention(h.numpy(), q.numpy(), W.numpy(), b.numpy(), hm.numpy(), qm.numpy(), simiMatrix=2,
                             feat_order=order, bidirect=True)
        np.testing.assert_allclose(ha.numpy(), hb, rtol=1e-10)
        np.testing.assert_allclose(a.numpy(), bl, rtol=1e-10)
    # unmasked call (model.py:850-851 style: masks not both given)
    ha, _ = F.attention(h, q, W, b, hm, None, simiMatrix=2)
    hb, _ = L.attention(h.numpy(), q.numpy(), W.numpy(), b.numpy(), hm.numpy(), None, simiMatrix=2)
    np.testing.assert_allclose(ha.numpy(), hb, rtol=1e-10)


def test_attention_gru_cell_agree():
    g = torch.Generator().manual_seed(5)
    d, B = 6, 4
    args = [torch.randn(B, d + 1, generator=g, dtype=torch.float64), torch.randn(B, d, generator=g, dtype=torch.float64),
            torch.randn(2 * d, d, generator=g, dtype=torch.float64), torch.randn(d, generator=g, dtype=torch.float64),
            torch.randn(d, d, generator=g, dtype=torch.float64), torch.randn(d, d, generator=g, dtype=torch.float64),
            torch.randn(d, generator=g, dtype=torch.float64)]
    a = F.attention_gru_cell(*args).numpy()
    b = L.attention_gru_cell(*[x.numpy() for x in args])
    np.testing.assert_allclose(a, b, rtol=1e-12)


@pytest.mark.parametrize("W,cd,cw,wd", [(16, 8, 100, 100), (7, 3, 5, 11)])
def test_embedding_front_end_agree(W, cd, cw, wd):
    """model_v2.py:52-70, 524-645: literal sliding-window conv1d vs the unfold+matmul restatement, fp64."""
    g = np.random.default_rng(W + cw)
    VW, VF, VC = 9, 6, 13
    we, fe, ce = g.normal(size=(VW, wd)), g.normal(size=(VF, wd)), g.normal(size=(VC, cd))
    fl, b = g.normal(size=(1, 5, cd, cw)) * 0.3, g.normal(size=cw) * 0.1
    ids = g.integers(0, VW + VF, size=(2, 3, 4))
    ch = g.integers(0, VC, size=(2, 3, 4, W))
    lit = L.embed_tokens(ids, ch, we, fe, ce, fl, b)
    t = torch.tensor
    fus = F.embed_tokens(t(ids), t(ch), t(we), t(fe), t(ce), t(fl), t(b)).numpy()
    assert lit.shape == (2, 3, 4, cw + wd)
    np.testing.assert_allclose(fus, lit, rtol=1e-12, atol=1e-12)
    # char part first, word part second (model_v2.py:611), frozen rows index past the trainable ones (:590)
    np.testing.assert_array_equal(lit[..., cw:], np.concatenate([we, fe], 0)[ids])
    np.testing.assert_allclose(F.embed_tokens(t(ids), None, t(we), t(fe), None, None, None).numpy(),
                               L.embed_tokens(ids, None, we, fe, None, None, None))
    feat, Wm, bm = g.normal(size=(5, 17)), g.normal(size=(17, 6)), g.normal(size=6)
    pis = g.integers(0, 5, size=(2, 2, 3))
    np.testing.assert_allclose(F.image_features(t(pis), t(feat), t(Wm), t(bm), True).numpy(),
                               L.image_features(pis, feat, Wm, bm, True), rtol=1e-12)
    np.testing.assert_array_equal(L.image_features(pis, feat), feat[pis])


def test_conv1d_known_answer():
    """one filter that picks char-embedding channel 0 of the window's first position: output = relu(max over the
    W-4 window starts of that value + bias)."""
    x = np.zeros((1, 1, 8, 2))
    x[0, 0, :, 0] = [3, -1, 7, 2, 9, 100, 100, 100]          # only starts 0..3 exist for height 5
    filt = np.zeros((1, 5, 2, 1))
    filt[0, 0, 0, 0] = 1.0
    assert L.conv1d(x, filt, np.array([-1.0]))[0, 0, 0] == 6.0      # max(3,-1,7,2) - 1
    assert L.conv1d(x, filt, np.array([-10.0]))[0, 0, 0] == 0.0     # relu


@pytest.mark.parametrize("flags", [
    {},
    dict(use_ml_att=True, use_mm_att=True),
    dict(use_direct_links=True, use_choices_att=True, use_question_att=True),
    dict(use_ml_att=True, use_mm_att=True, use_direct_links=True, use_choices_att=True, use_question_att=True,
         use_eu_output=True),
    dict(use_bidirection=True, use_mm_att=True, use_choices_att=True, use_question_att=True, use_direct_links=True),
    dict(use_bidirection=True),
    dict(use_tgif_ml_att=True, use_mm_att=True),
    dict(concat=True, use_ml_att=True, use_choices_att=True),
    dict(concat=True, use_eu_output=True),
])
@pytest.mark.parametrize("simi", [1, 2, 3])
def test_model_v1_literal_vs_fused(flags, simi):
    """model.py's soft-attention baselines (model.py:831-1013): the literal restatement (tile / concat / linear as written,
    NumPy fp64) against the fused one (bilinear logits, torch fp64) on ragged inputs"""
    from fvta_memexqa_amd.synth import make_params_v1
    spec = SynthSpec(N=3, A=2, P=2, S=1, L=4, d=6, SA=2, dense=False, simiMatrix=simi, text_in=5, img_in=4)
    inputs = to_dtype(make_inputs(spec), torch.float64)
    params = to_dtype(make_params_v1(spec, inputs, use_eu_output=bool(flags.get("use_eu_output")),
                                     concat=bool(flags.get("concat"))), torch.float64)
    cfg = {**spec.cfg(), "use_question_att": False, "add_tanh": False, **flags}
    a = F.model_v1_forward(params, inputs, cfg)
    b = L.model_v1_forward(to_numpy(params), to_numpy(inputs), cfg)
    for k in ("g1", "g1_all", "gq", "gchoices", "logits", "yp", "loss"):
        np.testing.assert_allclose(a[k].detach().numpy(), b[k], rtol=1e-9, atol=1e-11, err_msg=k)


def test_model_v1_rejected_flag_sets():
    """combinations the reference's graph construction cannot build (shape mismatches / an undefined name)"""
    from fvta_memexqa_amd.synth import make_params_v1
    spec = SynthSpec(N=2, A=1, P=2, S=1, L=3, d=4, SA=1, dense=False, text_in=5, img_in=4)
    inputs = to_dtype(make_inputs(spec), torch.float64)
    params = to_dtype(make_params_v1(spec, inputs), torch.float64)
    for flags in (dict(use_bidirection=True, use_ml_att=True), dict(concat=True, use_question_att=True),
                  dict(concat=True, use_direct_links=True)):
        with pytest.raises(ValueError):
            F.model_v1_forward(params, inputs, {**spec.cfg(), "use_question_att": False, "add_tanh": False, **flags})


def test_dmn_memory_episode_matches_literal():
    """The fused restatement of the DMN+ hop loop (oracle.fvta_fused.dmn_memory, the checker of fvta_memexqa_amd/dmn.py)
    against the literal `_generate_episode` (oracle.fvta_literal.dmn_generate_episode) on the first hop's episode."""
    from oracle import fvta_fused as F
    from oracle import fvta_literal as L
    g = torch.Generator().manual_seed(4)
    N, Fn, d = 4, 6, 16
    cell = "memory/attention_gru/rnn/attention_gru_cell/"
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64) * 0.3
    p = {"memory/attention/fc1/weights": rnd(4 * d, d), "memory/attention/fc1/biases": rnd(d),
         "memory/attention/fc2/weights": rnd(d, 1), "memory/attention/fc2/biases": rnd(1),
         cell + "gates/weights": rnd(2 * d, d), cell + "gates/biases": rnd(d), cell + "candidate/weights": rnd(d, d),
         cell + "input/weights": rnd(d, d), cell + "input/biases": rnd(d),
         "memory/hop_0/dense/kernel": rnd(3 * d, d), "memory/hop_0/dense/bias": rnd(d)}
    gq, facts, lens = rnd(N, d), rnd(N, Fn, d), torch.tensor([6, 3, 1, 5])
    eps = []
    out = F.dmn_memory(gq, facts, lens, p, 1, episodes=eps)
    lit = L.dmn_generate_episode(gq.numpy(), gq.numpy(), facts.numpy(), lens.numpy(),
                                 dict(fc1_W=p["memory/attention/fc1/weights"].numpy(), fc1_b=p["memory/attention/fc1/biases"].numpy(),
                                      fc2_W=p["memory/attention/fc2/weights"].numpy(), fc2_b=p["memory/attention/fc2/biases"].numpy(),
                                      Wg=p[cell + "gates/weights"].numpy(), bg=p[cell + "gates/biases"].numpy(),
                                      Wc=p[cell + "candidate/weights"].numpy(), Wi=p[cell + "input/weights"].numpy(),
                                      bi=p[cell + "input/biases"].numpy()))
    np.testing.assert_allclose(eps[0].numpy(), lit, rtol=1e-10, atol=1e-12)
    ref = np.maximum(np.concatenate([gq.numpy(), lit, gq.numpy()], 1) @ p["memory/hop_0/dense/kernel"].numpy()
                     + p["memory/hop_0/dense/bias"].numpy(), 0.0)
    np.testing.assert_allclose(out.numpy(), ref, rtol=1e-10, atol=1e-12)


def test_dropout_hash_masks():
    """The counter-based keep masks the library and the oracle share (oracle.fvta_fused.dropout_keep_masks): keep rate,
    the two directions' independence, determinism, seed sensitivity, keep_prob 1 keeps everything."""
    from oracle import fvta_fused as F
    n = 200000
    a = F.dropout_keep_masks(n, 0.7, 123)
    assert a.shape == (2, n) and a.dtype == torch.bool
    assert abs(float(a.float().mean()) - 0.7) < 0.005
    both = float((a[0] & a[1]).float().mean())
    assert abs(both - 0.49) < 0.006                     # independent directions: P(both kept) = 0.7^2
    assert torch.equal(a, F.dropout_keep_masks(n, 0.7, 123))
    assert not torch.equal(a, F.dropout_keep_masks(n, 0.7, 124))
    assert bool(F.dropout_keep_masks(1000, 1.0, 5).all())
    assert torch.equal(F.dropout_keep_flat(2 * n, 0.7, 123).reshape(2, n), a)
