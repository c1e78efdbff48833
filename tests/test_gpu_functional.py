"""The reference's functional helpers, stand-alone: each op through libfvta_hip.so vs the literal oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _g(seed=0):
    return torch.Generator().manual_seed(seed)


def _close(got, ref, rtol=1e-4, atol=1e-6, msg=""):
    np.testing.assert_allclose(got.detach().cpu().numpy(), np.asarray(ref), rtol=rtol, atol=atol, err_msg=msg)


@pytest.mark.parametrize("shape", [(3, 5, 7), (2, 4, 3, 130), (1, 1)])
def test_softmax(shape):
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    x = torch.randn(*shape, generator=_g(1)) * 3
    _close(Fn.softmax(x.cuda()), L.softmax(x.double().numpy()))


@pytest.mark.parametrize("lead,J,d", [((3, 2), 7, 33), ((5,), 300, 256), ((2, 3, 4), 1, 5)])
def test_softsel(lead, J, d):
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    t = torch.randn(*lead, J, d, generator=_g(2))
    lg = torch.randn(*lead, J, generator=_g(3)) * 2
    _close(Fn.softsel(t.cuda(), lg.cuda()), L.softsel(t.double().numpy(), lg.double().numpy()))


def test_softsel_of_exp_masked_logits_ignores_masked_rows():
    from fvta_memexqa_amd import functional as Fn
    t = torch.randn(4, 6, 8, generator=_g(4))
    lg = torch.randn(4, 6, generator=_g(5))
    mask = torch.rand(4, 6, generator=_g(6)) > 0.4
    mask[:, 0] = True
    out = Fn.softsel(t.cuda(), Fn.exp_mask(lg.cuda(), mask.cuda()))
    w = torch.softmax(lg.masked_fill(~mask, float("-inf")), -1)
    _close(out, (w[..., None] * t).sum(1).numpy())
    # exp_mask itself: utils.py:213 literally
    _close(Fn.exp_mask(lg.cuda(), mask.cuda()), (lg + (1 - mask.float()) * -1e30).numpy(), rtol=0, atol=0)


@pytest.mark.parametrize("add_tanh", [False, True])
def test_linear_creates_and_reuses_its_variables(add_tanh):
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    Fn.reset_default_graph()
    x = torch.randn(3, 5, 77, generator=_g(7))
    with Fn.variable_scope("emb"):
        y = Fn.linear(x.cuda(), 19, scope="image_trans_linear", add_tanh=add_tanh, wd=0.01)
        y2 = Fn.linear(x.cuda(), 19, scope="image_trans_linear", add_tanh=add_tanh)
    assert set(Fn.variables) == {"emb/image_trans_linear/W", "emb/image_trans_linear/b"}
    W, b = Fn.variables["emb/image_trans_linear/W"], Fn.variables["emb/image_trans_linear/b"]
    assert abs(float(W.std()) - 0.088) < 0.02 and float(W.abs().max()) <= 0.2 + 1e-6 and float(b.abs().max()) == 0   # TN(0.1), zeros
    ref = L.linear(x.double().numpy(), W.cpu().double().numpy(), b.cpu().double().numpy(), add_tanh)
    _close(y, ref, msg="linear")
    assert torch.equal(y, y2)
    assert len(Fn.losses) == 2                                              # add_wd: one l2 term per variable of the scope
    np.testing.assert_allclose(float(Fn.losses[0]), 0.01 * 0.5 * float((W.double() ** 2).sum()), rtol=1e-5)


@pytest.mark.parametrize("simi,add_tanh,masked,w", [(1, False, True, 64), (2, True, True, 100), (3, True, False, 128), (4, False, True, 100)])
def test_attention_3d_signature_and_values(simi, add_tanh, masked, w):
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    Fn.reset_default_graph()
    N, K, M, JX, JQ = 2, 3, 2, 5, 4
    h = torch.randn(N, K, M, JX, w, generator=_g(8)) * 0.5
    q = torch.randn(N, JQ, w, generator=_g(9)) * 0.5
    hm = torch.rand(N, K, M, JX, generator=_g(10)) > 0.3
    hm[:, :, 0, 0] = True
    qm = torch.rand(N, JQ, generator=_g(11)) > 0.2
    qm[:, 0] = True
    with Fn.variable_scope("attention"):
        h_a, a = Fn.attention_3d(h.cuda(), q.cuda(), hm.cuda() if masked else None, qm.cuda() if masked else None,
                                 simiMatrix=simi, add_tanh=add_tanh, scope="all")
    assert tuple(h_a.shape) == (N, w) and tuple(a.shape) == (N, K, M * JX, JQ)
    Wv = Fn.variables.get("attention/all/att_logits/W")
    bv = Fn.variables.get("attention/all/att_logits/b")
    assert (Wv is None) == (simi == 4)
    ref_h, ref_a = L.attention_3d(h.double().numpy(), q.double().numpy(),
                                  None if Wv is None else Wv.cpu().double().numpy(), None if bv is None else bv.cpu().double().numpy(),
                                  hm.numpy() if masked else None, qm.numpy() if masked else None, simiMatrix=simi,
                                  add_tanh=add_tanh)
    _close(h_a, ref_h, rtol=2e-4, atol=2e-6, msg="h_a")
    ref_a = np.asarray(ref_a).reshape(N, K, M * JX, JQ)
    valid = ref_a > -1e29
    _close(a.cpu()[torch.from_numpy(valid)], ref_a[valid], rtol=2e-4, atol=2e-5, msg="a_logits")
    assert (a.cpu().numpy()[~valid] < -1e29).all()


def test_attention_1d_question_attention_form():
    """model_v2.py:1044: attention(hq, g1[:,None], q_mask, ones) -> gq"""
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    Fn.reset_default_graph()
    N, JQ, w = 3, 6, 64
    hq = torch.randn(N, JQ, w, generator=_g(12)) * 0.5
    g1 = torch.randn(N, 1, w, generator=_g(13)) * 0.5
    qm = torch.rand(N, JQ, generator=_g(14)) > 0.3
    qm[:, 0] = True
    with Fn.variable_scope("question_emb"):
        gq, a = Fn.attention(hq.cuda(), g1.cuda(), qm.cuda(), torch.ones(N, 1, dtype=torch.bool).cuda(), simiMatrix=2,
                             add_tanh=True, scope="question_att")
    W, b = Fn.variables["question_emb/question_att/att_logits/W"], Fn.variables["question_emb/question_att/att_logits/b"]
    ref_g, ref_a = L.attention(hq.double().numpy(), g1.double().numpy(), W.cpu().double().numpy(), b.cpu().double().numpy(),
                               qm.numpy(), np.ones((N, 1), bool), simiMatrix=2, add_tanh=True)
    assert tuple(a.shape) == (N, JQ, 1)
    _close(gq, ref_g, rtol=2e-4, atol=2e-6)


def test_unknown_similarity_and_unbuilt_switches():
    from fvta_memexqa_amd import functional as Fn
    h, q = torch.zeros(1, 1, 2, 64).cuda(), torch.zeros(1, 2, 64).cuda()
    with pytest.raises(ValueError, match="similarity matrix not implemented"):
        Fn.attention_3d(h, q, simiMatrix=5)
    with pytest.raises(NotImplementedError):
        Fn.attention_3d(h, q, bidirect=True)                 # shape error in the reference's 3-D branch too


def test_attention_3d_functional_time_warp_att():
    """attention_3d(..., time_warp_att=True, C=C) with the reference's keyword signature: C [N,T,T] arbitrary (its row
    sums scale the max-pooled logits, model_v2.py:269-275), masks given, vs the literal oracle."""
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    Fn.reset_default_graph()
    g = torch.Generator().manual_seed(21)
    N, K, M, JX, JQ, w = 2, 3, 2, 6, 4, 64
    T = M * JX
    h = torch.randn(N, K, M, JX, w, generator=g) * 0.5
    q = torch.randn(N, JQ, w, generator=g) * 0.5
    hm = torch.rand(N, K, M, JX, generator=g) < 0.7
    hm[:, :, 0, 0] = True
    qm = torch.ones(N, JQ, dtype=torch.bool)
    C = torch.randn(N, T, T, generator=g) * 0.4
    ha, a = Fn.attention_3d(h.cuda(), q.cuda(), hm.cuda(), qm.cuda(), simiMatrix=2, add_tanh=True, time_warp_att=True,
                            C=C.cuda(), scope="tw")
    W = Fn.variables["tw/att_logits/W"].cpu().double().numpy()
    b = Fn.variables["tw/att_logits/b"].cpu().double().numpy()
    ref, ref_a = L.attention_3d(h.double().numpy(), q.double().numpy(), W, b, hm.numpy(), qm.numpy(), simiMatrix=2,
                                add_tanh=True, time_warp_att=True, C=C.double().numpy())
    _close(ha, torch.from_numpy(ref), rtol=2e-4, atol=2e-5)
    with pytest.raises(ValueError, match="time_warp_att"):
        Fn.attention_3d(h.cuda(), q.cuda(), time_warp_att=True)


@pytest.mark.parametrize("simi,masked,tanh", [(2, True, True), (1, False, False), (3, True, False)])
def test_attention_bidirect(simi, masked, tanh):
    """attention(..., bidirect=True) (model_v2.py:184-192, model.py:169-177): h_a [N,2w] = concat(softsel over the rows,
    mean over the rows of the question attended by each row -- masked rows attend it uniformly)."""
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    Fn.reset_default_graph()
    g = torch.Generator().manual_seed(31 + simi)
    N, V, JQ, w = 3, 11, 6, 100
    h = torch.randn(N, V, w, generator=g) * 0.5
    q = torch.randn(N, JQ, w, generator=g) * 0.5
    hm = torch.arange(V)[None, :] < torch.randint(1, V + 1, (N,), generator=g)[:, None]
    qm = torch.arange(JQ)[None, :] < torch.randint(1, JQ + 1, (N,), generator=g)[:, None]
    kw = dict(hinfo_mask=hm.cuda(), hq_mask=qm.cuda()) if masked else {}
    ha, a = Fn.attention(h.cuda(), q.cuda(), simiMatrix=simi, add_tanh=tanh, bidirect=True, scope="bi", **kw)
    W = Fn.variables["bi/att_logits/W"].cpu().double().numpy()
    b = Fn.variables["bi/att_logits/b"].cpu().double().numpy()
    ref, ref_a = L.attention(h.double().numpy(), q.double().numpy(), W, b, hm.numpy() if masked else None,
                             qm.numpy() if masked else None, simiMatrix=simi, add_tanh=tanh, bidirect=True)
    assert tuple(ha.shape) == (N, 2 * w)
    _close(ha, ref, rtol=2e-4, atol=2e-5)
    _close(a, ref_a, rtol=2e-4, atol=2e-5)


def test_attention_keeprank1_bidirect():
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    Fn.reset_default_graph()
    g = torch.Generator().manual_seed(77)
    N, M, V, JQ, w = 2, 3, 7, 5, 64
    h = torch.randn(N, M, V, w, generator=g) * 0.5
    q = torch.randn(N, JQ, w, generator=g) * 0.5
    hm = torch.rand(N, M, V, generator=g) < 0.7
    hm[:, :, 0] = True
    qm = torch.ones(N, JQ, dtype=torch.bool)
    out = Fn.attention_keeprank1(h.cuda(), q.cuda(), hm.cuda(), qm.cuda(), simiMatrix=2, bidirect=True, scope="kb")
    W = Fn.variables["kb/att_logits/W"].cpu().double().numpy()
    b = Fn.variables["kb/att_logits/b"].cpu().double().numpy()
    ref = L.attention_keeprank1(h.double().numpy(), q.double().numpy(), W, b, hm.numpy(), qm.numpy(), simiMatrix=2,
                                bidirect=True)
    assert tuple(out.shape) == (N, M, 2 * w)
    _close(out, ref, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("simi,masked,w", [(1, True, 64), (2, True, 100), (3, False, 128)])
def test_attention_keeprank1(simi, masked, w):
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    Fn.reset_default_graph()
    N, M, V, JQ = 3, 4, 9, 5
    h = torch.randn(N, M, V, w, generator=_g(20)) * 0.5
    q = torch.randn(N, JQ, w, generator=_g(21)) * 0.5
    hm = torch.rand(N, M, V, generator=_g(22)) > 0.3
    hm[:, :, 0] = True
    qm = torch.rand(N, JQ, generator=_g(23)) > 0.2
    qm[:, 0] = True
    with Fn.variable_scope("album_att"):
        h_a = Fn.attention_keeprank1(h.cuda(), q.cuda(), hm.cuda() if masked else None, qm.cuda() if masked else None,
                                     simiMatrix=simi, scope="keeprank")
    W, b = Fn.variables["album_att/keeprank/att_logits/W"], Fn.variables["album_att/keeprank/att_logits/b"]
    ref = L.attention_keeprank1(h.double().numpy(), q.double().numpy(), W.cpu().double().numpy(), b.cpu().double().numpy(),
                                hm.numpy() if masked else None, qm.numpy() if masked else None, simiMatrix=simi)
    assert tuple(h_a.shape) == (N, M, w)
    _close(h_a, ref, rtol=2e-4, atol=2e-6)


def test_attention_tgif():
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    Fn.reset_default_graph()
    N, V, mlp = 3, 11, 16
    w = 2 * mlp
    lens = torch.tensor([11, 4, 7])
    hm = torch.arange(V)[None, :] < lens[:, None]
    h = torch.randn(N, V, w, generator=_g(30)) * 0.5 * hm[..., None]          # dynamic_rnn zeroes the padded outputs
    lq = torch.randn(N, w, generator=_g(31)) * 0.5
    logits, att = Fn.attention_tgif(h.cuda(), lq.cuda(), hm.cuda(), mlp_dim=mlp, scope="tgif")
    v = lambda n: Fn.variables["tgif/" + n].cpu().double().numpy()
    ref, ref_att = L.attention_tgif(h.double().numpy(), lq.double().numpy(), v("mlp_q/W"), v("mlp_q/b"), v("mlp_h/W"),
                                    v("mlp_h/b"), v("preatt/W"), v("preatt/b"), v("final/W"), v("final/b"), hm.numpy())
    assert tuple(logits.shape) == (N, w) and tuple(att.shape) == (N, V)
    _close(logits, ref, rtol=2e-4, atol=2e-6)
    valid = hm.numpy()
    _close(att.cpu()[hm], ref_att[valid], rtol=2e-4, atol=1e-7)
    assert (att.cpu().numpy()[~valid] < -1e29).all()                           # the reference's exp_mask on probabilities
