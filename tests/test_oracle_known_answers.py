"""Known-answer cases derivable by hand from the reference math.
They pin BOTH CPU restatements; the reference itself has no tests to borrow."""
import math

import numpy as np
import pytest
import torch

from oracle import fvta_fused as F
from oracle import fvta_literal as L


def _both_att3d(h, q, W, b, hm, qm, **kw):
    a1, l1 = L.attention_3d(h, q, W, b, hm, qm, **kw)
    t = lambda x: None if x is None else torch.from_numpy(np.asarray(x))
    a2, l2 = F.attention_3d(t(h), t(q), t(W), t(b), t(hm), t(qm), **kw)
    np.testing.assert_allclose(a1, a2.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(l1, l2.numpy(), rtol=1e-10)
    return a1, l1


def test_zero_weight_uniform_over_valid():
    rng = np.random.default_rng(0)
    N, K, T, JQ, w = 2, 3, 7, 4, 6
    h = rng.standard_normal((N, K, 1, T, w))
    q = rng.standard_normal((N, JQ, w))
    hm = np.zeros((N, K, 1, T), bool)
    hm[:, 0, 0, :3] = True
    hm[:, 1, 0, :5] = True           # k=2 stays empty
    qm = np.ones((N, JQ), bool)
    beta = 0.3
    ha, a = _both_att3d(h, q, np.zeros((2 * w, 1)), np.array([beta]), hm, qm, simiMatrix=2, add_tanh=True)
    valid = hm.reshape(N, K, T)
    assert np.allclose(a[valid][:, :], math.tanh(beta))
    assert (a[~valid] == -1e30).all()
    exp = 0.5 * (h[:, 0, 0, :3].mean(1) + h[:, 1, 0, :5].mean(1))   # empty K gets weight 0
    np.testing.assert_allclose(ha, exp, rtol=1e-12)


def test_fully_masked_row_goes_uniform_over_all_T():
    rng = np.random.default_rng(1)
    N, K, T, JQ, w = 1, 2, 5, 3, 4
    h = rng.standard_normal((N, K, 1, T, w))
    q = rng.standard_normal((N, JQ, w))
    hm = np.zeros((N, K, 1, T), bool)       # every (n,k) fully masked
    qm = np.ones((N, JQ), bool)
    ha, a = _both_att3d(h, q, rng.standard_normal((2 * w, 1)), np.array([0.1]), hm, qm, simiMatrix=2)
    assert (a == -1e30).all()
    np.testing.assert_allclose(ha, h.reshape(N, K, T, w).mean(2).mean(1), rtol=1e-12)
    # padded batch row: q fully masked, context valid -> same uniform behaviour
    ha2, _ = _both_att3d(h, q, rng.standard_normal((2 * w, 1)), np.array([0.1]), ~hm, np.zeros((N, JQ), bool), simiMatrix=2)
    np.testing.assert_allclose(ha2, h.reshape(N, K, T, w).mean(2).mean(1), rtol=1e-12)


def test_single_valid_t_selects_that_row():
    rng = np.random.default_rng(2)
    N, K, T, JQ, w = 2, 1, 6, 3, 4
    h = rng.standard_normal((N, K, 1, T, w))
    q = rng.standard_normal((N, JQ, w))
    hm = np.zeros((N, K, 1, T), bool)
    hm[0, 0, 0, 4] = True
    hm[1, 0, 0, 1] = True
    ha, _ = _both_att3d(h, q, rng.standard_normal((3 * w, 1)), np.array([0.0]), hm, np.ones((N, JQ), bool), simiMatrix=1)
    np.testing.assert_allclose(ha[0], h[0, 0, 0, 4], rtol=1e-12)
    np.testing.assert_allclose(ha[1], h[1, 0, 0, 1], rtol=1e-12)


def test_cosine_of_identical_vectors_is_one():
    rng = np.random.default_rng(3)
    v = rng.standard_normal((1, 1, 1, 1, 5))
    h = np.repeat(v, 3, axis=3)
    q = v.reshape(1, 1, 5)
    _, a = _both_att3d(h, q, None, None, np.ones((1, 1, 1, 3), bool), np.ones((1, 1), bool), simiMatrix=4)
    np.testing.assert_allclose(a, 1.0, rtol=1e-12)


def test_lstm_zero_kernel_known_answer():
    """kernel=0, bias=beta on the j block only: c_t=sig(1)c_{t-1}+0.5tanh(beta), h_t=0.5tanh(c_t)."""
    d, din, J, beta = 3, 2, 5, 0.7
    kernel = np.zeros((din + d, 4 * d))
    bias = np.zeros(4 * d)
    bias[d:2 * d] = beta
    x = np.random.default_rng(4).standard_normal((3, J, din))
    lens = np.array([5, 2, 0])
    out, (c, h) = L.dynamic_rnn(x, lens, kernel, bias)
    sig1 = 1 / (1 + math.exp(-1.0))
    cs, c_t = [], 0.0
    for _ in range(J):
        c_t = sig1 * c_t + 0.5 * math.tanh(beta)
        cs.append(c_t)
    for b_, Lb in enumerate(lens):
        for t in range(J):
            exp = 0.5 * math.tanh(cs[t]) if t < Lb else 0.0
            np.testing.assert_allclose(out[b_, t], exp, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(h[b_], 0.5 * math.tanh(cs[Lb - 1]) if Lb else 0.0, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(c[b_], cs[Lb - 1] if Lb else 0.0, rtol=1e-12, atol=1e-15)
    # fused oracle: same numbers (fw and bw identical here because x is ignored by a zero kernel)
    of, hf = F.lstm_direction(torch.from_numpy(x), torch.from_numpy(lens), torch.from_numpy(kernel), torch.from_numpy(bias), False)
    ob, hb = F.lstm_direction(torch.from_numpy(x), torch.from_numpy(lens), torch.from_numpy(kernel), torch.from_numpy(bias), True)
    np.testing.assert_allclose(of.numpy(), out, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(hf.numpy(), h, rtol=1e-12, atol=1e-15)
    # reversed run: position p holds step (L-1-p)
    for b_, Lb in enumerate(lens):
        for p in range(J):
            exp = 0.5 * math.tanh(cs[Lb - 1 - p]) if p < Lb else 0.0
            np.testing.assert_allclose(ob[b_, p].numpy(), exp, rtol=1e-12, atol=1e-15)


def test_bw_final_state_is_position_zero():
    rng = np.random.default_rng(5)
    d, din, J = 4, 3, 6
    kernel = rng.standard_normal((din + d, 4 * d)) * 0.3
    bias = rng.standard_normal(4 * d) * 0.1
    x = rng.standard_normal((2, J, din))
    lens = np.array([6, 3])
    (of, ob), ((_, hf), (_, hb)) = L.bidirectional_dynamic_rnn(x, lens, kernel, bias)
    np.testing.assert_allclose(hb, ob[:, 0], rtol=1e-12)
    np.testing.assert_allclose(hf[0], of[0, 5], rtol=1e-12)
    np.testing.assert_allclose(hf[1], of[1, 2], rtol=1e-12)
    assert (of[1, 3:] == 0).all() and (ob[1, 3:] == 0).all()


def test_loss_counts_padded_rows():
    """model_v2.py:1090 averages over ALL N rows; a padded row (y all False) adds 0 but divides."""
    logits = np.array([[1.0, 2.0, 0.5, -1.0], [0.3, 0.1, 0.2, 0.0]])
    y = np.array([[False, True, False, False], [False, False, False, False]])
    lse = math.log(sum(math.exp(v) for v in logits[0]))
    np.testing.assert_allclose(L.softmax_cross_entropy_mean(logits, y), (lse - 2.0) / 2, rtol=1e-12)
    np.testing.assert_allclose(F.softmax_cross_entropy_mean(torch.from_numpy(logits), torch.from_numpy(y)).item(), (lse - 2.0) / 2, rtol=1e-12)


def test_optimizer_known_steps():
    v, a, au = L.adadelta_step(np.array([1.0]), np.array([2.0]), np.zeros(1), np.zeros(1), lr=0.5)
    upd = math.sqrt(1e-8) / math.sqrt(0.05 * 4 + 1e-8) * 2.0
    np.testing.assert_allclose(v, 1.0 - 0.5 * upd, rtol=1e-12)
    np.testing.assert_allclose(a, 0.2)
    np.testing.assert_allclose(au, 0.05 * upd * upd, rtol=1e-12)
    v, m, s = L.adam_step(np.array([1.0]), np.array([2.0]), np.zeros(1), np.zeros(1), 1, lr=0.1)
    np.testing.assert_allclose(v, 1.0 - 0.1 * math.sqrt(0.001) / 0.1 * 0.2 / (math.sqrt(0.004) + 1e-8), rtol=1e-9)


def test_tf_softmax_xent_gradient_on_all_false_label_rows():
    """TF-1's softmax_cross_entropy_with_logits backprop is softmax - labels on every row (model_v2.py:1088):
    a padded row (labels all False, model_v2.py:1270) has loss 0 but still sends softmax/N into the scorer.  Pins the
    fused oracle's custom gradient against the literal restatement and against the hand value."""
    logits = torch.tensor([[0.3, -1.2, 0.8, 0.1], [2.0, 0.0, -1.0, 0.5], [0.0, 0.0, 0.0, 0.0]], dtype=torch.float64)
    y = torch.tensor([[0, 0, 1, 0], [1, 0, 0, 0], [0, 0, 0, 0]], dtype=torch.bool)   # last row: padded
    lt = logits.clone().requires_grad_()
    loss = F.softmax_cross_entropy_mean(lt, y)            # default: TF gradient
    loss.backward()
    np.testing.assert_allclose(float(loss), L.softmax_cross_entropy_mean(logits.numpy(), y.numpy()), rtol=1e-12)
    np.testing.assert_allclose(lt.grad.numpy(), L.softmax_cross_entropy_mean_grad(logits.numpy(), y.numpy()), rtol=1e-12)
    np.testing.assert_allclose(lt.grad[2].numpy(), np.full(4, 0.25 / 3), rtol=1e-12)    # uniform softmax / N
    lm = logits.clone().requires_grad_()
    loss_m = F.softmax_cross_entropy_mean(lm, y, tf_grad=False)
    loss_m.backward()
    assert float(loss_m) == float(loss)
    np.testing.assert_allclose(lm.grad[:2].numpy(), lt.grad[:2].numpy(), rtol=1e-12)   # rows with one label agree
    assert (lm.grad[2] == 0).all()                                                      # the mathematical form: nothing


def test_time_warp_att_masked_rows_with_negative_scale_take_the_softmax():
    """model_v2.py:263-275: the scale sum_t' C[n,t,t'] multiplies the max-pooled logits AFTER exp_mask, so a masked
    row's softmax logit is -1e30 * scale.  scale > 0 everywhere: masked rows weigh 0 and u is the softmax over the
    valid rows of amax * scale.  One masked row with scale < 0: it takes ALL the weight, u = that row's h.  Two masked
    rows tied at the smallest scale: they share it equally."""
    rng = np.random.default_rng(3)
    N, K, T, JQ, w = 1, 1, 6, 2, 4
    h = rng.standard_normal((N, K, 1, T, w))
    q = rng.standard_normal((N, JQ, w))
    hm = np.zeros((N, K, 1, T), bool)
    hm[..., :3] = True                                     # rows 3, 4, 5 are masked
    qm = np.ones((N, JQ), bool)
    W, b = np.zeros((2 * w, 1)), np.array([0.4])           # a = tanh(0.4) on valid cells
    am = math.tanh(0.4)

    def run(scale):
        C = np.diag(scale)[None]                           # row sums = scale
        ha, _ = _both_att3d(h, q, W, b, hm, qm, simiMatrix=2, add_tanh=True, time_warp_att=True, C=C)
        return ha[0]

    s_pos = np.array([0.5, 1.0, 2.0, 0.3, 0.7, 0.9])
    p = np.exp(am * s_pos[:3])
    p /= p.sum()
    np.testing.assert_allclose(run(s_pos), (p[:, None] * h[0, 0, 0, :3]).sum(0), rtol=1e-12)
    s_neg = s_pos.copy()
    s_neg[4] = -0.2                                        # masked row 4: logit +2e29
    np.testing.assert_allclose(run(s_neg), h[0, 0, 0, 4], rtol=1e-12)
    s_tie = s_pos.copy()
    s_tie[3] = s_tie[5] = -0.6
    np.testing.assert_allclose(run(s_tie), 0.5 * (h[0, 0, 0, 3] + h[0, 0, 0, 5]), rtol=1e-12)
