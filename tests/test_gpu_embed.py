"""Embedding front-end (model_v2.py:52-70, 524-645) on the GPU vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _close(a, b, rtol=RTOL, atol=1e-5, msg=""):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def _case(seed, B, J, W, cd, cw, wd, VW, VF, VC):
    g = torch.Generator().manual_seed(seed)
    p = dict(word_emb=torch.randn(VW, wd, generator=g), fixed=torch.randn(VF, wd, generator=g),
             char_emb=torch.randn(VC, cd, generator=g), filt=torch.randn(1, 5, cd, cw, generator=g) * 0.3,
             bias=torch.randn(cw, generator=g) * 0.2)
    ids = torch.randint(0, VW + VF, (B, J), generator=g, dtype=torch.int32)
    ch = torch.randint(0, VC, (B, J, W), generator=g, dtype=torch.int32)
    return p, ids, ch


@pytest.mark.parametrize("B,J,W,cd,cw,wd,VW,VF,VC", [(3, 7, 16, 8, 100, 100, 50, 30, 40), (5, 4, 9, 4, 24, 300, 7, 5, 11),
                                                      (64, 30, 16, 8, 100, 100, 500, 2000, 97),
                                                      (4, 6, 16, 100, 100, 100, 20, 30, 60),   # README.MD:144 flags
                                                      (40, 30, 16, 100, 100, 60, 30, 30, 70),  # ... with more tokens than waves: the token loops
                                                      (5, 7, 9, 100, 100, 40, 10, 10, 30),     # ... and short words (5 windows)
                                                      (6, 9, 16, 100, 100, 140, 10, 10, 200),  # ... wdim > 128 and VC > 128: outside the published-shape kernels -> the general deep-window kernels (the tested fallback)
                                                      (3, 5, 21, 8, 64, 50, 9, 9, 20),          # W > 16: general kernels
                                                      (7, 9, 9, 8, 100, 60, 30, 20, 50),        # short words, 100 filters
                                                      (33, 30, 16, 8, 100, 100, 40, 2000, 256),
                                                      (1, 1, 16, 8, 100, 100, 5, 5, 30),        # one token: three idle waves
                                                      (2, 3, 5, 8, 100, 100, 10, 10, 20),       # W = height: one window
                                                      (2, 5, 16, 8, 100, 140, 10, 10, 300),     # wdim > 128; VC > 256: the general kernels
                                                      (3, 5, 16, 8, 100, 140, 10, 10, 30)])     # wdim > 128 on the wave kernels
def test_token_embed_forward_backward(B, J, W, cd, cw, wd, VW, VF, VC):
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    p, ids, ch = _case(B + J + W, B, J, W, cd, cw, wd, VW, VF, VC)
    p64 = {k: v.double().requires_grad_() for k, v in p.items()}
    ref = F.embed_tokens(ids, ch, p64["word_emb"], p64["fixed"], p64["char_emb"], p64["filt"], p64["bias"])
    g = torch.Generator().manual_seed(5)
    gout = torch.randn(B, J, cw + wd, generator=g)
    (ref * gout.double()).sum().backward()
    # rows land in an arena with a stride wider than the row (as the padded encoder input has)
    stride = cw + wd + 8
    ntok = B * J
    op = ops.TokenEmbed(ntok, W, cd, cw, wd, VW, VW + VF, VC)
    cu = lambda t: t.cuda().contiguous()
    tok_off = cu(torch.arange(ntok, dtype=torch.int64) * stride + 4)
    x = torch.zeros(ntok * stride + 4, device="cuda")
    filt = cu(p["filt"].reshape(5, cd, cw))
    args = (cu(ids.reshape(-1)), cu(ch.reshape(-1, W)), tok_off)
    op.forward(*args, cu(p["word_emb"]), cu(p["fixed"]), cu(p["char_emb"]), filt, cu(p["bias"]), x)
    got = x[4:].view(ntok, stride)[:, :cw + wd]
    _close(got, ref.reshape(ntok, -1), msg="x")
    dx = torch.zeros_like(x)
    dx[4:].view(ntok, stride)[:, :cw + wd] = cu(gout.reshape(ntok, -1))
    dwe, dce = torch.zeros(VW, wd, device="cuda"), torch.zeros(VC, cd, device="cuda")
    dfl, dbi = torch.zeros(5, cd, cw, device="cuda"), torch.zeros(cw, device="cuda")
    op.backward(*args, cu(p["char_emb"]), filt, dx, dwe, dce, dfl, dbi)
    _close(dwe, p64["word_emb"].grad, atol=1e-4, msg="d word_emb")
    _close(dce, p64["char_emb"].grad, atol=1e-4, msg="d char_emb")
    # d filt of the wide shape (char_emb_size 100) runs on the bf16 matrix pipe with the three-term split of the bi-LSTM's
    # bf16x3 engine (~2^-16 per product): that engine's bound -- rtol 1e-4 plus 3e-5 of the tensor's scale (a sum over every
    # token of terms of that scale; tests/test_gpu_bf16.py) -- and never looser than the exact kernels' 1e-4
    ref_dfl = p64["filt"].grad.reshape(5, cd, cw)
    _close(dfl, ref_dfl, atol=max(1e-4, 3e-5 * float(ref_dfl.abs().max())) if cd > 8 else 1e-4, msg="d filt")
    _close(dbi, p64["bias"].grad, atol=1e-4, msg="d bias")
    assert p64["fixed"].grad is not None  # the oracle differentiates it; the library treats it as frozen (model_v2.py:590)


@pytest.mark.parametrize("B,J,W,cd,cw,wd,VW,VF,VC", [(3, 7, 16, 8, 100, 100, 50, 30, 40),    # the 5 x 8 register kernels
                                                      (5, 4, 9, 4, 24, 300, 7, 5, 11),        # generic kernels
                                                      (4, 6, 16, 100, 100, 100, 20, 30, 60),  # wide: sparse backward pair
                                                      (3, 5, 21, 8, 64, 50, 9, 9, 20),        # W > 16
                                                      (7, 9, 9, 8, 100, 60, 30, 20, 50)])     # short words, 100 filters
def test_token_embed_char_dropout(B, J, W, cd, cw, wd, VW, VF, VC):
    """conv1d's dropout of the gathered char embeddings while training (model_v2.py:58-62): forward rows and every
    gradient against the oracle run with the same keep mask (the library's hash, oracle dropout_keep_flat)."""
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    keep, seed = 0.8, 0xABCDEF0123
    p, ids, ch = _case(B + J + W + 1, B, J, W, cd, cw, wd, VW, VF, VC)
    ntok = B * J
    km = F.dropout_keep_flat(ntok * W * cd, keep, seed).reshape(B, J, W, cd)
    assert 0.7 < km.float().mean() < 0.9
    p64 = {k: v.double().requires_grad_() for k, v in p.items()}
    ref = F.embed_tokens(ids, ch, p64["word_emb"], p64["fixed"], p64["char_emb"], p64["filt"], p64["bias"], char_keep=km,
                         keep_prob=keep)
    g = torch.Generator().manual_seed(6)
    gout = torch.randn(B, J, cw + wd, generator=g)
    (ref * gout.double()).sum().backward()
    stride = cw + wd
    op = ops.TokenEmbed(ntok, W, cd, cw, wd, VW, VW + VF, VC)
    op.set_dropout(keep, seed)
    cu = lambda t: t.cuda().contiguous()
    tok_off = cu(torch.arange(ntok, dtype=torch.int64) * stride)
    x = torch.zeros(ntok * stride, device="cuda")
    filt = cu(p["filt"].reshape(5, cd, cw))
    args = (cu(ids.reshape(-1)), cu(ch.reshape(-1, W)), tok_off)
    op.forward(*args, cu(p["word_emb"]), cu(p["fixed"]), cu(p["char_emb"]), filt, cu(p["bias"]), x)
    _close(x.view(ntok, stride), ref.reshape(ntok, -1), msg="x")
    dx = cu(gout.reshape(-1))
    dwe, dce = torch.zeros(VW, wd, device="cuda"), torch.zeros(VC, cd, device="cuda")
    dfl, dbi = torch.zeros(5, cd, cw, device="cuda"), torch.zeros(cw, device="cuda")
    op.backward(*args, cu(p["char_emb"]), filt, dx, dwe, dce, dfl, dbi)
    _close(dwe, p64["word_emb"].grad, atol=1e-4, msg="d word_emb")
    _close(dce, p64["char_emb"].grad, atol=1e-4, msg="d char_emb")
    ref_dfl = p64["filt"].grad.reshape(5, cd, cw)    # (wide shape: the bf16x3 bound, see test_token_embed_forward_backward)
    _close(dfl, ref_dfl, atol=max(1e-4, 3e-5 * float(ref_dfl.abs().max())) if cd > 8 else 1e-4, msg="d filt")
    _close(dbi, p64["bias"].grad, atol=1e-4, msg="d bias")
    # switched off again: the plain rows
    op.set_dropout(1.0, 0)
    op.forward(*args, cu(p["word_emb"]), cu(p["fixed"]), cu(p["char_emb"]), filt, cu(p["bias"]), x)
    plain = F.embed_tokens(ids, ch, p["word_emb"], p["fixed"], p["char_emb"], p["filt"], p["bias"])
    _close(x.view(ntok, stride), plain.reshape(ntok, -1), msg="x without dropout")


def test_token_embed_without_char_cnn():
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    p, ids, ch = _case(3, 4, 6, 16, 8, 100, 50, 20, 10, 30)
    ref = F.embed_tokens(ids, ch, p["word_emb"], p["fixed"], None, None, None)
    ntok = 24
    op = ops.TokenEmbed(ntok, 16, 8, 0, 50, 20, 30, 30)
    cu = lambda t: t.cuda().contiguous()
    x = torch.zeros(ntok * 50, device="cuda")
    op.forward(cu(ids.reshape(-1)), None, cu(torch.arange(ntok, dtype=torch.int64) * 50), cu(p["word_emb"]), cu(p["fixed"]),
               None, None, None, x)
    _close(x.view(ntok, 50), ref.reshape(ntok, 50))


@pytest.mark.parametrize("trans,tanh,shape", [(True, True, (37, 157, 100, 45)), (True, False, (37, 157, 100, 45)),
                                              (False, False, (37, 157, 157, 45)),
                                              (True, True, (300, 2537, 100, 2560)),   # the metric shape (matrix-pipe kernels)
                                              (True, True, (20, 61, 8, 33)),          # narrow: one column group
                                              (True, False, (20, 61, 128, 19)),       # widest the matrix-pipe kernels take
                                              (True, True, (20, 61, 30, 33))])        # tdim % 4 != 0: generic kernels
def test_image_features(trans, tanh, shape):
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    g = torch.Generator().manual_seed(11)
    VI, idim, tdim, M = shape
    feat = torch.randn(VI, idim, generator=g)
    W = (torch.randn(idim, tdim, generator=g) * 0.1).double().requires_grad_() if trans else None
    b = (torch.randn(tdim, generator=g) * 0.1).double().requires_grad_() if trans else None
    pis = torch.randint(0, VI, (M,), generator=g, dtype=torch.int32)
    ref = F.image_features(pis, feat.double(), W, b, tanh)
    stride = tdim + 4
    cu = lambda t: None if t is None else t.detach().float().cuda().contiguous()
    row_off = cu(torch.arange(M, dtype=torch.int64) * stride).long()
    x = torch.zeros(M * stride, device="cuda")
    op = ops.ImageTrans(M, idim, tdim, tanh)
    op.forward(cu(pis).int(), row_off, cu(feat), cu(W), cu(b), x)
    _close(x.view(M, stride)[:, :tdim], ref, msg="x")
    if trans:
        gout = torch.randn(M, tdim, generator=g)
        (ref * gout.double()).sum().backward()
        dx = torch.zeros_like(x)
        dx.view(M, stride)[:, :tdim] = cu(gout)
        dW, db = torch.zeros(idim, tdim, device="cuda"), torch.zeros(tdim, device="cuda")
        op.backward(cu(pis).int(), row_off, cu(feat), x, dx, dW, db)
        _close(dW, W.grad, atol=1e-4, msg="dW")
        _close(db, b.grad, atol=1e-4, msg="db")


def test_token_embed_metric_size_properties():
    """BASELINE.json configs[2] size (385,920 tokens of 16 characters): every wave of the wave-per-token kernels walks
    ~190 tokens and 2,048 slabs are reduced.  Forward: a random sample of 3,000 tokens against the oracle (tokens are
    independent).  Backward: additivity over a split of the token list -- the parameter gradients of all tokens equal the
    sum of the two halves' (size-independent; fixed-order sums, so only rounding differs) -- and a bitwise repeat."""
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    B, J, W, cd, cw, wd, VW, VF, VC = 12864, 30, 16, 8, 100, 100, 400, 20000, 100
    p, ids, ch = _case(77, B, J, W, cd, cw, wd, VW, VF, VC)
    ntok, stride = B * J, cw + wd + 24
    cu = lambda t: t.cuda().contiguous()
    ids_d, ch_d = cu(ids.reshape(-1)), cu(ch.reshape(-1, W))
    tok_off = cu(torch.arange(ntok, dtype=torch.int64) * stride)
    x = torch.zeros(ntok * stride, device="cuda")
    filt = cu(p["filt"].reshape(5, cd, cw))
    we, fe, ce, bi = cu(p["word_emb"]), cu(p["fixed"]), cu(p["char_emb"]), cu(p["bias"])
    op = ops.TokenEmbed(ntok, W, cd, cw, wd, VW, VW + VF, VC)
    op.forward(ids_d, ch_d, tok_off, we, fe, ce, filt, bi, x)
    g = torch.Generator().manual_seed(9)
    pick = torch.randperm(ntok, generator=g)[:3000]
    ref = F.embed_tokens(ids.reshape(-1)[pick].reshape(1, -1), ch.reshape(-1, W)[pick].reshape(1, -1, W),
                         p["word_emb"].double(), p["fixed"].double(), p["char_emb"].double(), p["filt"].double(),
                         p["bias"].double())
    _close(x.view(ntok, stride)[pick.cuda(), :cw + wd], ref.reshape(3000, -1), msg="x (sample)")
    dx = torch.randn(ntok * stride, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))

    def grads(lo, hi):
        n = hi - lo
        o = ops.TokenEmbed(n, W, cd, cw, wd, VW, VW + VF, VC)
        o.argpos.copy_(op.argpos.view(ntok, cw)[lo:hi].reshape(-1))
        out = [torch.zeros(VW, wd, device="cuda"), torch.zeros(VC, cd, device="cuda"), torch.zeros(5, cd, cw, device="cuda"),
               torch.zeros(cw, device="cuda")]
        o.backward(ids_d[lo:hi].contiguous(), ch_d[lo:hi].contiguous(), tok_off[lo:hi].contiguous(), ce, filt, dx, *out)
        return out
    whole, again = grads(0, ntok), grads(0, ntok)
    half = ntok // 2 + 7
    a, b = grads(0, half), grads(half, ntok)
    for name, w_, r_, a_, b_ in zip(("d word_emb", "d char_emb", "d filt", "d bias"), whole, again, a, b):
        if name != "d word_emb":                       # (the word rows are atomics: summation order varies)
            assert torch.equal(w_, r_), name + " repeat"
        scale = float(w_.abs().max())
        _close(w_, a_ + b_, rtol=1e-4, atol=2e-5 * scale, msg=name + " additivity")
