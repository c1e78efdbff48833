"""GPU parity of the time warp (model_v2.py:953-1009, closed form) forward and backward vs the oracle
(whose closed form is itself checked against the literal O(T^2) restatement in tests/test_oracle_cross.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("warp_type", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("N,K,T,w", [(2, 3, 17, 64), (3, 6, 50, 256)])
def test_timewarp_forward_backward(warp_type, N, K, T, w):
    from fvta_memexqa_amd import ops
    from oracle import fvta_fused as F
    g = torch.Generator().manual_seed(warp_type * 100 + T)
    hall = torch.randn(N, K, 1, T, w, generator=g) * 0.5
    lq = torch.randn(N, w, generator=g) * 0.5
    WH_W = torch.randn(2 * w, w, generator=g) * (0.3 / w ** 0.5)
    WH_b = torch.randn(w, generator=g) * 0.1
    WC_W = torch.randn(w, 1, generator=g) * (0.5 / w ** 0.5)
    WC_b = torch.randn(1, generator=g) * 0.1
    gout = torch.randn(N, K, 1, T, w, generator=g)
    leaves = [t.double().requires_grad_() for t in (hall, lq, WH_W, WH_b, WC_W, WC_b)]
    ref, c = F.time_warp_closed(*leaves, warp_type=warp_type, window_t=2.3)
    (ref * gout.double()).sum().backward()
    cu = lambda t: t.cuda().contiguous()
    op = ops.TimeWarp(N, K, T, w, warp_type, 2.3)
    hd, lqd, WHd, WHbd, WCd, WCbd = cu(hall.reshape(N, K, T, w)), cu(lq), cu(WH_W), cu(WH_b), cu(WC_W.reshape(-1)), cu(WC_b)
    warp = torch.empty_like(hd)
    op.forward(hd, lqd, WHd, WHbd, WCd, WCbd, warp)
    np.testing.assert_allclose(warp.cpu().numpy(), ref.detach().reshape(N, K, T, w).numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(op.c.cpu().numpy(), c.detach().numpy(), rtol=1e-4, atol=1e-6)
    d_hall = torch.empty_like(hd)
    d_lq = torch.zeros_like(lqd)
    dWH, dWHb, dWC, dWCb = torch.zeros_like(WHd), torch.zeros_like(WHbd), torch.zeros_like(WCd), torch.zeros_like(WCbd)
    op.backward(hd, lqd, WHd, WHbd, WCd, WCbd, cu(gout.reshape(N, K, T, w)), d_hall, d_lq, dWH, dWHb, dWC, dWCb)
    for got, want, name in [(d_hall, leaves[0].grad.reshape(N, K, T, w), "d_hall"), (d_lq, leaves[1].grad, "d_lq"),
                            (dWH, leaves[2].grad, "dWH_W"), (dWHb, leaves[3].grad, "dWH_b"),
                            (dWC, leaves[4].grad.reshape(-1), "dWC_W"), (dWCb, leaves[5].grad, "dWC_b")]:
        wv = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), wv, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(wv).max()), err_msg=name)


def test_unknown_warp_type_raises_like_the_reference():
    from fvta_memexqa_amd import ops
    with pytest.raises(Exception, match="time warping type not implemented"):
        ops.TimeWarp(1, 1, 4, 64, 9)
