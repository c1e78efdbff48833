"""Algebraic identities the HIP kernels rely on, checked on the oracle."""
import numpy as np
import torch

from oracle import fvta_fused as F
from oracle import fvta_literal as L


def test_tanh_max_commute():
    x = np.random.default_rng(0).standard_normal((50, 7)) * 3
    np.testing.assert_array_equal(np.tanh(x).max(1), np.tanh(x.max(1)))


def test_masked_tanh_is_exactly_minus_1e30_in_fp32():
    x = np.float32(np.random.default_rng(1).standard_normal(1000) * 5)
    assert (np.tanh(x) + np.float32(-1e30) == np.float32(-1e30)).all()
    assert (x + np.float32(-1e30) == np.float32(-1e30)).all()


def test_gemm_decomposition_all_simi():
    g = torch.Generator().manual_seed(0)
    w = 10
    h = torch.randn(3, 6, w, generator=g, dtype=torch.float64)
    q = torch.randn(3, 4, w, generator=g, dtype=torch.float64)
    b = torch.randn(1, generator=g, dtype=torch.float64)
    for simi, Fdim in ((1, 3 * w), (2, 2 * w), (3, 4 * w)):
        W = torch.randn(Fdim, 1, generator=g, dtype=torch.float64)
        x = F.simi_logits(h, q, W, b, simi, False)
        h_aug = h[:, :, None, :].expand(3, 6, 4, w).numpy()
        q_aug = q[:, None, :, :].expand(3, 6, 4, w).numpy()
        lit = L.linear(L._simi_features(h_aug, q_aug, simi), W.numpy(), b.numpy())[..., 0]
        np.testing.assert_allclose(x.numpy(), lit, rtol=1e-11, atol=1e-12)


def test_amax_grad_splits_ties_like_tf_reduce_max():
    """[TF-internal] _MinOrMaxGrad divides the gradient equally among ties; torch.amax does too."""
    x = torch.tensor([[1.0, 3.0, 3.0, 0.0]], requires_grad=True, dtype=torch.float64)
    x.amax(dim=1).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), [[0, 0.5, 0.5, 0]])
