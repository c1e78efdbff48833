"""GPU parity of AttentionGRUCell (attention_gru_cell.py:50-70) forward and backward vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,d", [(7, 16), (130, 80), (64, 100)])
def test_attention_gru_cell_forward_backward(B, d):
    from fvta_memexqa_amd import ops
    from fvta_memexqa_amd.attention_gru_cell import AttentionGRUCell
    from oracle import fvta_fused as F
    g = torch.Generator().manual_seed(B + d)
    inputs = torch.randn(B, d + 1, generator=g)
    inputs[:, d] = torch.rand(B, generator=g)          # attention gate in [0,1]
    state = torch.randn(B, d, generator=g) * 0.5
    P = [torch.randn(2 * d, d, generator=g) * 0.2, torch.randn(d, generator=g) * 0.1, torch.randn(d, d, generator=g) * 0.2,
         torch.randn(d, d, generator=g) * 0.2, torch.randn(d, generator=g) * 0.1]
    gout = torch.randn(B, d, generator=g)
    leaves = [t.double().requires_grad_() for t in [inputs, state] + P]
    ref = F.attention_gru_cell(*leaves)
    (ref * gout.double()).sum().backward()
    cu = lambda t: t.cuda().contiguous()
    cell = AttentionGRUCell(d)
    params = {"gates/weights": cu(P[0]), "gates/biases": cu(P[1]), "candidate/weights": cu(P[2]),
              "input/weights": cu(P[3]), "input/biases": cu(P[4])}
    new_h, new_state = cell(cu(inputs), cu(state), params)
    np.testing.assert_allclose(new_h.cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    assert new_state is new_h
    z = lambda t: torch.zeros_like(cu(t))
    dWg, dbg, dWc, dWi, dbi = z(P[0]), z(P[1]), z(P[2]), z(P[3]), z(P[4])
    d_in, d_st = ops.attgru_bwd(cu(inputs), cu(state), params["gates/weights"], params["candidate/weights"],
                                params["input/weights"], cell.saved, cu(gout), dWg, dbg, dWc, dWi, dbi)
    for got, want, name in [(d_in, leaves[0].grad, "d_inputs"), (d_st, leaves[1].grad, "d_state"), (dWg, leaves[2].grad, "dWg"),
                            (dbg, leaves[3].grad, "dbg"), (dWc, leaves[4].grad, "dWc"), (dWi, leaves[5].grad, "dWi"),
                            (dbi, leaves[6].grad, "dbi")]:
        w = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), w, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(w).max()), err_msg=name)
    with pytest.raises(ValueError):
        cell(cu(inputs)[:, :d], cu(state), params)
