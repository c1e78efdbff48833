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


def test_dmn_generate_episode_two_hops():
    """model_dmnplus.py:113-136 `_generate_episode` (the loop the AttentionGRUCell lives in): attention MLP over the facts,
    softmax over all facts, gated recurrence with per-row lengths; a second hop reuses the variables."""
    from fvta_memexqa_amd import functional as Fn
    from oracle import fvta_literal as L
    Fn.reset_default_graph()
    g = torch.Generator().manual_seed(9)
    N, F, d = 5, 11, 64
    facts = torch.randn(N, F, d, generator=g) * 0.5
    q = torch.randn(N, d, generator=g) * 0.5
    lens = torch.tensor([11, 7, 1, 4, 11])
    mem = q.clone()
    eps = []
    for hop in range(2):
        ep = Fn.generate_episode(mem.cuda(), q.cuda(), facts.cuda(), lens, hop, d, scope="memory")
        eps.append(ep.cpu())
        mem = ep.cpu()
    v = {k: t.cpu().double().numpy() for k, t in Fn.variables.items()}
    p = dict(fc1_W=v["memory/attention/fc1/weights"], fc1_b=v["memory/attention/fc1/biases"],
             fc2_W=v["memory/attention/fc2/weights"], fc2_b=v["memory/attention/fc2/biases"],
             Wg=v["memory/attention_gru/attention_gru_cell/gates/weights"], bg=v["memory/attention_gru/attention_gru_cell/gates/biases"],
             Wc=v["memory/attention_gru/attention_gru_cell/candidate/weights"],
             Wi=v["memory/attention_gru/attention_gru_cell/input/weights"], bi=v["memory/attention_gru/attention_gru_cell/input/biases"])
    m64 = q.double().numpy()
    for hop in range(2):
        ref = L.dmn_generate_episode(m64, q.double().numpy(), facts.double().numpy(), lens.numpy(), p)
        np.testing.assert_allclose(eps[hop].numpy(), ref, rtol=2e-4, atol=2e-5)
        m64 = ref
    assert len([k for k in Fn.variables if k.startswith("memory/")]) == 9          # the second hop created nothing new


def test_dmn_episodic_memory_forward_backward():
    """model_dmnplus.py:503-516: the hop loop (shared attention MLP + AttentionGRU, one dense+relu per hop) -- output and
    every gradient (question, facts, all 13 variables) against autograd of the fp64 restatement; rows of different lengths,
    one of length 1."""
    from fvta_memexqa_amd.dmn import EpisodicMemory
    from oracle import fvta_fused as O
    g = torch.Generator().manual_seed(21)
    N, F, d, hops = 6, 9, 64, 3
    facts = torch.randn(N, F, d, generator=g) * 0.5
    gq = torch.randn(N, d, generator=g) * 0.5
    lens = torch.tensor([9, 5, 1, 9, 3, 7])
    d_out = torch.randn(N, d, generator=g)
    mem = EpisodicMemory(d, hops, seed=3)
    for k in mem.params:                                   # non-zero biases so that their gradients are exercised
        if k.endswith("biases") or k.endswith("bias"):
            mem.params[k].copy_(torch.randn(mem.params[k].shape, generator=g) * 0.1)
    out = mem(gq.cuda(), facts.cuda(), lens)
    d_gq, d_facts = mem.backward(d_out.cuda())
    p64 = {k: v.cpu().double().requires_grad_(True) for k, v in mem.params.items()}
    gq64, f64 = gq.double().requires_grad_(True), facts.double().requires_grad_(True)
    ref = O.dmn_memory(gq64, f64, lens, p64, hops)
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    (ref * d_out.double()).sum().backward()

    def close(a, b, tag):
        a, b = a.cpu().double().numpy(), b.numpy()
        # (fc2's bias shifts every logit of the softmax alike: its true gradient is 0, hence the absolute term)
        err, ref = np.linalg.norm(a - b), np.linalg.norm(b)
        assert err <= 2e-4 * ref + 1e-5, "%s: |a-b| %.3g, |b| %.3g" % (tag, err, ref)

    close(d_gq, gq64.grad, "d_gq")
    close(d_facts, f64.grad, "d_facts")
    for k in mem.params:
        close(mem.grads[k], p64[k].grad, k)
    # a second backward accumulates into the parameter gradients
    before = {k: v.clone() for k, v in mem.grads.items()}
    mem.backward(d_out.cuda())
    for k in mem.grads:
        assert torch.allclose(mem.grads[k], 2 * before[k], rtol=1e-5, atol=1e-6), k
