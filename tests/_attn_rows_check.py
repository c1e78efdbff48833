"""The focal attention's OWN parameter gradients (attention/all/att_logits/{W,b}, model_v2.py:1016-1020) checked on the rows
the engine itself produced.

Between two engines these two gradients can only be held to a loose bound: they hang off arg-max positions (max over j,
model_v2.py:268; max over t, :278), which move where the engines' context rows -- a few 1e-3 apart -- meet a near-tie.  Here
the routing question is taken out: the fp64 oracle's attention_3d runs on the context rows, question rows and upstream
gradient d loss / d g1 of THIS engine's step (read back from the device), with the kernels' tie convention
(max_grad="first"), album chunk by album chunk, and its d W / d b must equal the engine's at 2e-3 relative L2 -- fifty times
below the cross-engine bound, so an error in attn_bwd_params_kernel cannot hide behind "arg-max routing"."""
import numpy as np
import torch


def attention_param_grads_on_engine_rows(model, L, chunk=4):
    """-> {"W": rel-L2, "b": rel-L2} of the model's att_logits gradients (after forward + backward on L, with zero_grad
    before) against fp64 autograd on the engine's own rows.  No time warp, simiMatrix 1-3."""
    from oracle import fvta_fused as F
    assert not model.use_time_warp and model.simi in (1, 2, 3)
    N, K, T, JQ, wp = L.N, L.K, L.T, L.JQ, model.wp
    hall = model.hall.view(N, K, T, wp)                         # (under shadow rows: fp32 of the bf16 rows, exactly)
    hq = L.hq.view(N, JQ, wp)
    hm = L.hall_mask.view(N, K, T).bool().cpu()
    qm = L.q_mask.view(N, JQ).bool().cpu()
    dg1 = L.dg1.view(N, wp).cpu().double()
    W = model.params.view(model.N_ATT_W).detach().cpu().double().reshape(-1, 1).requires_grad_()
    b = model.params.view(model.N_ATT_B).detach().cpu().double().reshape(1).requires_grad_()
    for n0 in range(0, N, chunk):
        sl = slice(n0, min(N, n0 + chunk))
        ha, _ = F.attention_3d(hall[sl].cpu().double(), hq[sl].cpu().double(), W, b, hm[sl], qm[sl], simiMatrix=model.simi,
                               add_tanh=model.add_tanh, max_grad="first")
        (ha * dg1[sl]).sum().backward()
    gW = model.params.view(model.N_ATT_W, True).detach().cpu().double().reshape(-1)
    gb = model.params.view(model.N_ATT_B, True).detach().cpu().double().reshape(-1)
    rel = lambda a, r: float((a - r).norm() / (r.norm() + 1e-30))
    return {"W": rel(gW, W.grad.reshape(-1)), "b": rel(gb, b.grad.reshape(-1))}
