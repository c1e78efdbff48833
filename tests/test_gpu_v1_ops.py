"""The pieces fvta_memexqa_amd.model (the model.py graph) adds to the C ABI: the strided 1-D attention
(fvta_attn_desc.hinfo_stride: one stream of an [N][all streams] arena, model.py:836-846 / :929-944) and the two shape ops
(fvta_rows_reduce / fvta_rows_broadcast = reduce_mean / tile and each other's backward)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,simi", [(64, 1), (128, 2), (256, 3)])
@pytest.mark.parametrize("masked", [True, False])
def test_strided_attention_equals_dense_on_the_gathered_stream(w, simi, masked):
    """forward and backward on a slice [N, V, w] of a wider [N, Vtot, w] arena == the same kernel on a dense copy
    (bitwise: same kernel, same order), in overwrite and accumulate mode; rows outside the slice are never touched"""
    from fvta_memexqa_amd import ops
    dev = ops.require_gpu()
    g = torch.Generator().manual_seed(5)
    N, V, off, Vtot, JQ = 5, 37, 11, 70, 9
    F = {1: 3, 2: 2, 3: 4}[simi]
    arena = torch.randn(N, Vtot, w, generator=g).to(dev)
    hq = torch.randn(N, JQ, w, generator=g).to(dev)
    W = (torch.randn(F * w, generator=g) * 0.1).to(dev)
    b = torch.zeros(1, device=dev)
    hm = (torch.rand(N, V, generator=g) > 0.3).to(torch.uint8).to(dev) if masked else None
    qm = (torch.rand(N, JQ, generator=g) > 0.2).to(torch.uint8).to(dev) if masked else None
    if masked:
        hm[:, 0] = 1
        qm[:, 0] = 1
    dense_h = arena[:, off:off + V].contiguous()
    op_d = ops.FocalAttention(N, 1, V, JQ, w, simi, False, feat_order=1)
    op_s = ops.FocalAttention(N, 1, V, JQ, w, simi, False, feat_order=1, hinfo_stride=Vtot * w)
    ha_d, lg_d = op_d.forward(dense_h, hq, hm, qm, W, b, True)
    ha_s, lg_s = op_s.forward(arena.view(-1)[off * w:], hq, hm, qm, W, b, True)
    assert torch.equal(ha_d, ha_s) and torch.equal(lg_d, lg_s)
    d_ha = torch.randn(N, w, generator=g).to(dev)
    for acc in (0, 1):
        base = torch.randn(N, Vtot, w, generator=g).to(dev)
        d_dense = base[:, off:off + V].contiguous()
        d_hq_d = torch.ones(N, JQ, w, device=dev)
        dW_d, db_d = torch.zeros(F * w, device=dev), torch.zeros(1, device=dev)
        op_d.backward(dense_h, hq, hm, qm, W, b, d_ha, d_dense, d_hq_d, dW_d, db_d, accumulate=acc)
        d_arena = base.clone()
        d_hq_s = torch.ones(N, JQ, w, device=dev)
        dW_s, db_s = torch.zeros(F * w, device=dev), torch.zeros(1, device=dev)
        op_s.backward(arena.view(-1)[off * w:], hq, hm, qm, W, b, d_ha, d_arena.view(-1)[off * w:], d_hq_s, dW_s, db_s,
                      accumulate=acc)
        assert torch.equal(d_arena[:, off:off + V], d_dense), acc
        assert torch.equal(d_arena[:, :off], base[:, :off]) and torch.equal(d_arena[:, off + V:], base[:, off + V:]), acc
        assert torch.equal(d_hq_d, d_hq_s) and torch.equal(dW_d, dW_s) and torch.equal(db_d, db_s)


def test_strided_attention_argument_checks():
    from fvta_memexqa_amd import _lib, ops
    ops.require_gpu()
    with pytest.raises(_lib.FvtaError):
        ops.FocalAttention(2, 3, 8, 4, 64, 1, False, hinfo_stride=8 * 64 * 3)      # K must be 1
    with pytest.raises(_lib.FvtaError):
        ops.FocalAttention(2, 1, 8, 4, 64, 1, False, hinfo_stride=4 * 64)          # shorter than the stream


def test_rows_reduce_and_broadcast():
    from fvta_memexqa_amd import ops
    dev = ops.require_gpu()
    g = torch.Generator().manual_seed(9)
    R, J, d, K = 7, 5, 300, 3
    x = torch.randn(R, J, d, generator=g).to(dev)
    out = torch.randn(R, K, d, generator=g).to(dev)
    want = out.clone()
    want[:, 1] = x.mean(1)
    ops.rows_reduce(x, out.view(-1)[d:], R, J, d, K * d, 1.0 / J)                   # into slot 1 of a [R,K,d] stack
    torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-6)
    ops.rows_reduce(x, out.view(-1)[d:], R, J, d, K * d, 2.0, accumulate=True)
    want[:, 1] += 2.0 * x.sum(1)
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    v = torch.randn(R, K, d, generator=g).to(dev)
    y = torch.randn(R, J, d, generator=g).to(dev)
    wanty = v[:, 2][:, None, :].expand(R, J, d) * 0.25
    ops.rows_broadcast(v.view(-1)[2 * d:], y, R, J, d, K * d, 0.25)
    assert torch.equal(y, wanty.contiguous())
    ops.rows_broadcast(v.view(-1)[2 * d:], y, R, J, d, K * d, 1.0, accumulate=True)
    torch.testing.assert_close(y, wanty + v[:, 2][:, None, :], rtol=1e-6, atol=1e-6)
