"""Shadow rows (bf16 engine): the context tensor of model_v2.py:863-914 is never stored in fp32 -- the focal attention
(model_v2.py:210-298) reads the bf16 half-rows the bi-LSTM keeps as its own MFMA operands, through a table of addresses.

  * fvta_lstm_desc.out_skip / fvta_lstm_shadow_rows / fvta_rows_from_shadow: the rows behind the table ARE bf16(out), the
    fp32 rows below out_skip are left alone, rows above it are stored as always;
  * fvta_attn_fwd_shadow / fvta_attn_bwd_shadow against fvta_attn_fwd (exact-fp32 kernel) / fvta_attn_bwd on the
    bf16-rounded rows: values, saved max-pooled logits, arg-max positions, every gradient;
  * the whole model with and without shadow rows: the same train step up to the bf16 rounding of the context rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = 1


def _close(a, b, rtol, atol, msg=""):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


@pytest.mark.parametrize("B,J,din,d,dense,share", [(64, 30, 200, 512, False, True), (70, 6, 200, 256, False, False),
                                                   (1500, 4, 200, 512, False, True), (33, 5, 12, 128, True, True),
                                                   (5, 6, 8, 32, False, True)])
def test_lstm_shadow_rows_are_the_bf16_output_rows(B, J, din, d, dense, share):
    from fvta_memexqa_amd import ops
    g = torch.Generator().manual_seed(B + J + d)
    x = torch.randn(B, J, din, generator=g).cuda()
    lens = torch.full((B,), J) if dense else torch.randint(0, J + 1, (B,), generator=g)
    lim = (6.0 / (din + 5 * d)) ** 0.5
    mk = lambda: (((torch.rand(din + d, 4 * d, generator=g) * 2 - 1) * lim * 2).cuda(), (torch.randn(4 * d, generator=g) * 0.1).cuda())
    kf, bf = mk()
    kb, bb = (None, None) if share else mk()
    ref_out, _, _ = ops.bilstm_simple(x, lens, kf, bf, kb, bb, training=True, precision=BF16)
    # the same call with the first `skip_seq` sequences' rows skipped (arena order = sequence order here)
    ar = torch.arange(B, dtype=torch.int64)
    wp = 2 * d
    for skip_seq in (B, B // 2):
        op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * wp, torch.full((B,), J, dtype=torch.int32), wp,
                        share_fw_bw=share, precision=BF16, training=True, out_skip=skip_seq * J * wp)
        op.make_plan(lens)
        out = torch.full((B, J, wp), 7.0, device="cuda")
        op.forward(x, out, kf, bf, kb, bb)
        nrows = skip_seq * J
        zero_half = torch.zeros(d, dtype=torch.bfloat16, device="cuda")
        table = torch.full((2, nrows), zero_half.data_ptr(), dtype=torch.int64, device="cuda")
        op.shadow_rows(table, nrows)
        got = torch.full((nrows, wp), 3.0, device="cuda")
        ops.rows_from_shadow(table, nrows, d, wp, got)
        torch.cuda.synchronize()
        want = ref_out.view(B * J, wp)[:nrows].bfloat16().float()      # (rows t >= len are zero in ref_out: the zero row)
        assert torch.equal(got, want), "shadow rows != bf16(out), skip %d" % skip_seq
        assert bool((out.view(B * J, wp)[:nrows] == 7.0).all()), "fp32 rows below out_skip were written"
        assert torch.equal(out.view(B * J, wp)[nrows:], ref_out.view(B * J, wp)[nrows:]), "rows above out_skip differ"
        # the backward does not read the fp32 rows: gradients bitwise those of the plain call's saved state
    with pytest.raises(Exception):
        ops.BiLstm(B, J, din, d, ar * J * din, ar * J * wp, torch.full((B,), J, dtype=torch.int32), wp, precision=0,
                   out_skip=wp).forward(x, torch.zeros(B, J, wp, device="cuda"), kf, bf)


def _shadow_table(hb, zero_rows, seed):
    """hb [R, w] bf16 rows -> (table int64 [2, R], keep-alive buffers): each direction's half-rows live in a buffer of
    their own in a shuffled order (as hs[dir][t][i] is); rows in `zero_rows` point at one shared zero half-row"""
    R, w = hb.shape
    g = torch.Generator().manual_seed(seed)
    keep, tabs = [], []
    for half in range(2):
        perm = torch.randperm(R, generator=g).cuda()
        buf = torch.empty(R + 1, w // 2, dtype=torch.bfloat16, device="cuda")
        buf[perm] = hb[:, half * (w // 2):(half + 1) * (w // 2)]
        buf[R].zero_()
        slot = perm.clone()
        slot[zero_rows] = R
        tabs.append(buf.data_ptr() + slot.to(torch.int64) * (w // 2) * 2)
        keep.append(buf)
    return torch.stack(tabs).contiguous(), keep


def test_attention_over_shadow_rows_randomised_differential(attn_select):
    from fvta_memexqa_amd import ops
    from test_gpu_forward import _att_case
    rng = np.random.RandomState(4242)
    cu = lambda t: None if t is None else t.cuda().contiguous()
    for case in range(16):
        N, K = int(rng.randint(1, 9)), int(rng.randint(1, 8))
        T = int(rng.choice([1, 5, 16, 17, 96, 150, 333, 700]))
        JQ = int(rng.choice([1, 9, 16, 17, 23, 32]))
        w = int(rng.choice([512, 1024]))
        simi = int(rng.choice([1, 2, 3]))
        tanh = bool(rng.rand() < 0.5)
        masked = bool(rng.rand() < 0.8)
        h, q, W, b, hm, qm = _att_case(N, K, T, JQ, w, simi, tanh, masked, seed=7000 + case, p_valid=float(rng.choice([0.1, 0.6, 0.95])))
        tag = "case %d (N %d K %d T %d JQ %d w %d simi %d tanh %s masked %s)" % (case, N, K, T, JQ, w, simi, tanh, masked)
        hb = cu(h).clamp(-1, 1).bfloat16()               # encoder outputs lie in (-1, 1)
        hmu = cu(ops.as_mask_u8(hm))
        # masked rows: zeros in the context tensor (dynamic_rnn's zero_output) -- the shared zero half-row
        if hmu is not None:
            hb = hb * hmu.view(N, K, T, 1).to(hb.dtype)
            zero_rows = (hmu.view(-1) == 0).nonzero().view(-1)
        else:
            zero_rows = torch.zeros(0, dtype=torch.int64, device="cuda")
        table, keep = _shadow_table(hb.view(N * K * T, w), zero_rows, case)
        h32 = hb.float().contiguous()
        op = ops.FocalAttention(N, K, T, JQ, w, simi, tanh)
        rest = (cu(q), hmu, cu(ops.as_mask_u8(qm)), cu(W.reshape(-1)), cu(b))
        sh = op.forward_shadow(table, *rest)
        sh, saved_sh = sh.cpu(), op.saved.clone()
        again = op.forward_shadow(table, *rest)
        assert torch.equal(again.cpu(), sh) and torch.equal(op.saved, saved_sh), tag + ": not reproducible"
        g = torch.randn(N, w, generator=torch.Generator().manual_seed(case)).cuda()
        # (NaN sentinels: accumulate = 0 overwrites -- every row must be written, valid or masked)
        mk = lambda: [torch.full((N, K, T, w), float("nan"), device="cuda"), torch.full((N, JQ, w), float("nan"), device="cuda"),
                      torch.zeros_like(rest[3]), torch.zeros(1, device="cuda")]
        grads_sh = mk()
        op.backward_shadow(table, *rest, g, *grads_sh, accumulate=0)
        assert torch.isfinite(grads_sh[0]).all() and torch.isfinite(grads_sh[1]).all(), tag + ": rows left unwritten"
        attn_select.exact()
        exact, _ = op.forward(h32, *rest)
        assert torch.isfinite(sh).all(), tag
        _close(sh, exact.cpu(), rtol=5e-5, atol=5e-6, msg=tag)
        nkt = N * K * T
        amax_s = saved_sh[:4 * nkt].view(torch.float32).cpu()
        amax_e = op.saved[:4 * nkt].view(torch.float32).cpu()
        ok = torch.isfinite(amax_e) & (amax_e > -1e29)
        np.testing.assert_allclose(amax_s[ok].numpy(), amax_e[ok].numpy(), rtol=5e-5, atol=5e-6, err_msg=tag + " amax")
        grads_ex = mk()
        op.backward(h32, *rest, g, *grads_ex, accumulate=0)
        for name, a_, b_ in zip(("d_hinfo", "d_hq", "dW", "db"), grads_sh, grads_ex):
            _close(a_, b_.cpu(), rtol=2e-4, atol=2e-5 * max(1.0, float(b_.abs().max())), msg=tag + " " + name)
        # accumulate = 2 (the model's mode): valid rows written, masked rows untouched
        d2 = torch.full((N, K, T, w), 5.0, device="cuda")
        dq2 = torch.zeros(N, JQ, w, device="cuda")
        op.forward_shadow(table, *rest)
        op.backward_shadow(table, *rest, g, d2, dq2, torch.zeros_like(rest[3]), torch.zeros(1, device="cuda"), accumulate=2)
        if hmu is not None and rest[2] is not None:
            valid = hmu.view(N, K, T).bool()
            allm = ~valid.any(2, keepdim=True) | ~rest[2].bool().any(1).view(N, 1, 1)   # fully masked streams: uniform over all T
            written = valid | allm
            assert bool((d2[~written.expand_as(valid)] == 5.0).all()), tag + ": masked rows written under accumulate=2"
            _close(d2[written.expand_as(valid)], grads_sh[0][written.expand_as(valid)].cpu(), rtol=0, atol=0, msg=tag + " accumulate=2")
    # shapes the shadow kernels do not take fail loudly
    op = ops.FocalAttention(2, 2, 8, 40, 512, 1, False)
    with pytest.raises(Exception):
        op.forward_shadow(torch.zeros(2, 32, dtype=torch.int64, device="cuda"), torch.zeros(2, 40, 512, device="cuda"), None, None,
                          torch.zeros(3 * 512, device="cuda"), torch.zeros(1, device="cuda"))


@pytest.mark.parametrize("cfgname,N", [("plumbing", None), ("plumbing_w512", None), ("metric", 2)])
def test_model_with_shadow_rows_is_the_same_train_step(cfgname, N):
    """precision = bf16 with and without shadow rows: the attention sees bf16(hall) instead of hall -- yp / loss within the
    bf16 engine's own tolerance against the oracle, gradients within a few percent of the plain bf16 run's (relative L2 per
    parameter), the vis tensor `hall` equal to bf16 of the plain run's, want_logits still served."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs, make_params
    # plumbing_w512: configs[0]'s sizes at hidden 256 (w = 512, the narrowest width the shadow kernels take; the plain
    # plumbing config, w = 256, checks that shapes outside them fall back to the fp32 rows -- and skips the comparison)
    cfg = dict(CONFIGS["plumbing"], d=256) if cfgname == "plumbing_w512" else dict(CONFIGS[cfgname])
    if N:
        cfg["N"] = N
    spec = SynthSpec(dense=False, **cfg)
    params, inputs = make_params(spec), make_inputs(spec)
    from tests._attn_rows_check import attention_param_grads_on_engine_rows
    res = {}
    for shadow in (False, True):
        model = Model(dict(spec.cfg(), batch_size=spec.N, precision="bf16", shadow_rows=shadow), text_in=spec.text_in, img_in=spec.img_in)
        model.set_oracle_params(params)
        L = model.load_inputs(inputs, training=True)
        if shadow and not L.shadow:
            pytest.skip("shape outside the shadow kernels (w = %d, JQ = %d)" % (model.wp, L.JQ))
        assert L.shadow == shadow
        model.zero_grad()
        yp = model.forward(L).cpu().double()
        model.backward(L, need_dx=True)
        grads = {k: torch.from_numpy(np.asarray(v)).double() for k, v in model.get_oracle_grads().items()}
        own = attention_param_grads_on_engine_rows(model, L)       # on this run's own rows and routing: tight
        assert max(own.values()) < 2e-3, "shadow=%s: att_logits gradients vs fp64 autograd on the engine's own rows: %r" % (shadow, own)
        hall = model.hall.clone()
        model.forward(L, want_logits=True)
        res[shadow] = dict(yp=yp, loss=model.loss.cpu().double(), grads=grads, hall=hall, att=model.att_logits.cpu(),
                           yp2=model.yp.cpu().double())
    a, b = res[True], res[False]
    assert torch.equal(a["hall"], b["hall"].bfloat16().float()), "hall"
    _close(a["yp"], b["yp"], rtol=0, atol=5e-3, msg="yp")
    _close(a["loss"], b["loss"], rtol=0, atol=5e-3, msg="loss")
    _close(a["yp2"], a["yp"], rtol=0, atol=1e-4, msg="yp through the fp32-row kernel on the filled-in rows")
    _close(a["att"], b["att"], rtol=0, atol=2e-2, msg="att_logits")
    for k, gb in b["grads"].items():
        ga = a["grads"][k]
        if float(gb.norm()) < 1e-6:      # out_b: sum_c (softmax - y) = 0 when every row has a label -- rounding noise on both sides
            assert float(ga.abs().max()) < 1e-5, k
            continue
        err = float((ga - gb).norm() / (gb.norm() + 1e-30))
        assert err < (0.2 if "att_logits" in k else 4e-2), "grad %s: relative L2 %.4f" % (k, err)
