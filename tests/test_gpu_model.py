"""GPU parity of the whole hot path (encoders -> context tensor -> attention -> scorer -> loss,
forward AND backward AND one optimiser step) through the Model / Trainer / Tester mirror,
against the CPU oracle.  Tolerance 1e-4 relative fp32, answer argmax bit-exact (north_star)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-4, atol=1e-5, msg=""):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale, err_msg=msg)


def _run(spec, check_grads=True):
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import make_inputs, make_params, to_dtype
    from oracle import fvta_fused as F
    params = make_params(spec)
    inputs = make_inputs(spec)
    cfg = spec.cfg()
    # ---- oracle (fp64, autograd)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    ref = F.fvta_forward(p64, to_dtype(inputs, torch.float64), cfg)
    ref["loss"].backward()
    # ---- HIP path
    model = Model(dict(cfg, batch_size=spec.N), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L, want_logits=True)
    d, dp = model.d, model.dp
    unpad = lambda t: torch.cat([t[..., :d], t[..., dp:dp + d]], -1)
    _close(unpad(model.hall).reshape(ref["hall"].shape), ref["hall"], msg="hall")
    _close(unpad(L.hq), ref["hq"], msg="hq")
    _close(unpad(L.lch), ref["lchoices"], msg="lchoices")
    _close(unpad(L.g1), ref["g1_all"], msg="g1_all")
    _close(unpad(L.gq), ref["gq"], msg="gq")
    _close(model.att_logits, ref["att_logits"], atol=2e-5, msg="att_logits")
    _close(model.logits, ref["logits"], msg="logits")
    _close(yp, ref["yp"], msg="yp")
    _close(model.loss, ref["loss"].reshape(1), msg="loss")
    assert (yp.argmax(1).cpu() == ref["yp"].argmax(1)).all(), "answer argmax must be bit-exact"
    if check_grads:
        model.backward(L, need_dx=True)
        grads = model.get_oracle_grads()
        for k, v in p64.items():
            if v.grad is None:      # e.g. qatt_W when use_question_att is off
                continue
            _close(grads[k].reshape(v.grad.shape), v.grad, rtol=2e-4, atol=2e-5, msg="grad " + k)
    return model, L, ref


@pytest.mark.parametrize("simi,tanh,qatt,share", [(2, True, True, True), (1, False, False, False), (3, True, True, True)])
@pytest.mark.parametrize("dense", [False, True])
def test_small_model_forward_backward(simi, tanh, qatt, share, dense):
    from fvta_memexqa_amd.synth import SynthSpec
    spec = SynthSpec(N=3, A=2, P=3, S=2, L=5, d=32, SA=1, dense=dense, simiMatrix=simi, add_tanh=tanh,
                     use_question_att=qatt, share_fw_bw=share, text_in=12, img_in=8)
    _run(spec)


def test_model_with_input_dropout():
    """--keep_prob < 1 (DropoutWrapper on both cells while training, model_v2.py:657-661): a training step's loss, answers
    and every parameter gradient against the oracle run with the SAME keep masks (the library's counter-based hash,
    restated in oracle.fvta_fused.dropout_keep_masks, sliced per stream out of each cell's input arena); the evaluation
    layout does not drop; a second step draws new masks."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params, to_dtype
    from oracle import fvta_fused as F
    spec = SynthSpec(N=3, A=2, P=3, S=2, L=5, d=32, SA=1, dense=False, simiMatrix=2, add_tanh=True,
                     use_question_att=True, share_fw_bw=True, text_in=12, img_in=8)
    params, inputs, cfg, keep = make_params(spec), make_inputs(spec), spec.cfg(), 0.75
    model = Model(dict(cfg, batch_size=spec.N, keep_prob=keep, dropout_seed=5), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L)
    loss1 = float(model.loss)
    model.backward(L, need_dx=True)

    def keep_of(name, x):
        for G in L.groups.values():
            for sg in G.segs:
                if sg["name"] == name:
                    km = F.dropout_keep_masks(G.x.numel(), keep, G.drop_seed)
                    n = sg["count"] * sg["J"] * G.din
                    km = km[:, sg["x_elem0"]:sg["x_elem0"] + n].reshape(2, sg["count"], sg["J"], G.din)
                    return km[..., :x.shape[-1]].reshape(2, *x.shape)
        raise KeyError(name)

    inp = to_dtype(inputs, torch.float64)
    inp["q"]["keep"] = keep_of("q", inputs["q"]["x"])
    inp["choices"]["keep"] = keep_of("choices", inputs["choices"]["x"])
    for k, st in enumerate(inp["ctx"]):
        st["keep"] = keep_of("ctx%d" % k, inputs["ctx"][k]["x"])
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    ref = F.fvta_forward(p64, inp, dict(cfg, keep_prob=keep))
    ref["loss"].backward()
    _close(yp, ref["yp"], msg="yp")
    _close(model.loss, ref["loss"].reshape(1), msg="loss")
    grads = model.get_oracle_grads()
    for k, v in p64.items():
        if v.grad is not None:
            _close(grads[k].reshape(v.grad.shape), v.grad, rtol=2e-4, atol=2e-5, msg="grad " + k)
    # evaluation: no dropout
    ref_eval = F.fvta_forward({k: v.double() for k, v in params.items()}, to_dtype(inputs, torch.float64), cfg)
    Le = model.load_inputs(inputs, training=False)
    _close(model.forward(Le), ref_eval["yp"], msg="yp (evaluation layout)")
    assert abs(float(ref_eval["loss"]) - loss1) > 1e-6
    # the next training step draws other masks
    L = model.load_inputs(inputs, training=True)
    model.forward(L)
    assert abs(float(model.loss) - loss1) > 1e-7
    with pytest.raises(ValueError):
        Model(dict(cfg, batch_size=spec.N, keep_prob=0.0), text_in=spec.text_in, img_in=spec.img_in)


def test_hidden_size_padding_is_exact():
    """hidden_size 20 (w=40) runs at the padded width 64; results equal the unpadded oracle."""
    from fvta_memexqa_amd.synth import SynthSpec
    spec = SynthSpec(N=2, A=1, P=3, S=2, L=4, d=20, dense=False, text_in=12, img_in=8)
    model, L, _ = _run(spec)
    assert model.dp == 32 and model.wp == 64


def test_plumbing_config():
    """BASELINE.json configs[0]: 2 albums x 5 photos x 2 text streams x 10 tokens, hidden 128, batch 4."""
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec
    _run(SynthSpec(dense=False, **CONFIGS["plumbing"]))


def test_cosine_similarity_model():
    """simiMatrix 4 (README.MD:87 ablation): no att_logits parameters, forward and backward."""
    from fvta_memexqa_amd.synth import SynthSpec
    spec = SynthSpec(N=2, A=1, P=3, S=2, L=4, d=32, dense=False, simiMatrix=4, add_tanh=False, text_in=12, img_in=8)
    _run(spec)


def test_trainer_tester_mirror_and_update():
    """Trainer.step -> (loss, None, None) and one Adadelta update equal to the oracle's;
    Tester.step -> yp[:num_examples]."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params, to_dtype
    from fvta_memexqa_amd.tester import Tester
    from fvta_memexqa_amd.trainer import Trainer
    from oracle import fvta_fused as F
    from oracle import fvta_literal as Lit
    spec = SynthSpec(N=4, A=1, P=3, S=2, L=5, d=32, dense=False, text_in=12, img_in=8)
    params, inputs, cfg = make_params(spec), make_inputs(spec), spec.cfg()
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    ref = F.fvta_forward(p64, to_dtype(inputs, torch.float64), cfg)
    ref["loss"].backward()
    model = Model(dict(cfg, batch_size=spec.N, init_lr=0.5), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    trainer = Trainer(model, dict(init_lr=0.5))
    batch = dict(inputs, num_examples=3)
    loss, summary, train_op = trainer.step(None, (None, batch))
    assert summary is None and train_op is None
    _close(torch.tensor([loss]), ref["loss"].reshape(1))
    assert model.global_step == 1
    new = model.get_weights()
    names = {"text_kernel": Model.N_TEXT_K % "fw", "att_W": Model.N_ATT_W, "out_W": Model.N_OUT_W, "qatt_W": Model.N_QATT_W,
             "image_kernel": Model.N_IMG_K % "fw", "out_b": Model.N_OUT_B}
    for k, name in names.items():
        v = p64[k]
        exp, _, _ = Lit.adadelta_step(v.detach().numpy(), v.grad.numpy(), np.zeros_like(v.detach().numpy()),
                                      np.zeros_like(v.detach().numpy()), 0.5)
        _close(new[name].reshape(exp.shape), exp, rtol=2e-4, atol=2e-5, msg="updated " + k)
    # tester on the ORIGINAL weights
    model.set_oracle_params(params)
    yp = Tester(model, None).step(None, (None, batch))
    assert yp.shape == (3, 4)
    _close(yp, ref["yp"][:3])


def test_metric_shape_forward_vs_oracle_subset():
    """BASELINE.json configs[1] shape (batch 64, 40 photos x 5 streams x 30 tok, h=512), ragged:
    full forward on the GPU; the fp32 CPU oracle checks the first 2 QA pairs (it is batch-independent)."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs, make_params
    from oracle import fvta_fused as F
    spec = SynthSpec(dense=False, **dict(CONFIGS["metric"], N=8))
    params, inputs = make_params(spec), make_inputs(spec)
    model = Model(dict(spec.cfg(), batch_size=spec.N), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs)
    yp = model.forward(L)
    n = 2
    sub = dict(ctx=[dict(x=s["x"][:n], mask=s["mask"][:n], cell=s["cell"]) for s in inputs["ctx"]],
               q=dict(x=inputs["q"]["x"][:n], mask=inputs["q"]["mask"][:n]),
               choices=dict(x=inputs["choices"]["x"][:n], mask=inputs["choices"]["mask"][:n]), y=inputs["y"][:n])
    with torch.no_grad():
        ref = F.fvta_forward(params, sub, spec.cfg())
    _close(L.g1[:n], ref["g1_all"], rtol=1e-4, atol=2e-5, msg="g1")
    _close(yp[:n], ref["yp"], rtol=1e-4, atol=1e-5, msg="yp")
    assert (yp[:n].argmax(1).cpu() == ref["yp"].argmax(1)).all()


@pytest.mark.parametrize("warp_type", [1, 5])
def test_model_with_time_warp(warp_type):
    """The published FVTA flag set adds --use_time_warp (README.MD:144-147): forward and gradients vs the oracle."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params, to_dtype
    from oracle import fvta_fused as F
    spec = SynthSpec(N=3, A=2, P=3, S=2, L=5, d=32, SA=1, dense=False, text_in=12, img_in=8)
    params, inputs = make_params(spec), make_inputs(spec)
    g = torch.Generator().manual_seed(17)
    w = spec.w
    params.update(WH_W=torch.randn(2 * w, w, generator=g) * 0.05, WH_b=torch.randn(w, generator=g) * 0.05,
                  WC_W=torch.randn(w, 1, generator=g) * 0.1, WC_b=torch.randn(1, generator=g) * 0.05)
    cfg = dict(spec.cfg(), use_time_warp=True, warp_type=warp_type, window_t=1.4)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    ref = F.fvta_forward(dict(p64, window_t=1.4), to_dtype(inputs, torch.float64), cfg)
    ref["loss"].backward()
    model = Model(dict(cfg, batch_size=spec.N), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L)
    model.backward(L)
    _close(yp, ref["yp"], msg="yp")
    _close(model.warp_h.reshape(ref["hall"].shape), ref["hall"], msg="warp_h")
    grads = model.get_oracle_grads()
    for k, v in p64.items():
        if v.grad is not None:
            _close(grads[k].reshape(v.grad.shape), v.grad, rtol=2e-4, atol=2e-5, msg="grad " + k)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_forward_backward_bitwise_reproducible(precision):
    """Whole path, ragged metric-shaped batch (N = 4): the context tensor, g1, yp and the flat gradient buffer are
    bitwise equal across repeated steps on the same inputs (no atomics on the value path, fixed reduction orders,
    counted-wait pipelines without races)."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs, make_params
    spec = SynthSpec(dense=False, **dict(CONFIGS["metric"], N=4))
    params, inputs = make_params(spec), make_inputs(spec)
    model = Model(dict(spec.cfg(), batch_size=spec.N, precision=precision), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    ref = None
    for _ in range(4):
        model.zero_grad()
        yp = model.forward(L)
        model.backward(L)
        torch.cuda.synchronize()
        cur = (L.arena.clone(), L.g1.clone(), yp.clone(), model.params.grad.clone())
        if ref is None:
            ref = cur
        else:
            for a, b, name in zip(cur, ref, ("arena", "g1", "yp", "grad")):
                assert torch.equal(a, b), name


@pytest.mark.parametrize("use_char,use_image_trans", [(True, True), (False, False)])
def test_token_id_entry_matches_oracle(use_char, use_image_trans):
    """the model entered with the reference's own feed (word / char ids, photo indices):
    yp, loss and the gradients of the embedding parameters vs the oracle (embed_inputs -> fvta_forward)."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_embed_params, make_params, make_token_inputs
    from oracle import fvta_fused as F
    VW, VF, VC, W, cd, cw, wd, idim, tdim = 40, 60, 30, 12, 8, 24 if use_char else 0, 20, 57, 16 if use_image_trans else 57
    spec = SynthSpec(N=3, A=2, P=3, S=2, L=5, d=32, SA=1, dense=False, text_in=wd + cw, img_in=tdim)
    params = dict(make_params(spec), **make_embed_params(spec, VW, VF, VC, cd, cw, wd, idim, tdim, use_image_trans))
    tok = make_token_inputs(spec, VW, VF, VC, W)
    g = torch.Generator().manual_seed(3)
    tok["image_emb_mat"] = torch.randn(tok["n_image_rows"], idim, generator=g) * 0.5
    cfg = dict(spec.cfg(), batch_size=spec.N, word_vocab_size=VW, word_emb_size=wd, use_char=use_char, char_vocab_size=VC,
               max_word_size=W, char_emb_size=cd, char_out_size=cw, image_feat_dim=idim, use_image_trans=use_image_trans,
               image_trans_dim=tdim)
    p64 = {k: (v.double().requires_grad_() if k != "existing_emb_mat" else v.double()) for k, v in params.items()}
    ref = F.fvta_forward(p64, F.embed_inputs(p64, tok, cfg), cfg)
    ref["loss"].backward()
    model = Model(cfg)
    model.set_oracle_params(params)
    L = model.load_inputs(tok, training=True)
    model.zero_grad()
    yp = model.forward(L)
    _close(yp, ref["yp"], msg="yp")
    _close(model.loss, ref["loss"].reshape(1), msg="loss")
    model.backward(L)
    grads = model.get_oracle_grads()
    keys = ["word_emb", "text_kernel", "out_W"] + (["char_emb", "conv_filter", "conv_bias"] if use_char else []) \
        + (["img_W", "img_b"] if use_image_trans else [])
    for k in keys:
        exp = p64[k].grad.numpy()
        _close(np.asarray(grads[k]).reshape(exp.shape), exp, rtol=2e-4, atol=2e-6, msg="grad " + k)


def test_token_id_entry_with_dropout():
    """--keep_prob < 1 on the reference's own feed: conv1d's dropout of the char embeddings (model_v2.py:58-62) AND the
    cells' input dropout (:657-661) in one training step -- loss, answers and the gradients of the embedding, encoder and
    scorer parameters against the oracle run with the same masks."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_embed_params, make_params, make_token_inputs
    from oracle import fvta_fused as F
    VW, VF, VC, W, cd, cw, wd, idim, tdim, keep = 40, 60, 30, 12, 8, 24, 20, 57, 16, 0.8
    spec = SynthSpec(N=3, A=2, P=3, S=2, L=5, d=32, SA=1, dense=False, text_in=wd + cw, img_in=tdim)
    params = dict(make_params(spec), **make_embed_params(spec, VW, VF, VC, cd, cw, wd, idim, tdim, True))
    tok = make_token_inputs(spec, VW, VF, VC, W)
    g = torch.Generator().manual_seed(3)
    tok["image_emb_mat"] = torch.randn(tok["n_image_rows"], idim, generator=g) * 0.5
    cfg = dict(spec.cfg(), batch_size=spec.N, word_vocab_size=VW, word_emb_size=wd, use_char=True, char_vocab_size=VC,
               max_word_size=W, char_emb_size=cd, char_out_size=cw, image_feat_dim=idim, use_image_trans=True,
               image_trans_dim=tdim, keep_prob=keep, dropout_seed=11)
    model = Model(cfg)
    model.set_oracle_params(params)
    L = model.load_inputs(tok, training=True)
    model.zero_grad()
    yp = model.forward(L)
    model.backward(L)
    T_ = L.groups["text"]
    ckeep = F.dropout_keep_flat(T_.ntok * W * cd, keep, T_.char_drop_seed)

    def masks_of(name, st, cell):
        G = L.groups[cell]
        sg = next(x for x in G.segs if x["name"] == name)
        lead = tuple(st["ids"].shape[:-1]) if "ids" in st else tuple(st["pis"].shape[:-1])
        J = sg["J"]
        km = F.dropout_keep_masks(G.x.numel(), keep, G.drop_seed)
        n = sg["count"] * J * G.din
        true_in = (wd + cw) if cell == "text" else tdim
        st["keep"] = km[:, sg["x_elem0"]:sg["x_elem0"] + n].reshape(2, sg["count"], J, G.din)[..., :true_in] \
            .reshape((2,) + lead + (J, true_in))
        if cell == "text":
            m = sg["count"] * J
            st["char_keep"] = ckeep[sg["tok0"] * W * cd:(sg["tok0"] + m) * W * cd].reshape(lead + (J, W, cd))

    masks_of("q", tok["q"], "text")
    masks_of("choices", tok["choices"], "text")
    for k, st in enumerate(tok["ctx"]):
        masks_of("ctx%d" % k, st, st.get("cell", "text"))
    p64 = {k: (v.double().requires_grad_() if k != "existing_emb_mat" else v.double()) for k, v in params.items()}
    ref = F.fvta_forward(p64, F.embed_inputs(p64, tok, cfg), cfg)
    ref["loss"].backward()
    _close(yp, ref["yp"], msg="yp")
    _close(model.loss, ref["loss"].reshape(1), msg="loss")
    grads = model.get_oracle_grads()
    for k in ["word_emb", "text_kernel", "out_W", "char_emb", "conv_filter", "conv_bias", "img_W", "img_b"]:
        exp = p64[k].grad.numpy()
        _close(np.asarray(grads[k]).reshape(exp.shape), exp, rtol=2e-4, atol=2e-6, msg="grad " + k)


@pytest.mark.gpu
@pytest.mark.parametrize("token", [False, True])
def test_weight_decay_matches_oracle(token):
    """--wd (model_v2.py:347-354): the l2 terms in the loss and their gradients, incl. the 7x cover of the shared char-CNN
    filter; yp is unaffected."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_embed_params, make_inputs, make_params, make_token_inputs
    from oracle import fvta_fused as F
    wd = 0.002
    if token:
        VW, VF, VC, W, cd, cw, wdim, idim, tdim = 40, 60, 30, 12, 8, 24, 20, 57, 16
        spec = SynthSpec(N=3, A=2, P=3, S=2, L=5, d=32, SA=1, dense=False, text_in=wdim + cw, img_in=tdim)
        params = dict(make_params(spec), **make_embed_params(spec, VW, VF, VC, cd, cw, wdim, idim, tdim, True))
        inputs = make_token_inputs(spec, VW, VF, VC, W)
        inputs["image_emb_mat"] = torch.randn(inputs["n_image_rows"], idim, generator=torch.Generator().manual_seed(3)) * 0.5
        cfg = dict(spec.cfg(), batch_size=spec.N, word_vocab_size=VW, word_emb_size=wdim, use_char=True, char_vocab_size=VC,
                   max_word_size=W, char_emb_size=cd, char_out_size=cw, image_feat_dim=idim, use_image_trans=True,
                   image_trans_dim=tdim, wd=wd)
    else:
        spec = SynthSpec(N=3, A=2, P=3, S=2, L=5, d=32, SA=1, dense=False, text_in=12, img_in=8)
        params, inputs = make_params(spec), make_inputs(spec)
        cfg = dict(spec.cfg(), batch_size=spec.N, wd=wd)
    p64 = {k: (v.double().requires_grad_() if k != "existing_emb_mat" else v.double()) for k, v in params.items()}
    from fvta_memexqa_amd.synth import to_dtype
    oin = F.embed_inputs(p64, inputs, cfg) if token else to_dtype(inputs, torch.float64)
    ref = F.fvta_forward(p64, oin, cfg)
    ref0 = F.fvta_forward(p64, oin, dict(cfg, wd=None))
    assert float(ref["loss"].detach()) > float(ref0["loss"].detach()) + 1e-3          # the terms are not negligible in this test
    ref["loss"].backward()
    model = Model(cfg) if token else Model(cfg, text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L)
    _close(yp, ref0["yp"], msg="yp")
    _close(model.loss, ref["loss"].reshape(1), msg="loss")
    model.backward(L)
    grads = model.get_oracle_grads()
    for k in ["text_kernel", "text_bias", "image_kernel", "att_W", "att_b", "qatt_W", "out_W"] + \
            (["conv_filter", "conv_bias", "img_W", "img_b", "word_emb", "char_emb"] if token else []):
        if p64.get(k) is None or p64[k].grad is None:
            continue
        exp = p64[k].grad.numpy()
        _close(np.asarray(grads[k]).reshape(exp.shape), exp, rtol=2e-4, atol=2e-6, msg="grad " + k)


@pytest.mark.gpu
def test_side_stream_runs_beside_the_main_stream():
    """The photo cell's side stream is chosen by measurement (ops.pick_side_stream): two spin waves, one per
    stream, must take about one wait, and a stream paired with itself about two."""
    import torch
    from fvta_memexqa_amd import ops
    dev = ops.require_gpu()
    main = torch.cuda.current_stream(dev)
    side, ratio = ops.pick_side_stream(dev)
    assert side.cuda_stream != main.cuda_stream
    assert ratio < 1.4, ratio
    assert ops.concurrency_ratio(main, main) > 1.6          # the same queue serialises: the probe can tell


@pytest.mark.gpu
def test_weights_npz_round_trip(tmp_path):
    """main.py:578-588 layout: TF variable names as keys, reference shapes; load restores them bit for bit."""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec
    spec = SynthSpec(N=2, A=1, P=3, S=2, L=4, d=20, dense=False, simiMatrix=2, add_tanh=True, use_question_att=True,
                     text_in=12, img_in=8)
    cfg = dict(spec.cfg(), use_time_warp=True, warp_type=5, window_t=2.3)
    a = Model(cfg, scope="model_fvta", text_in=12, img_in=8)
    a.init_parameters(seed=3)
    f = a.save_weights(str(tmp_path / "w"))
    z = np.load(f)
    assert "model_fvta/attention/all/att_logits/W:0" in z.files
    assert z["model_fvta/reader/text/utext/fw/basic_lstm_cell/kernel:0"].shape == (12 + 20, 80)
    assert z["model_fvta/attention/all/att_logits/W:0"].shape == (80, 1)          # simiMatrix 2: [h*q, (h-q)^2], w = 40
    assert abs(float(z["model_fvta/time_warp/time_warp_C/time_warp_window_t:0"]) - 2.3) < 1e-6
    lines = open(tmp_path / "w" / "all.txt").read().splitlines()
    assert "model_fvta/output/choicelogits/b:0 (1,)" in lines and "model_fvta/attention/all/att_logits/W:0 (80, 1)" in lines
    b = Model(dict(cfg, window_t=3.0), scope="other", text_in=12, img_in=8)
    b.init_parameters(seed=9)
    b.load_weights(str(tmp_path / "w"))
    wa, wb = a.get_weights(), b.get_weights()
    assert set(wa) == set(wb) and all(np.array_equal(wa[k], wb[k]) for k in wa)
    assert abs(b.window_t - 2.3) < 1e-6
    with pytest.raises(Exception, match="Model not exists"):
        b.load_weights(str(tmp_path / "nope"))
    # a file holding variables the model has no place for is refused (it would load a different network):
    # separate fw / bw cells into a share_fw_bw=True model, and the global step travels with the weights
    c = Model(dict(cfg, share_fw_bw=False), scope="model_fvta", text_in=12, img_in=8)
    c.global_step = 7
    c.save_weights(str(tmp_path / "w2"))
    with pytest.raises(KeyError, match="share_fw_bw=False"):
        b.load_weights(str(tmp_path / "w2"))
    d2 = Model(dict(cfg, share_fw_bw=False), scope="x", text_in=12, img_in=8)
    d2.load_weights(str(tmp_path / "w2"))
    assert d2.global_step == 7


@pytest.mark.parametrize("optimizer", ["adadelta", "adam"])
def test_trainer_checkpoint_resume_continues_bitwise(tmp_path, optimizer):
    """Trainer.save / restore (the reference's Saver writes the optimiser slots too, main.py:296, 430-440): three steps,
    checkpoint, two more == five uninterrupted steps, bit for bit, hidden size 20 (padded layout <-> reference shapes)"""
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params
    from fvta_memexqa_amd.trainer import Trainer
    spec = SynthSpec(N=4, A=1, P=3, S=2, L=5, d=20, dense=False, text_in=12, img_in=8)
    params, inputs = make_params(spec), make_inputs(spec)
    cfg = dict(spec.cfg(), batch_size=spec.N, init_lr=0.5 if optimizer == "adadelta" else 1e-3, optimizer=optimizer)

    def fresh():
        m = Model(cfg, text_in=spec.text_in, img_in=spec.img_in)
        m.set_oracle_params(params)
        return m, Trainer(m, cfg)
    batch = (None, dict(inputs, num_examples=4))
    m1, t1 = fresh()
    ref_losses = [t1.step(None, batch)[0] for _ in range(5)]
    m2, t2 = fresh()
    losses = [t2.step(None, batch)[0] for _ in range(3)]
    t2.save(str(tmp_path))
    m3, t3 = fresh()
    assert t3.restore(str(tmp_path)) and m3.global_step == 3
    losses += [t3.step(None, batch)[0] for _ in range(2)]
    assert losses == ref_losses
    assert torch.equal(m3.params.flat, m1.params.flat)
    with pytest.raises(ValueError):
        other = Trainer(m3, dict(cfg, optimizer="adam" if optimizer == "adadelta" else "adadelta"))
        other.restore(str(tmp_path))


def test_restore_adam_step_count_from_beta1_power(tmp_path):
    """tf.train.AdamOptimizer keeps beta1_power = beta1 ** (applies + 1) (initialised to beta1, multiplied once per
    apply): a checkpoint written after k applies resumes with opt.t == k -- the next step corrects with beta ** (k + 1),
    as TensorFlow's would -- whatever global_step says; a denormal power falls back to global_step."""
    from fvta_memexqa_amd import tf_checkpoint as tc
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params
    from fvta_memexqa_amd.trainer import Trainer
    spec = SynthSpec(N=4, A=1, P=3, S=2, L=5, d=20, dense=False, text_in=12, img_in=8)
    cfg = dict(spec.cfg(), batch_size=spec.N, init_lr=0.001, optimizer="adam")
    m1 = Model(cfg, scope="model_memoryqa", text_in=spec.text_in, img_in=spec.img_in)
    m1.set_oracle_params(make_params(spec))
    t1 = Trainer(m1, cfg)
    t1.step(None, (None, dict(make_inputs(spec), num_examples=4)))      # (creates the slot buffers)
    b1 = t1.opt.b1
    for k, b1p, want in ((5, np.float32(b1 ** 6), 5), (0, np.float32(b1), 0), (900, np.float32(b1 ** 901), 7)):
        tensors = {"%s/%s" % (m1.scope, n): v for n, v in m1.get_weights().items()}
        for slot, flat in zip(t1.opt.SLOTS, t1.opt.state):
            tensors.update({"%s/%s/%s" % (m1.scope, n, slot): v for n, v in m1.get_weights(flat=flat).items()})
        tensors["%s/global_step" % m1.scope] = np.int64(7)
        tensors["beta1_power"] = b1p
        tensors["beta2_power"] = np.float32(0.999 ** (k + 1))
        d = tmp_path / ("save%d" % k)
        tc.write_checkpoint(str(d / "model-7"), tensors)
        m2 = Model(cfg, scope="model_memoryqa", text_in=spec.text_in, img_in=spec.img_in)
        t2 = Trainer(m2, cfg)
        assert t2.restore_tf_checkpoint(str(d)) and t2.opt.t == want, (k, t2.opt.t)


def test_restore_from_tf_checkpoint_files(tmp_path):
    """main.py:640-665: a V2 checkpoint under the reference's variable names (model + Adadelta slots + global_step, as its
    Saver writes them) restores into Model / Trainer and continues like the run that wrote it.  The files come from
    tf_checkpoint.write_checkpoint (no TensorFlow here: the reader is pinned against our own writer only)."""
    from fvta_memexqa_amd import tf_checkpoint as tc
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params
    from fvta_memexqa_amd.trainer import Trainer
    spec = SynthSpec(N=4, A=1, P=3, S=2, L=5, d=20, dense=False, text_in=12, img_in=8)
    params, inputs = make_params(spec), make_inputs(spec)
    cfg = dict(spec.cfg(), batch_size=spec.N, init_lr=0.5)
    batch = (None, dict(inputs, num_examples=4))
    m1 = Model(cfg, scope="model_memoryqa", text_in=spec.text_in, img_in=spec.img_in)
    m1.set_oracle_params(params)
    t1 = Trainer(m1, cfg)
    for _ in range(3):
        t1.step(None, batch)
    tensors = {"%s/%s" % (m1.scope, k): v for k, v in m1.get_weights().items()}
    for slot, flat in zip(t1.opt.SLOTS, t1.opt.state):
        tensors.update({"%s/%s/%s" % (m1.scope, k, slot): v for k, v in m1.get_weights(flat=flat).items()})
    tensors["%s/global_step" % m1.scope] = np.int64(m1.global_step)
    tc.write_checkpoint(str(tmp_path / "save" / "model-3"), tensors)
    m2 = Model(cfg, scope="model_memoryqa", text_in=spec.text_in, img_in=spec.img_in)
    t2 = Trainer(m2, cfg)
    assert t2.restore_tf_checkpoint(str(tmp_path / "save")) and m2.global_step == 3
    assert torch.equal(m2.params.flat, m1.params.flat)
    assert t1.step(None, batch)[0] == t2.step(None, batch)[0] and torch.equal(m2.params.flat, m1.params.flat)
    # a model with a variable the checkpoint lacks / without one it holds refuses, like restore-by-name does
    with pytest.raises(KeyError):
        Model(dict(cfg, use_time_warp=True), text_in=spec.text_in, img_in=spec.img_in).load_tf_checkpoint(str(tmp_path / "save"))
    with pytest.raises(KeyError):
        Model(dict(cfg, use_question_att=False), text_in=spec.text_in, img_in=spec.img_in).load_tf_checkpoint(str(tmp_path / "save"))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_batches_with_other_lengths_through_one_layout(precision):
    """A cached layout keeps its plan memory, arenas and dx buffers across batches (out_pads_persist, dx_overwrite): a
    sequence of ragged batches of the same shapes through ONE model must give, batch by batch, bitwise what a fresh model
    gives -- context tensor (zeros where a batch is shorter than its predecessor), loss, input and parameter gradients."""
    import dataclasses
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params
    base = SynthSpec(N=4, A=2, P=3, S=2, L=7, d=32, SA=1, dense=False, text_in=12, img_in=8)
    params = make_params(base)
    mk = lambda: Model(dict(base.cfg(), batch_size=base.N, precision=precision), text_in=base.text_in, img_in=base.img_in)
    kept = mk()
    kept.set_oracle_params(params)
    layouts = set()
    for seed in (1, 2, 3, 2, 5):
        inputs = make_inputs(dataclasses.replace(base, seed=seed))
        res = []
        for model in (kept, mk()):
            if model is not kept:
                model.set_oracle_params(params)
            L = model.load_inputs(inputs, training=True)
            model.zero_grad()
            model.forward(L)
            model.backward(L, need_dx=True)
            torch.cuda.synchronize()
            res.append((L.arena.clone(), model.loss.clone(), model.params.grad.clone(),
                        [G.dx.clone() for _, G in sorted(L.groups.items())]))
            if model is kept:
                layouts.add(id(L))
        (a0, l0, g0, d0), (a1, l1, g1, d1) = res
        assert torch.equal(a0, a1), "seed %d: the kept layout's arena differs from a fresh one (stale padded rows?)" % seed
        assert torch.equal(l0, l1) and torch.equal(g0, g1)
        for x0, x1 in zip(d0, d1):
            assert torch.equal(x0, x1), "seed %d: dx differs (stale rows under dx_overwrite?)" % seed
    assert len(layouts) == 1, "the batches were meant to share one cached layout"
