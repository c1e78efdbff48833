"""The reference's own entry, end to end on the GPU: Dataset.get_batches -> Trainer.step / Tester.step
(trainer.py:30-40, tester.py:16-25) on the golden synthetic dataset, checked against the oracle run on the SAME feed
arrays (embed_inputs -> fvta_forward).  Streams have different lengths here (album title / description / when / where
/ photo titles / photos), so the context tensor's per-stream padding to JMAX (model_v2.py:863-914) is exercised with
real shapes, and the last batch is short (num_examples < batch_size)."""
import json
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


class Config:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _setup(name, is_train, **over):
    from test_feed_golden import MAXMETA, load_case
    from fvta_memexqa_amd import utils as U
    js, z, shared, case, config = load_case(os.path.join(HERE, "golden", name + ".json"))
    config.is_train = is_train
    U.update_config(config, [U.Dataset(deepcopy(js["data"]), "x", shared=shared)])
    config.__dict__.update(hidden_size=32, simiMatrix=2, add_tanh=True, use_question_att=True, num_choice=4,
                           word_vocab_size=len(shared["word2idx"]) + 2, char_vocab_size=len(shared["char2idx"]) + 2,
                           word_emb_size=shared["existing_emb_mat"].shape[1], use_char=True, char_emb_size=8, char_out_size=12,
                           image_feat_dim=z["pid2feat"].shape[1], use_image_trans=True, image_trans_dim=8, init_lr=0.5,
                           precision="f32", **over)
    ds = U.Dataset(deepcopy(js["data"]), "train" if is_train else "val", shared=shared)
    return config, ds, case


def _oracle(model, feed, cfg):
    from oracle import fvta_fused as F
    tok = model.inputs_from_feed(feed)                 # (uploads the fed existing_emb_mat on first sight)
    p = {k: v.double() for k, v in model.get_oracle_params().items()}
    tok["image_emb_mat"] = torch.from_numpy(np.asarray(tok["image_emb_mat"])).double()
    ocfg = dict(hidden_size=cfg.hidden_size, simiMatrix=cfg.simiMatrix, add_tanh=cfg.add_tanh,
                use_question_att=cfg.use_question_att, num_choice=4)
    return F.fvta_forward(p, F.embed_inputs(p, tok, ocfg), ocfg)


def test_tester_step_on_dataset_batches():
    from fvta_memexqa_amd.model_v2 import get_model
    from fvta_memexqa_amd.tester import Tester
    config, ds, case = _setup("feed_test_nocaps", False)
    model = get_model(config)
    tester = Tester(model, config)
    seen = 0
    for batch in ds.get_batches(case["batch_size"], case["steps"], shuffle=False):
        yp = tester.step(None, batch)
        assert yp.shape == (batch[1].num_examples, 4)
        ref = _oracle(model, model.get_feed_dict(batch[1], is_train=False), config)["yp"].numpy()[:batch[1].num_examples]
        np.testing.assert_allclose(yp, ref, rtol=1e-4, atol=1e-6)
        assert (yp.argmax(1) == ref.argmax(1)).all()
        seen += len(yp)
    assert seen == ds.num_examples                       # 10 QA pairs in batches of 3: the last batch holds one


def test_trainer_step_on_dataset_batches():
    from fvta_memexqa_amd.model_v2 import get_model
    from fvta_memexqa_amd.trainer import Trainer
    config, ds, case = _setup("feed_train_shuffle", True)
    model = get_model(config)
    trainer = Trainer(model, config)
    import random
    random.seed(5)
    for b, batch in enumerate(ds.get_batches(case["batch_size"], 3, shuffle=True)):
        np.random.seed(100 + b)                           # the slot of the correct answer is drawn inside get_feed_dict
        feed = model.get_feed_dict(batch[1], is_train=True)
        ref = _oracle(model, feed, config)                # with the parameters BEFORE the update
        np.random.seed(100 + b)
        loss, summary, train_op = trainer.step(None, batch)
        assert summary is None
        np.testing.assert_allclose(loss, float(ref["loss"]), rtol=1e-4)
    assert model.global_step == 3


def test_no_photo_drops_the_photo_stream():
    from fvta_memexqa_amd.model_v2 import get_model
    config, ds, case = _setup("feed_test_nocaps", False, no_photo=True)
    model = get_model(config)
    batch = next(ds.get_batches(case["batch_size"], 1, shuffle=False))
    L = model.load_inputs(model.get_feed_dict(batch[1]), training=False)
    assert L.K == 5 and "image" not in L.groups          # model_v2.py:864-865
    model.forward(L)
    assert torch.isfinite(model.yp).all()


def test_published_fvta_flag_set_on_dataset_batches():
    """README.MD:141-147 / 219-226 flag set at its own sizes where they matter for the code paths: hidden_size 50 (padded to
    64 by the kernels), char_emb_size 100 (the general char-CNN kernels), time warp type 5, question attention,
    image_trans 100 -- Tester.step on Dataset batches vs the oracle on the same feed."""
    from fvta_memexqa_amd.model_v2 import get_model
    from fvta_memexqa_amd.tester import Tester
    from oracle import fvta_fused as F
    config, ds, case = _setup("feed_test_nocaps", False, use_time_warp=True, warp_type=5)
    config.__dict__.update(hidden_size=50, char_emb_size=100, char_out_size=100, image_trans_dim=100)
    model = get_model(config)
    tester = Tester(model, config)
    for batch in ds.get_batches(case["batch_size"], 2, shuffle=False):
        yp = tester.step(None, batch)
        feed = model.get_feed_dict(batch[1], is_train=False)
        tok = model.inputs_from_feed(feed)
        p = {k: (v.double() if torch.is_tensor(v) else v) for k, v in model.get_oracle_params().items()}
        p["window_t"] = model.window_t
        tok["image_emb_mat"] = torch.from_numpy(np.asarray(tok["image_emb_mat"])).double()
        ocfg = dict(hidden_size=50, simiMatrix=2, add_tanh=True, use_question_att=True, num_choice=4, use_time_warp=True,
                    warp_type=5)
        ref = F.fvta_forward(p, F.embed_inputs(p, tok, ocfg), ocfg)["yp"].numpy()[:batch[1].num_examples]
        np.testing.assert_allclose(yp, ref, rtol=1e-4, atol=1e-6)
        assert (yp.argmax(1) == ref.argmax(1)).all()


def test_step_vis_returns_the_reference_tuple():
    """tester.py:30-43: 28 values in the reference's order, trimmed like there."""
    from fvta_memexqa_amd.model_v2 import get_model
    from fvta_memexqa_amd.tester import Tester
    config, ds, case = _setup("feed_test_nocaps", False, use_time_warp=True, warp_type=5)
    config.hidden_size = 50                                     # padded to 64 inside: the vis tensors come back as [.., 100]
    model = get_model(config)
    batches = list(ds.get_batches(case["batch_size"], 4, shuffle=False))
    batch = batches[-1]                                         # the short last batch: one example in a batch of 3
    out = Tester(model, config).step_vis(None, batch)
    assert len(out) == 28
    (yp, C, C_win, att, qatt, at_mask, ad_mask, when_mask, where_mask, pts_mask, pis_mask, q_mask, hat_len, had_len, hwhen_len,
     hwhere_len, hpts_len, hpis_len, JXP, warp_h, h, at, ad, when, where, pts, pis, q) = out
    n, N = batch[1].num_examples, config.batch_size
    assert n == 1 and yp.shape == (n, 4)
    feed = model.get_feed_dict(batch[1])
    K, M, T = 6, feed["at"].shape[1], h.shape[2] * h.shape[3]
    assert h.shape[:3] == (N, K, M) and h.shape[-1] == 100 and warp_h.shape == h.shape
    assert C.shape == (n, T, T)
    assert C_win == -1            # a scalar is not an ndarray: tester.py:28's `trim` turns it into -1, there and here
    assert att.shape == (n, K, T, feed["q"].shape[1]) and qatt.shape[0] == n
    assert at_mask.shape[0] == n and pts_mask.shape[0] == n and ad_mask.shape[0] == N      # only some are trimmed (:42)
    assert np.array_equal(hat_len, feed["at_mask"].sum(2)) and np.array_equal(hpts_len, feed["pts_mask"].sum(3))
    assert JXP == feed["pts"].shape[3] and np.array_equal(pts, feed["pts"]) and np.array_equal(q, feed["q"])
    # the context tensor rows past each stream's own length are zero (model_v2.py:871-888 pads with zeros)
    JXA = feed["at"].shape[2]
    assert np.abs(h[:, 0, :, JXA:, :]).max() == 0


ALL_V1 = dict(use_ml_att=True, use_mm_att=True, use_direct_links=True, use_choices_att=True, use_question_att=True)


def _oracle_v1(model, feed, cfg, flags):
    from oracle import fvta_fused as F
    tok = model.inputs_from_feed(feed)
    p = {k: v.double() for k, v in model.get_oracle_params().items()}
    tok["image_emb_mat"] = torch.from_numpy(np.asarray(tok["image_emb_mat"])).double()
    ocfg = dict(hidden_size=cfg.hidden_size, simiMatrix=cfg.simiMatrix, add_tanh=cfg.add_tanh, num_choice=4, **flags)
    return F.model_v1_forward(p, F.embed_inputs(p, tok, ocfg), dict(ocfg, add_tanh=False))


@pytest.mark.parametrize("flags", [{}, ALL_V1], ids=["lstm_baseline", "all_attentions"])
def test_model_py_graph_on_dataset_batches(flags):
    """main.py:161-165 without --use_3d imports model.py: the soft-attention baselines on the reference's own feed
    (six streams of different lengths at / ad / when / where / pts / pis, short last batch), Tester.step and
    Trainer.step against the oracle on the same feed arrays.  (add_tanh stays on: in model.py it reaches only the
    image_trans_linear layer, :611.)"""
    from fvta_memexqa_amd.model import get_model
    from fvta_memexqa_amd.tester import Tester
    from fvta_memexqa_amd.trainer import Trainer
    config, ds, case = _setup("feed_test_nocaps", False)
    config.__dict__.update(dict(dict(use_question_att=False), **flags))
    model = get_model(config)
    tester = Tester(model, config)
    for batch in ds.get_batches(case["batch_size"], case["steps"], shuffle=False):
        yp = tester.step(None, batch)
        ref = _oracle_v1(model, model.get_feed_dict(batch[1], is_train=False), config, flags)["yp"].numpy()[:batch[1].num_examples]
        np.testing.assert_allclose(yp, ref, rtol=1e-4, atol=1e-6)
        assert (yp.argmax(1) == ref.argmax(1)).all()
    with pytest.raises(AttributeError):                  # tester.py:37 asks the model for C / warp_h / hall
        tester.step_vis(None, next(iter(ds.get_batches(case["batch_size"], 1, shuffle=False))))
    config, ds, case = _setup("feed_train_shuffle", True)
    config.__dict__.update(dict(dict(use_question_att=False), **flags))
    model = get_model(config)
    trainer = Trainer(model, config)
    import random
    random.seed(5)
    for b, batch in enumerate(ds.get_batches(case["batch_size"], 3, shuffle=True)):
        np.random.seed(100 + b)
        ref = _oracle_v1(model, model.get_feed_dict(batch[1], is_train=True), config, flags)
        np.random.seed(100 + b)
        loss, _, _ = trainer.step(None, batch)
        np.testing.assert_allclose(loss, float(ref["loss"]), rtol=1e-4)
