"""Host batch assembly against the REFERENCE's own code: tests/golden/feed_*.{json,npz} were produced by running the
reference's `Dataset.get_batches`, `update_config` and `Model.get_feed_dict` (converted py2 -> py3 in memory) on a
synthetic dataset (tests/golden/make_feed_golden.py).  The product code (fvta_memexqa_amd/utils.py, feed.py) and the
literal restatement (oracle/feed_literal.py) must reproduce every array byte for byte."""
import glob
import json
import os
import random
from copy import deepcopy

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "feed_*.json")) if not p.endswith("feed_eval.json"))
ARRAY_KEYS = ["at", "at_c", "at_mask", "ad", "ad_c", "ad_mask", "when", "when_c", "when_mask", "where", "where_c",
              "where_mask", "pts", "pts_c", "pts_mask", "pis", "pis_mask", "q", "q_c", "q_mask", "choices", "choices_c",
              "choices_mask", "image_emb_mat", "existing_emb_mat"]
MAXMETA = ("max_num_albums", "max_num_photos", "max_sent_album_title_size", "max_sent_photo_title_size", "max_sent_des_size",
           "max_when_size", "max_where_size", "max_answer_size", "max_question_size", "max_word_size")


class Config:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def load_case(path):
    js = json.load(open(path))
    z = np.load(path[:-5] + ".npz")
    emb = z["existing_emb_mat"]
    glove = sorted(js["existing_word2idx"], key=js["existing_word2idx"].get)
    shared = dict(albums=js["albums"], pid2feat={p: z["pid2feat"][i] for i, p in enumerate(js["pids"])},
                  word2idx=js["word2idx"], existing_word2idx=js["existing_word2idx"], char2idx=js["char2idx"],
                  existing_emb_mat=emb, word2vec={w: emb[i] for i, w in enumerate(glove)})
    case = js["case"]
    config = Config(batch_size=case["batch_size"], is_train=case["is_train"], showspecs=False, hidden_size=8,
                    maxmeta=MAXMETA, **case["thres"])
    return js, z, shared, case, config


def check_feed(feed, z, b, is_train):
    for k in ARRAY_KEYS + (["y"] if is_train else []):
        want, got = z["b%d_%s" % (b, k)], np.asarray(feed[k])
        assert got.dtype == want.dtype and got.shape == want.shape, (k, got.dtype, want.dtype, got.shape, want.shape)
        assert np.array_equal(got, want), "batch %d, %s differs" % (b, k)
    assert feed["is_train"] == bool(z["b%d_is_train" % b])
    if not is_train:
        assert "y" not in feed


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-5] for p in CASES])
@pytest.mark.parametrize("impl", ["product", "product_walk", "literal"])
def test_batches_and_feed_match_reference(path, impl):
    js, z, shared, case, config = load_case(path)
    from fvta_memexqa_amd import utils as U
    if impl.startswith("product"):         # "product": album tables gathered by album index; "product_walk": nested lists walked
        from fvta_memexqa_amd.feed import build_feed_dict
        DS = U.Dataset
        feed_of = lambda batch, memo: build_feed_dict(config, batch, case["is_train"], 4, memo, use_tables=impl == "product")
    else:
        from oracle import feed_literal as FL
        DS = FL.Dataset
        feed_of = lambda batch, memo: (FL.get_feed_dict(config, batch, case["is_train"], 4), None)
    ds = DS(deepcopy(js["data"]), "train" if case["is_train"] else "val", shared=shared)
    U.update_config(config, [U.Dataset(deepcopy(js["data"]), "x", shared=shared)])
    for k, v in js["config_max"].items():
        assert getattr(config, k) == v, k
    random.seed(case["seed"])
    np.random.seed(case["seed"])
    memo, nb = None, 0
    for b, (idxs, batch) in enumerate(ds.get_batches(case["batch_size"], case["steps"], shuffle=case["shuffle"])):
        assert tuple(idxs) == tuple(z["b%d_idxs" % b])
        assert batch.num_examples == int(z["b%d_num_examples" % b])
        feed, memo = feed_of(batch, memo)
        check_feed(feed, z, b, case["is_train"])
        nb += 1
    assert nb == js["num_batches"]
    assert js["data"]["cs"] == ds.data["cs"]        # inserting the answer must not touch the dataset (deepcopy, :1249)


def test_word_lookup_order_and_unknowns():
    """model_v2.py:1325-1342: as written > lower > capitalised > upper, trainable before frozen (+len shift), else 1."""
    from fvta_memexqa_amd.feed import VocabMemo
    sh = dict(word2idx={"lake": 2, "Tahoe": 3, "HIKE": 4}, existing_word2idx={"lake": 0, "nyc": 1, "Usa": 2}, char2idx={"a": 2})
    m = VocabMemo(sh, 3)
    ids = m.word_ids(["lake", "LAKE", "tahoe", "hike", "NYC", "usa", "zzz", "Lake"])
    assert ids.tolist() == [2, 2, 3, 4, 3 + 1, 3 + 2, 1, 2]
    assert m.char_rows([list("abca"), []]).tolist() == [[2, 1, 1], [0, 0, 0]]


def test_question_longer_than_cap_raises_like_the_reference():
    """the reference never clips the question (model_v2.py:1525-1531): a question longer than max_question_size
    indexes past the array -> IndexError there and here."""
    path = CASES[0]
    js, z, shared, case, config = load_case(path)
    from fvta_memexqa_amd import utils as U
    from fvta_memexqa_amd.feed import build_feed_dict
    U.update_config(config, [U.Dataset(deepcopy(js["data"]), "x", shared=shared)])
    config.max_question_size = 2
    ds = U.Dataset(deepcopy(js["data"]), "train", shared=shared)
    _, batch = next(ds.get_batches(case["batch_size"], 1, shuffle=False))
    with pytest.raises(IndexError):
        build_feed_dict(config, batch, False, 4)


def test_eval_batch_without_answers_fails_like_the_reference():
    """no y/cy/yidx -> three choices per question -> the reference's `assert len(ci) == self.num_choice` (:1497)."""
    js, z, shared, case, config = load_case(CASES[0])
    from fvta_memexqa_amd import utils as U
    from fvta_memexqa_amd.feed import build_feed_dict
    U.update_config(config, [U.Dataset(deepcopy(js["data"]), "x", shared=shared)])
    data = {k: v for k, v in deepcopy(js["data"]).items() if k != "yidx"}
    _, batch = next(U.Dataset(data, "test", shared=shared).get_batches(case["batch_size"], 1, shuffle=False))
    with pytest.raises(AssertionError):
        build_feed_dict(config, batch, False, 4)


def test_get_answers_and_score():
    from fvta_memexqa_amd import utils as U
    batch = (None, U.Dataset(dict(qid=[7, 8, 9], yidx=[0, 3, 1]), "val"))
    yp = np.array([[.7, .1, .1, .1], [.1, .2, .3, .4], [.4, .3, .2, .1]])
    pred, real = U.getAnswers(yp, batch)
    assert pred == {7: 0, 8: 3, 9: 0} and real == {7: 0, 8: 3, 9: 1}
    assert abs(U.getEvalScore(pred, real) - 2 / 3) < 1e-12
    assert U.grouper([1, 2, 3, 4, 5], 2) == [(1, 2), (3, 4), (5, None)]


def test_eval_helpers_match_reference():
    """getAnswers / getAnswers_yp / getEvalScore / grouper / sec2time vs the values the reference's own functions
    produced (tests/golden/feed_eval.json, made by make_feed_golden.run_eval_case)."""
    from fvta_memexqa_amd import utils as U
    z = json.load(open(os.path.join(HERE, "golden", "feed_eval.json")))
    yp = np.asarray(z["yp"], "float32")
    batch = (tuple(range(len(z["qid"]))), U.Dataset(dict(qid=z["qid"], yidx=z["yidx"]), "val"))
    pred, real = U.getAnswers(yp, batch)
    assert {str(k): v for k, v in pred.items()} == z["pred"] and {str(k): v for k, v in real.items()} == z["real"]
    p2, r2, id2yp = U.getAnswers_yp(yp, batch)
    assert p2 == pred and r2 == real and all(np.array_equal(id2yp[q], yp[i]) for i, q in enumerate(z["qid"]))
    assert U.getEvalScore(pred, real) == z["score"]
    assert [list(g) for g in U.grouper(list(range(7)), 3)] == z["grouper"]
    for t, want in z["sec2time"].items():
        assert U.sec2time(float(t)) == want
