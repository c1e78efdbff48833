"""Generates tests/golden/feed_*.json + feed_*.npz -- golden vectors of the reference's HOST batch assembly.

Unlike the arithmetic of the path (TensorFlow; cannot run here), the reference's batch assembly is plain
Python + NumPy: `Dataset` / `grouper` (utils.py:12-198) and `Model.get_feed_dict` (model_v2.py:1099-1565).  The
files are Python-2 syntax, so this script -- run in THIS container only, where /root/reference exists --

  1. reads the two reference files, converts them to Python 3 IN MEMORY with lib2to3 (nothing is written anywhere),
  2. pulls out exactly `grouper`, `Dataset`, `update_config` and `get_feed_dict` by AST (the rest of the modules
     needs TensorFlow), and executes those definitions,
  3. runs them on a small seeded synthetic MemexQA-shaped dataset (albums, photo titles, questions, choices),
  4. stores the INPUT (the dataset, as JSON + the feature matrices) and the OUTPUT (every array of every feed dict,
     the batch index tuples, the config maxima) as fixtures.

The fixtures are data; no reference source text is stored.  tests/test_feed_golden.py replays the inputs through
fvta_memexqa_amd.utils.Dataset / Model.get_feed_dict and oracle/feed_literal.py and demands byte-identical arrays.

    python tests/golden/make_feed_golden.py
"""
import ast
import json
import os
import random
import sys
import warnings
from copy import deepcopy  # noqa: F401  (used by the reference code)

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

FEED_KEYS = ["at", "at_c", "at_mask", "ad", "ad_c", "ad_mask", "when", "when_c", "when_mask", "where", "where_c",
             "where_mask", "pts", "pts_c", "pts_mask", "pis", "pis_mask", "q", "q_c", "q_mask", "choices", "choices_c",
             "choices_mask", "y", "is_train", "image_emb_mat", "existing_emb_mat"]


def load_reference():
    """-> namespace holding the reference's grouper / Dataset / update_config / get_feed_dict (py3-converted)."""
    warnings.filterwarnings("ignore")
    from lib2to3 import refactor
    rt = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    ns = {"np": np, "deepcopy": deepcopy}
    exec("import random, itertools, math\nfrom collections import defaultdict\nfrom itertools import zip_longest\n", ns)

    def pull(fname, names, inside_class=None):
        src = str(rt.refactor_string(open(os.path.join(REF, fname)).read(), fname))
        tree = ast.parse(src)
        body = tree.body
        if inside_class:
            body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == inside_class).body
        for node in body:
            if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
                exec(compile(ast.Module(body=[node], type_ignores=[]), fname, "exec"), ns)

    pull("utils.py", {"grouper", "Dataset", "update_config", "getAnswers", "getAnswers_yp", "getEvalScore", "sec2time"})
    pull("model_v2.py", {"get_feed_dict"}, inside_class="Model")
    return ns


class _Self:
    """what get_feed_dict reads from `self`: config, num_choice, and the placeholder objects used as dict keys
    (here: their names)."""

    def __init__(self, config, num_choice=4):
        self.config, self.num_choice = config, num_choice
        for k in FEED_KEYS:
            setattr(self, k, k)


class Config:
    def __init__(self, **kw):
        self.__dict__.update(kw)


# ------------------------------------------------------------------------------------------- the synthetic dataset
def make_dataset(seed, n_qa=10, n_albums=7, idim=6, wdim=5):
    rng = random.Random(seed)
    nr = np.random.RandomState(seed)
    base = ["beach", "party", "wedding", "trip", "paris", "2005", "july", "dog", "Birthday", "NYC", "the", "a", "in", "of",
            "my", "first", "day", "Lake", "tahoe", "hike", "friends", "family", "USA", "graduation", "what", "where",
            "when", "who", "did", "we", "go", "last", "eat", "?", ",", "'s", "café", "extraordinarily-long-token-here"]
    glove_words = ["beach", "party", "the", "a", "in", "of", "paris", "dog", "nyc", "usa", "what", "where", "did", "?"]
    # trainable vocabulary: idx+2 (0 = NULL, 1 = UNK), as main.py:245 builds it (words absent from the GloVe table)
    train_words = ["wedding", "trip", "2005", "july", "birthday", "my", "first", "day", "lake", "Tahoe", "HIKE", "friends"]
    word2idx = {w: i + 2 for i, w in enumerate(train_words)}
    existing_word2idx = {w: i for i, w in enumerate(glove_words)}
    chars = sorted(set("".join(base)) - set("zéq"))          # some characters are unknown -> 1
    char2idx = {c: i + 2 for i, c in enumerate(chars)}

    def sent(lo, hi):
        out = []
        for _ in range(rng.randint(lo, hi)):
            w = rng.choice(base)
            r = rng.random()
            out.append(w.upper() if r < 0.1 else w.capitalize() if r < 0.2 else w.lower() if r < 0.3 else w)
        return out

    def chars_of(s):
        return [list(w) for w in s]

    albums, pid2feat = {}, {}
    for a in range(n_albums):
        aid = "album%d" % a
        nph = rng.randint(1, 6)
        pids = ["p%d_%d" % (a, k) for k in range(nph)]
        if a >= 2 and rng.random() < 0.5:
            pids[0] = "p%d_%d" % (a - 1, 0)                     # a photo shared between albums
        titles = [sent(0, 7) for _ in pids]                     # empty photo titles occur
        alb = dict(title=sent(1, 6), description=sent(0, 9), where=sent(0, 3), when=sent(1, 3), photo_titles=titles,
                   photo_ids=pids)
        for k in ("title", "description", "where", "when"):
            alb[k + "_c"] = chars_of(alb[k])
        alb["photo_titles_c"] = [chars_of(t) for t in titles]
        albums[aid] = alb
        for p in pids:
            pid2feat.setdefault(p, nr.randn(idim).astype("float32"))
    data = dict(q=[], cq=[], y=[], cy=[], yidx=[], cs=[], ccs=[], aid=[], qid=[], idxs=[])
    for i in range(n_qa):
        q = sent(3, 9)
        y = sent(1, 4)
        cs = [sent(1, 7) for _ in range(3)]
        data["q"].append(q), data["cq"].append(chars_of(q))
        data["y"].append(y), data["cy"].append(chars_of(y))
        data["yidx"].append(rng.randint(0, 3))
        data["cs"].append(cs), data["ccs"].append([chars_of(c) for c in cs])
        data["aid"].append(rng.sample(sorted(albums), rng.randint(1, 3)))
        data["qid"].append(1000 + i), data["idxs"].append(i)
    emb = nr.randn(len(glove_words), wdim).astype("float32")
    shared = dict(albums=albums, pid2feat=pid2feat, word2idx=word2idx, existing_word2idx=existing_word2idx,
                  char2idx=char2idx, existing_emb_mat=emb, word2vec={w: emb[i] for i, w in enumerate(glove_words)})
    return data, shared


def dataset_to_json(data, shared):
    js = dict(data=data, albums=shared["albums"], word2idx=shared["word2idx"], existing_word2idx=shared["existing_word2idx"],
              char2idx=shared["char2idx"], pids=sorted(shared["pid2feat"]))
    return js


CASES = {
    # name: (seed, batch_size, is_train, shuffle, thresholds)
    "feed_train_caps": dict(seed=11, batch_size=4, is_train=True, shuffle=False, steps=3,
                            thres=dict(sent_album_title_size_thres=4, sent_photo_title_size_thres=3, sent_des_size_thres=5,
                                       sent_when_size_thres=2, sent_where_size_thres=2, answer_size_thres=3,
                                       question_size_thres=25, num_photos_thres=4, num_albums_thres=2, word_size_thres=6)),
    "feed_test_nocaps": dict(seed=12, batch_size=3, is_train=False, shuffle=False, steps=4,
                             thres=dict(sent_album_title_size_thres=10, sent_photo_title_size_thres=8, sent_des_size_thres=10,
                                        sent_when_size_thres=4, sent_where_size_thres=4, answer_size_thres=5,
                                        question_size_thres=25, num_photos_thres=10, num_albums_thres=8, word_size_thres=16)),
    "feed_train_shuffle": dict(seed=13, batch_size=4, is_train=True, shuffle=True, steps=5,
                               thres=dict(sent_album_title_size_thres=10, sent_photo_title_size_thres=8, sent_des_size_thres=10,
                                          sent_when_size_thres=4, sent_where_size_thres=4, answer_size_thres=5,
                                          question_size_thres=25, num_photos_thres=10, num_albums_thres=8, word_size_thres=16)),
}
MAXMETA = ("max_num_albums", "max_num_photos", "max_sent_album_title_size", "max_sent_photo_title_size", "max_sent_des_size",
           "max_when_size", "max_where_size", "max_answer_size", "max_question_size", "max_word_size")


def run_case(ref, name, case):
    data, shared = make_dataset(case["seed"])
    js = dataset_to_json(data, shared)
    js["case"] = {k: v for k, v in case.items()}
    arrays = {"pid2feat": np.stack([shared["pid2feat"][p] for p in js["pids"]]), "existing_emb_mat": shared["existing_emb_mat"]}
    ds = ref["Dataset"](deepcopy(data), "train" if case["is_train"] else "val", shared=shared)
    config = Config(batch_size=case["batch_size"], is_train=case["is_train"], showspecs=False, hidden_size=8,
                    word_vocab_size=len(shared["word2idx"]) + 2, char_vocab_size=len(shared["char2idx"]) + 2,
                    maxmeta=MAXMETA, **case["thres"])
    ref["update_config"](config, [ds], showMeta=False)          # utils.py:294-: maxima, capped by the thresholds when training
    js["config_max"] = {k: getattr(config, k) for k in MAXMETA + ("char_vocab_size", "word_emb_size", "word_vocab_size")}
    model = _Self(config)
    random.seed(case["seed"])                                      # get_batches(shuffle=True) draws from `random`
    np.random.seed(case["seed"])                                   # get_feed_dict(is_train=True) draws correctIndex from np.random
    nb = 0
    for batch_idxs, batch in ref["Dataset"].get_batches(ds, case["batch_size"], case["steps"], shuffle=case["shuffle"]):
        feed = ref["get_feed_dict"](model, batch, is_train=case["is_train"])
        arrays["b%d_idxs" % nb] = np.asarray(batch_idxs, np.int64)
        arrays["b%d_num_examples" % nb] = np.asarray(batch.num_examples)
        for k, v in feed.items():
            arrays["b%d_%s" % (nb, k)] = np.asarray(v)
        nb += 1
    js["num_batches"] = nb
    json.dump(js, open(os.path.join(HERE, name + ".json"), "w"), sort_keys=True, ensure_ascii=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print(name, nb, "batches;", {k: arrays["b0_" + k].shape for k in ("at", "pts_c", "q", "choices")}, js["config_max"])


def run_eval_case(ref):
    """getAnswers / getAnswers_yp / getEvalScore / grouper / sec2time (utils.py:12-29, 247-292) on a fixed yp"""
    rng = np.random.RandomState(5)
    yp = rng.rand(7, 4).astype("float32")
    yp[2] = [0.25, 0.25, 0.25, 0.25]                                # a tie: argmax takes the first
    qid, yidx = [101, 102, 103, 104, 105, 106, 107], [0, 3, 0, 2, 1, 1, 3]
    batch = (tuple(range(7)), ref["Dataset"](dict(qid=qid, yidx=yidx), "val"))
    pred, real = ref["getAnswers"](yp, batch)
    pred2, real2, id2yp = ref["getAnswers_yp"](yp, batch)
    out = dict(yp=yp.tolist(), qid=qid, yidx=yidx, pred={str(k): int(v) for k, v in pred.items()},
               real={str(k): int(v) for k, v in real.items()}, score=ref["getEvalScore"](pred, real),
               grouper=[list(g) for g in ref["grouper"](list(range(7)), 3)],
               sec2time={str(t): ref["sec2time"](t) for t in (0.5, 9.999, 75.25, 3671.0)})
    assert pred == pred2 and real == real2 and len(id2yp) == 7
    json.dump(out, open(os.path.join(HERE, "feed_eval.json"), "w"), sort_keys=True)
    print("feed_eval", out["pred"], out["score"], out["sec2time"])


if __name__ == "__main__":
    ref = load_reference()
    for n, c in CASES.items():
        run_case(ref, n, c)
    run_eval_case(ref)
