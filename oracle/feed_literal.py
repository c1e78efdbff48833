"""TEST INFRASTRUCTURE -- CPU restatement of the reference's host batch assembly (not a product path).

Restates, as plain per-element Python loops, what `Model.get_feed_dict` (model_v2.py:1099-1565) and
`Dataset.get_batches` (utils.py:89-198) compute.  Unlike the arithmetic oracle this one IS pinned: the reference's own
functions were executed in this container (py2 -> py3 converted in memory) by tests/golden/make_feed_golden.py and
their outputs are the fixtures tests/golden/feed_*.npz, which tests/test_feed_golden.py replays through this file and
through the product code (fvta_memexqa_amd/utils.py, feed.py).  Only tests/ may import this module.
"""
import itertools
import math
import random
from collections import defaultdict
from copy import deepcopy

import numpy as np


def word_id(shared, word):
    """model_v2.py:1325-1336: trainable vocabulary first, then the frozen (GloVe) one shifted by len(word2idx); each
    tried as written, lower, capitalised, upper; 1 = -UNK-."""
    forms = (word, word.lower(), word.capitalize(), word.upper())
    d = shared["word2idx"]
    for f in forms:
        if f in d:
            return d[f]
    d2 = shared["existing_word2idx"]
    for f in forms:
        if f in d2:
            return d2[f] + len(d)
    return 1


def char_id(shared, ch):
    """model_v2.py:1338-1342."""
    return shared["char2idx"].get(ch, 1)


def _batch_max(nested, depth, floor_one=True):
    """max length found `depth` list levels down (model_v2.py:1126-1163); 0 -> 1."""
    items = nested
    for _ in range(depth):
        items = [x for sub in items for x in sub]
    m = max([len(x) for x in items] + [0])
    return 1 if (m == 0 and floor_one) else m


def get_feed_dict(config, batch, is_train=False, num_choice=4):
    """-> dict keyed by the reference's placeholder names."""
    d, sh = batch.data, batch.shared
    N = 2 if getattr(config, "showspecs", False) else config.batch_size
    W = config.max_word_size
    M = min(config.max_num_albums, _batch_max([d["album_title"]], 1))           # :1160-1163 (len of each sample)
    JXA = min(config.max_sent_album_title_size, _batch_max(d["album_title"], 1))
    JXP = min(config.max_sent_photo_title_size, _batch_max(d["photo_titles"], 2))
    JD = min(config.max_sent_des_size, _batch_max(d["album_description"], 1))
    JG = min(config.max_where_size, _batch_max(d["where"], 1))
    JT = min(config.max_when_size, _batch_max(d["when"], 1))
    JI = min(config.max_num_photos, _batch_max(d["photo_ids"], 1))
    JQ = min(config.max_question_size, _batch_max([d["q"]], 1))
    JA = config.max_answer_size
    f = {}

    def alloc(name, shape):
        f[name] = np.zeros(shape, "int32")
        f[name + "_c"] = np.zeros(tuple(shape) + (W,), "int32")
        f[name + "_mask"] = np.zeros(shape, "bool")

    alloc("at", (N, M, JXA)), alloc("ad", (N, M, JD)), alloc("when", (N, M, JT)), alloc("where", (N, M, JG))
    alloc("pts", (N, M, JI, JXP)), alloc("q", (N, JQ)), alloc("choices", (N, num_choice, JA))
    f["pis"], f["pis_mask"] = np.zeros((N, M, JI), "int32"), np.zeros((N, M, JI), "bool")
    f["is_train"] = is_train
    f["image_emb_mat"], f["existing_emb_mat"] = d["pidx2feat"], sh["existing_emb_mat"]

    C, Cc = deepcopy(d["cs"]), deepcopy(d["ccs"])
    if is_train:                                                                  # :1270-1288
        f["y"] = np.zeros((N, num_choice), "bool")
        correct = np.random.choice(num_choice, N)
        for i in range(len(d["y"])):
            f["y"][i, correct[i]] = True
            assert len(C[i]) == num_choice - 1
            C[i].insert(correct[i], d["y"][i])
            Cc[i].insert(correct[i], d["cy"][i])
    elif "y" in d and "cy" in d and "yidx" in d:                                  # :1296-1306
        for i in range(len(d["y"])):
            assert len(C[i]) == num_choice - 1
            C[i].insert(d["yidx"][i], d["y"][i])
            Cc[i].insert(d["yidx"][i], d["cy"][i])

    for i, sample in enumerate(d["photo_idxs"]):                                  # :1312-1324
        for j, album in enumerate(sample[:config.max_num_albums]):
            for k, p in enumerate(album[:config.max_num_photos]):
                assert isinstance(p, int)
                f["pis"][i, j, k] = p
                f["pis_mask"][i, j, k] = True

    def album_text(name, words, chars, cap):                                      # :1344-1450
        for i, sample in enumerate(words):
            for j, sent in enumerate(sample[:config.max_num_albums]):
                for k, w in enumerate(sent[:cap]):
                    f[name][i, j, k] = word_id(sh, w)
                    f[name + "_mask"][i, j, k] = True
        for i, sample in enumerate(chars):
            for j, sent in enumerate(sample[:config.max_num_albums]):
                for k, w in enumerate(sent[:cap]):
                    for l, ch in enumerate(w[:W]):
                        f[name + "_c"][i, j, k, l] = char_id(sh, ch)

    album_text("at", d["album_title"], d["album_title_c"], config.max_sent_album_title_size)
    album_text("ad", d["album_description"], d["album_description_c"], config.max_sent_des_size)
    album_text("when", d["when"], d["when_c"], config.max_when_size)
    album_text("where", d["where"], d["where_c"], config.max_where_size)

    for i, sample in enumerate(d["photo_titles"]):                                # :1453-1490
        for j, album in enumerate(sample[:config.max_num_albums]):
            for k, title in enumerate(album[:config.max_num_photos]):
                for l, w in enumerate(title[:config.max_sent_photo_title_size]):
                    f["pts"][i, j, k, l] = word_id(sh, w)
                    f["pts_mask"][i, j, k, l] = True
    for i, sample in enumerate(d["photo_titles_c"]):
        for j, album in enumerate(sample[:config.max_num_albums]):
            for k, title in enumerate(album[:config.max_num_photos]):
                for l, w in enumerate(title[:config.max_sent_photo_title_size]):
                    for o, ch in enumerate(w[:W]):
                        f["pts_c"][i, j, k, l, o] = char_id(sh, ch)

    for i, ci in enumerate(C):                                                    # :1495-1520
        assert len(ci) == num_choice
        for j, ans in enumerate(ci):
            for k, w in enumerate(ans[:config.max_answer_size]):
                f["choices"][i, j, k] = word_id(sh, w)
                f["choices_mask"][i, j, k] = True
    for i, ci in enumerate(Cc):
        assert len(ci) == num_choice, len(ci)
        for j, ans in enumerate(ci):
            for k, w in enumerate(ans[:config.max_answer_size]):
                for l, ch in enumerate(w[:W]):
                    f["choices_c"][i, j, k, l] = char_id(sh, ch)

    for i, qi in enumerate(d["q"]):                                               # :1525-1540: the question is not clipped
        for j, w in enumerate(qi):
            f["q"][i, j] = word_id(sh, w)
            f["q_mask"][i, j] = True
    for i, qi in enumerate(d["cq"]):
        for j, w in enumerate(qi):
            for k, ch in enumerate(w[:W]):
                f["q_c"][i, j, k] = char_id(sh, ch)
    return f


class Dataset:
    """utils.py:31-198, literal."""

    def __init__(self, data, datatype, shared=None, valid_idxs=None):
        self.data, self.datatype, self.shared = data, datatype, shared
        self.valid_idxs = range(len(next(iter(data.values())))) if valid_idxs is None else valid_idxs
        self.num_examples = len(self.valid_idxs)

    def get_batches(self, batch_size, num_steps, shuffle=True, cap=False):
        per_epoch = int(math.ceil(self.num_examples / float(batch_size)))
        if cap and num_steps > per_epoch:
            num_steps = per_epoch
        num_epochs = int(math.ceil(num_steps / float(per_epoch)))
        idxs = random.sample(list(self.valid_idxs), len(self.valid_idxs)) if shuffle else list(self.valid_idxs)

        def grouped():
            return list(itertools.zip_longest(*([iter(idxs)] * batch_size), fillvalue=None))

        it = itertools.chain.from_iterable(grouped() for _ in range(num_epochs))
        for _ in range(num_steps):
            batch_idxs = tuple(i for i in next(it) if i is not None)
            bd = defaultdict(list)
            for key, val in self.data.items():
                bd[key].extend(val[i] for i in batch_idxs)
            pid2idx = {}
            for aids in bd["aid"]:
                for aid in aids:
                    for pid in self.shared["albums"][aid]["photo_ids"]:
                        if pid not in pid2idx:
                            pid2idx[pid] = len(pid2idx)
            dim = next(iter(self.shared["pid2feat"].values())).shape[0]
            feats = np.zeros((len(pid2idx), dim), "float32")
            for pid, ix in pid2idx.items():
                feats[ix] = self.shared["pid2feat"][pid]
            bd["pidx2feat"] = feats
            fields = [("album_title", "title"), ("album_title_c", "title_c"), ("album_description", "description"),
                      ("album_description_c", "description_c"), ("where", "where"), ("where_c", "where_c"), ("when", "when"),
                      ("when_c", "when_c"), ("photo_titles", "photo_titles"), ("photo_titles_c", "photo_titles_c"),
                      ("photo_ids", "photo_ids")]
            extra = defaultdict(list)
            for aids in bd["aid"]:
                albums = [self.shared["albums"][aid] for aid in aids]
                for out_key, alb_key in fields:
                    extra[out_key].append([alb[alb_key] for alb in albums])
                extra["photo_idxs"].append([[pid2idx[p] for p in alb["photo_ids"]] for alb in albums])
            bd.update(extra)
            yield batch_idxs, Dataset(bd, self.datatype, shared=self.shared)
