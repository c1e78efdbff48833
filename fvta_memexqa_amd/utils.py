"""Host-side counterpart of the reference's utils.py for the batch path: `Dataset` with
`get_batches` (utils.py:31-198), `grouper` (12-18), `update_config` (294-396), `getAnswers` / `getEvalScore`
(247-292).  Python 3, same names, arguments and outputs; what a batch holds is pinned against the reference's own
code by tests/test_feed_golden.py.  The TF helpers of that file (`exp_mask`, `flatten`, `reconstruct`) have no
counterpart: the kernels inline them.
"""
import math
import random
from itertools import chain, zip_longest

import numpy as np

# album record field -> key of the per-batch list (utils.py:143-195)
_ALBUM_FIELDS = (("title", "album_title"), ("title_c", "album_title_c"), ("description", "album_description"),
                 ("description_c", "album_description_c"), ("where", "where"), ("where_c", "where_c"),
                 ("when", "when"), ("when_c", "when_c"), ("photo_titles", "photo_titles"),
                 ("photo_titles_c", "photo_titles_c"), ("photo_ids", "photo_ids"))


def grouper(l, n):
    """chunks of n, the last one filled with None (utils.py:12-18)."""
    return list(zip_longest(*([iter(l)] * n), fillvalue=None))


def sec2time(secs):
    m, s = divmod(secs, 60)
    h, m = divmod(m, 60)
    return ("%02d:%02d:%.3f" if s >= 10.0 else "%02d:%02d:0%.3f") % (h, m, s)


class Dataset:
    """`data`: dict of equally long per-QA lists (q, cq, y, cy, yidx, cs, ccs, aid, qid, idxs); `shared`: albums,
    pid2feat, word2idx, char2idx, existing_word2idx, existing_emb_mat.  A mini-batch is again a Dataset whose `data`
    additionally carries the album texts of its QA pairs and `pidx2feat`, the photo features this batch touches."""

    def __init__(self, data, datatype, shared=None, valid_idxs=None):
        self.data = data
        self.datatype = datatype
        self.shared = shared
        self.valid_idxs = range(self.get_data_size()) if valid_idxs is None else valid_idxs
        self.num_examples = len(self.valid_idxs)

    def get_data_size(self):
        return len(next(iter(self.data.values())))

    def get_by_idxs(self, idxs):
        return {key: [val[i] for i in idxs] for key, val in self.data.items()}

    def _mini_batch(self, batch_idxs):
        albums = self.shared["albums"]
        bd = self.get_by_idxs(batch_idxs)
        per_qa = [[albums[aid] for aid in aids] for aids in bd["aid"]]
        # batch-local photo index, in order of first appearance (utils.py:128-134)
        pid2idx = {}
        for qa in per_qa:
            for alb in qa:
                for pid in alb["photo_ids"]:
                    pid2idx.setdefault(pid, len(pid2idx))
        p2f = self.shared["pid2feat"]
        dim = next(iter(p2f.values())).shape[0]
        feats = np.zeros((len(pid2idx), dim), dtype="float32")
        if pid2idx:
            feats[list(pid2idx.values())] = [p2f[pid] for pid in pid2idx]
        bd["pidx2feat"] = feats
        for field, key in _ALBUM_FIELDS:
            bd[key] = [[alb[field] for alb in qa] for qa in per_qa]
        bd["photo_idxs"] = [[[pid2idx[pid] for pid in alb["photo_ids"]] for alb in qa] for qa in per_qa]
        return Dataset(bd, self.datatype, shared=self.shared)

    def get_batches(self, batch_size, num_steps, shuffle=True, cap=False, rank=0, world=1, seed=None):
        """yields (batch_idxs, Dataset) num_steps times; every epoch walks the same (shuffled once) order; `cap` limits
        the run to one epoch (utils.py:89-198).

        Data parallelism (no counterpart in the reference): with world > 1 every rank must pass the same
        `seed`; all ranks then walk ONE shuffled order in global batches of batch_size * world and rank r takes the r-th
        contiguous batch_size slice of each (a short last group is dealt round-robin instead, every rank gets at least one
        example; fewer examples than ranks raise), so an epoch covers every example exactly once across ranks.  The
        single-process call (world = 1, seed None) is the reference's: one draw from the global `random`."""
        if world > 1 and shuffle and seed is None:
            raise ValueError("get_batches: data-parallel ranks need a common shuffle seed")
        gbs = batch_size * world
        tail = self.num_examples % gbs
        if world > 1 and 0 < tail < world and num_steps >= int(math.ceil(self.num_examples / float(gbs))):
            # (checked before the first batch: the short last group would otherwise end an epoch of training with every
            #  rank crashing)
            raise ValueError("get_batches: the last global batch of an epoch would hold %d examples for %d ranks "
                             "(num_examples %% (batch_size * world) must be 0 or >= world)" % (tail, world))
        per_epoch = int(math.ceil(self.num_examples / float(gbs)))
        if cap and num_steps > per_epoch:
            num_steps = per_epoch
        num_epochs = int(math.ceil(num_steps / float(per_epoch)))
        rng = random if seed is None else random.Random(seed)
        order = rng.sample(list(self.valid_idxs), len(self.valid_idxs)) if shuffle else list(self.valid_idxs)
        groups = chain.from_iterable(grouper(order, gbs) for _ in range(num_epochs))
        for _ in range(num_steps):
            group = [i for i in next(groups) if i is not None]
            if world > 1 and len(group) < gbs:
                # the short last global group of an epoch: dealt round-robin, so that EVERY rank gets a batch (a rank
                # with no examples would leave the step's collectives one participant short)
                if len(group) < world:
                    raise ValueError("get_batches: %d examples left for %d ranks (num_examples %% (batch_size * world) "
                                     "must be 0 or >= world)" % (len(group), world))
                batch_idxs = tuple(group[rank::world])
            else:
                batch_idxs = tuple(group[rank * batch_size:(rank + 1) * batch_size])
            yield batch_idxs, self._mini_batch(batch_idxs)


def getAnswers(yp, batch):
    """qid -> predicted choice, qid -> correct choice (utils.py:247-260)."""
    pred, real = {}, {}
    for qid, yidx, ypi in zip(batch[1].data["qid"], batch[1].data["yidx"], yp):
        pred[qid] = int(np.argmax(ypi))
        real[qid] = yidx
        assert yidx < 4 and pred[qid] < 4
    return pred, real


def getAnswers_yp(yp, batch):
    pred, real = getAnswers(yp, batch)
    return pred, real, {qid: ypi for qid, ypi in zip(batch[1].data["qid"], yp)}


def getEvalScore(pred, gt):
    assert len(pred) == len(gt) and len(pred) > 0
    return sum(1 for qid in pred if pred[qid] == gt[qid]) / float(len(pred))


def update_config(config, datasets, showMeta=False):
    """The max_* sizes get_feed_dict allocates with: maxima over the given datasets, clipped by the *_thres flags
    (all of them when training, description / photo-title / word size always) (utils.py:294-396)."""
    longest = lambda seqs: max((len(s) for s in seqs), default=0)
    mx = dict.fromkeys(("num_albums", "num_photos", "sent_album_title_size", "sent_photo_title_size", "sent_des_size",
                        "when_size", "where_size", "answer_size", "question_size", "word_size"), 0)

    def up(key, v):
        if v > mx[key]:
            mx[key] = v

    for ds in datasets:
        for idx in ds.valid_idxs:
            q, y, cs = ds.data["q"][idx], ds.data["y"][idx], ds.data["cs"][idx]
            up("question_size", len(q))
            up("word_size", max(len(w) for w in q))
            for sent in cs + [y]:
                up("answer_size", len(sent))
                up("word_size", max(len(w) for w in sent))
            albums = [ds.shared["albums"][aid] for aid in ds.data["aid"][idx]]
            up("num_albums", len(albums))
            for alb in albums:
                up("num_photos", len(alb["photo_ids"]))
                up("sent_album_title_size", len(alb["title"]))
                for title in alb["photo_titles"]:
                    if title:
                        up("sent_photo_title_size", len(title))
                        up("word_size", longest(title))
                if alb["description"]:
                    up("sent_des_size", len(alb["description"]))
                    up("word_size", longest(alb["description"]))
                up("when_size", len(alb["when"]))
                up("word_size", max(len(w) for w in alb["title"]))
                up("word_size", max(len(w) for w in alb["when"]))
                if alb["where"]:
                    up("word_size", longest(alb["where"]))
                    up("where_size", len(alb["where"]))
    for k, v in mx.items():
        setattr(config, "max_" + k, v)
    if showMeta:
        print("max meta:\n\t" + " ,".join("%s:%s" % (k, getattr(config, k)) for k in config.maxmeta))
    clip = lambda name, thres: setattr(config, name, min(getattr(config, name), getattr(config, thres)))
    if config.is_train:
        clip("max_num_albums", "num_albums_thres"), clip("max_num_photos", "num_photos_thres")
        clip("max_sent_album_title_size", "sent_album_title_size_thres")
        clip("max_when_size", "sent_when_size_thres"), clip("max_where_size", "sent_where_size_thres")
        clip("max_answer_size", "answer_size_thres")
    clip("max_sent_photo_title_size", "sent_photo_title_size_thres")
    clip("max_sent_des_size", "sent_des_size_thres")
    clip("max_word_size", "word_size_thres")
    config.char_vocab_size = len(datasets[0].shared["char2idx"])
    config.word_emb_size = len(next(iter(datasets[0].shared["word2vec"].values())))
    config.word_vocab_size = len(datasets[0].shared["word2idx"])
