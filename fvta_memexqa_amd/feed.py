"""Host batch assembly: the arrays of the reference's feed dict (model_v2.py:1099-1565), built without the
per-word / per-character Python loops that dominate the reference's step time at small batches.

Same outputs, byte for byte (tests/test_feed_golden.py replays fixtures produced by the reference's own
`get_feed_dict`): int32 id arrays, `_c` char-id arrays with a trailing max_word_size axis, bool masks, `y`,
`image_emb_mat`, `existing_emb_mat`, keyed by the reference's placeholder names.

How: every nested list level above the words is walked once per SENTENCE (not per word / per character) to build flat
index vectors; word ids come from a memo over the vocabulary (`word -> id` with the reference's lookup order: as
written, lower, capitalised, upper; trainable vocabulary before the frozen one; model_v2.py:1325-1336) and char rows
from a memo `tuple(chars) -> int32[max_word_size]`; the arrays are filled by one fancy-index assignment per field.
"""
from copy import deepcopy

import numpy as np

TEXT_FIELDS = ("at", "ad", "when", "where", "pts", "q", "choices")


class VocabMemo:
    """Lookup memos for one `shared` dictionary set (word2idx / existing_word2idx / char2idx) and one max_word_size."""

    class _Words(dict):
        def __init__(self, w2i, e2i):
            super().__init__()
            self.w2i, self.e2i, self.shift = w2i, e2i, len(w2i)

        def __missing__(self, word):
            forms = (word, word.lower(), word.capitalize(), word.upper())
            v = 1                                                          # -UNK-
            for f in forms:
                if f in self.w2i:
                    v = self.w2i[f]
                    break
            else:
                for f in forms:
                    if f in self.e2i:
                        v = self.e2i[f] + self.shift
                        break
            self[word] = v
            return v

    class _Chars(dict):
        def __init__(self, c2i, W):
            super().__init__()
            self.c2i, self.W = c2i, W

        def __missing__(self, chars):                                       # chars: tuple of characters
            row = np.zeros(self.W, np.int32)
            get = self.c2i.get
            n = min(len(chars), self.W)
            row[:n] = [get(c, 1) for c in chars[:n]]
            self[chars] = row
            return row

    def __init__(self, shared, W):
        self.key = (id(shared["word2idx"]), len(shared["word2idx"]), id(shared["existing_word2idx"]),
                    len(shared["existing_word2idx"]), id(shared["char2idx"]), len(shared["char2idx"]), W)
        self.words = self._Words(shared["word2idx"], shared["existing_word2idx"])
        self.chars = self._Chars(shared["char2idx"], W)

    @staticmethod
    def key_of(shared, W):
        return (id(shared["word2idx"]), len(shared["word2idx"]), id(shared["existing_word2idx"]),
                len(shared["existing_word2idx"]), id(shared["char2idx"]), len(shared["char2idx"]), W)

    def word_ids(self, words):
        return np.fromiter(map(self.words.__getitem__, words), np.int32, len(words))

    def char_rows(self, words_c):
        if not words_c:
            return np.zeros((0, self.chars.W), np.int32)
        return np.stack(list(map(self.chars.__getitem__, map(tuple, words_c))))


def _longest(seqs):
    return max(map(len, seqs), default=0)


def _sentences(nested, caps):
    """Walk the list levels above the sentence.  nested: lists nested len(caps) deep whose leaves are sentences;
    caps[l] bounds the index at level l.  -> (tuple of index prefixes, sentence)."""
    out = [((), nested)]
    for cap in caps:
        out = [(ix + (j,), child) for ix, node in out for j, child in enumerate(node[:cap])]
    return out


def _fill_text(ids, mask, chars, words, words_c, caps, sent_cap, memo):
    """One text field.  words / words_c: nested lists (sentence = list of words / list of char lists).  The reference
    fills ids+mask from `words` and the char array from `words_c` in separate loops (e.g. model_v2.py:1344-1374), so the
    two sources stay separate here too."""
    for source, is_char in ((words, False), (words_c, True)):
        index_cols, flat = [], []
        sents = _sentences(source, caps)
        lens = [min(len(s), sent_cap) if sent_cap is not None else len(s) for _, s in sents]
        total = sum(lens)
        if total == 0:
            continue
        nd = len(caps)
        prefix = np.empty((total, nd + 1), np.int64)
        pos = 0
        for (ix, s), n in zip(sents, lens):
            if n:
                prefix[pos:pos + n, :nd] = ix
                prefix[pos:pos + n, nd] = np.arange(n)
                flat.extend(s[:n])
                pos += n
        index_cols = tuple(prefix[:, c] for c in range(nd + 1))
        if is_char:
            chars[index_cols] = memo.char_rows(flat)
        else:
            ids[index_cols] = memo.word_ids(flat)
            mask[index_cols] = True


def build_feed_dict(config, batch, is_train=False, num_choice=4, memo=None):
    """-> (feed dict keyed by placeholder name, memo).  `config` needs batch_size and the max_* sizes
    (utils.update_config); `batch` is a utils.Dataset mini-batch (data + shared)."""
    d, sh = batch.data, batch.shared
    g = (lambda k, dflt=None: config.get(k, dflt)) if isinstance(config, dict) else (lambda k, dflt=None: getattr(config, k, dflt))
    N = 2 if g("showspecs", False) else g("batch_size")
    W = g("max_word_size")
    cap_M, cap_JI = g("max_num_albums"), g("max_num_photos")
    # per-batch sizes: the longest item in THIS batch, at least 1, at most the configured cap (model_v2.py:1126-1166)
    one = lambda v: v if v else 1
    M = min(cap_M, one(_longest(d["album_title"])))
    JXA = min(g("max_sent_album_title_size"), one(_longest([t for s in d["album_title"] for t in s])))
    JXP = min(g("max_sent_photo_title_size"), one(_longest([t for s in d["photo_titles"] for a in s for t in a])))
    JD = min(g("max_sent_des_size"), one(_longest([t for s in d["album_description"] for t in s])))
    JG = min(g("max_where_size"), one(_longest([t for s in d["where"] for t in s])))
    JT = min(g("max_when_size"), one(_longest([t for s in d["when"] for t in s])))
    JI = min(cap_JI, one(_longest([a for s in d["photo_ids"] for a in s])))
    JQ = min(g("max_question_size"), one(_longest(d["q"])))
    JA = g("max_answer_size")
    if memo is None or memo.key != VocabMemo.key_of(sh, W):
        memo = VocabMemo(sh, W)

    f = {}
    shapes = dict(at=(N, M, JXA), ad=(N, M, JD), when=(N, M, JT), where=(N, M, JG), pts=(N, M, JI, JXP), q=(N, JQ),
                  choices=(N, num_choice, JA))
    for name in TEXT_FIELDS:
        f[name] = np.zeros(shapes[name], np.int32)
        f[name + "_c"] = np.zeros(shapes[name] + (W,), np.int32)
        f[name + "_mask"] = np.zeros(shapes[name], bool)
    f["pis"], f["pis_mask"] = np.zeros((N, M, JI), np.int32), np.zeros((N, M, JI), bool)
    f["is_train"] = is_train
    f["image_emb_mat"], f["existing_emb_mat"] = d["pidx2feat"], sh["existing_emb_mat"]

    # choices: the correct answer goes to a random slot when training (y marks it), to `yidx` when evaluating
    C, Cc = deepcopy(d["cs"]), deepcopy(d["ccs"])
    if is_train:                                                            # model_v2.py:1270-1288
        f["y"] = np.zeros((N, num_choice), bool)
        slot = np.random.choice(num_choice, N)                              # same draw as the reference (global NumPy RNG)
        n = len(d["y"])
        f["y"][np.arange(n), slot[:n]] = True
        where = slot
    elif "y" in d and "cy" in d and "yidx" in d:                            # model_v2.py:1296-1306
        where = d["yidx"]
    else:
        where = None
    if where is not None:
        for i in range(len(d["y"])):
            assert len(C[i]) == num_choice - 1, "C[i] len:%s" % len(C[i])
            C[i].insert(where[i], d["y"][i])
            Cc[i].insert(where[i], d["cy"][i])
    for ci in C:
        assert len(ci) == num_choice
    for ci in Cc:
        assert len(ci) == num_choice, len(ci)

    # photo indices (model_v2.py:1312-1324)
    rows = _sentences(d["photo_idxs"], (None, cap_M))
    for (i, j), album in rows:
        album = album[:cap_JI]
        if album:
            assert all(isinstance(p, int) for p in album)
            f["pis"][i, j, :len(album)] = album
            f["pis_mask"][i, j, :len(album)] = True

    def text(name, words, words_c, caps, sent_cap):
        _fill_text(f[name], f[name + "_mask"], f[name + "_c"], words, words_c, caps, sent_cap, memo)

    text("at", d["album_title"], d["album_title_c"], (None, cap_M), g("max_sent_album_title_size"))
    text("ad", d["album_description"], d["album_description_c"], (None, cap_M), g("max_sent_des_size"))
    text("when", d["when"], d["when_c"], (None, cap_M), g("max_when_size"))
    text("where", d["where"], d["where_c"], (None, cap_M), g("max_where_size"))
    text("pts", d["photo_titles"], d["photo_titles_c"], (None, cap_M, cap_JI), g("max_sent_photo_title_size"))
    text("choices", C, Cc, (None, None), JA)
    text("q", d["q"], d["cq"], (None,), None)                               # the question is never clipped (:1525)
    return f, memo
