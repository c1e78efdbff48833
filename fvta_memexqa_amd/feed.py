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
from itertools import chain

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
        """tuple of characters -> row of `table` (int32 [rows, W], grown by doubling)"""

        def __init__(self, c2i, W):
            super().__init__()
            self.c2i, self.W = c2i, W
            self.table = np.zeros((256, W), np.int32)
            self.n = 0

        def __missing__(self, chars):
            if self.n == len(self.table):
                self.table = np.concatenate([self.table, np.zeros_like(self.table)])
            get = self.c2i.get
            n = min(len(chars), self.W)
            self.table[self.n, :n] = [get(c, 1) for c in chars[:n]]
            self[chars] = self.n
            self.n += 1
            return self.n - 1

    def __init__(self, shared, W):
        self.key = self.key_of(shared, W)
        self.words = self._Words(shared["word2idx"], shared["existing_word2idx"])
        self.chars = self._Chars(shared["char2idx"], W)

    @staticmethod
    def key_of(shared, W):
        return (id(shared["word2idx"]), len(shared["word2idx"]), id(shared["existing_word2idx"]),
                len(shared["existing_word2idx"]), id(shared["char2idx"]), len(shared["char2idx"]), W)

    def word_ids(self, words):
        return np.fromiter(map(self.words.__getitem__, words), np.int32, len(words))

    def char_rows(self, words_c):
        rows = np.fromiter(map(self.chars.__getitem__, map(tuple, words_c)), np.int64, len(words_c))
        return self.chars.table[rows]


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
    nd = len(caps)
    for source, is_char in ((words, False), (words_c, True)):
        sents = _sentences(source, caps)
        if not sents:
            continue
        lens = np.fromiter((len(s) for _, s in sents), np.int64, len(sents))
        if sent_cap is not None:
            np.minimum(lens, sent_cap, out=lens)
        total = int(lens.sum())
        if total == 0:
            continue
        lead = np.asarray([ix for ix, _ in sents], np.int64).reshape(len(sents), nd)
        start = np.cumsum(lens) - lens
        cols = tuple(np.repeat(lead[:, c], lens) for c in range(nd)) + (np.arange(total) - np.repeat(start, lens),)
        flat = list(chain.from_iterable(s[:n] for (_, s), n in zip(sents, lens.tolist())))
        if is_char:
            chars[cols] = memo.char_rows(flat)
        else:
            ids[cols] = memo.word_ids(flat)
            mask[cols] = True


class AlbumTables:
    """Every album of `shared['albums']` converted ONCE (lazily, on first sight) into fixed-size id / char / length
    arrays cut at the configured caps; a batch then gathers rows by album index instead of walking words.  Valid because
    an album's texts are the same objects in every batch (utils.Dataset._mini_batch hands out shared['albums'][aid][...])
    and because truncating at the cap commutes with truncating at the smaller per-batch size."""

    FIELDS = (("at", "title", "max_sent_album_title_size"), ("ad", "description", "max_sent_des_size"),
              ("when", "when", "max_when_size"), ("where", "where", "max_where_size"))

    def __init__(self, g, memo):
        self.memo = memo
        self.W = g("max_word_size")
        self.caps = {name: g(cap) for name, _, cap in self.FIELDS}
        self.cap_JI, self.cap_JXP = g("max_num_photos"), g("max_sent_photo_title_size")
        self.key = (memo.key, tuple(sorted(self.caps.items())), self.cap_JI, self.cap_JXP)
        self.row = {}                  # album id -> row
        self.n = 0
        self.cap = 0
        self.t = {}

    def _grow(self, need):
        if need <= self.cap:
            return
        cap = max(64, 2 * self.cap, need)
        W, JI, JXP = self.W, self.cap_JI, self.cap_JXP
        shapes = {}
        for name, _, _ in self.FIELDS:
            c = self.caps[name]
            shapes[name] = ((c,), np.int32)
            shapes[name + "_c"] = ((c, W), np.int32)
            shapes[name + "_len"] = ((), np.int32)        # min(len, cap): mask length
            shapes[name + "_raw"] = ((), np.int32)        # uncapped: feeds the per-batch maxima
        shapes.update(pts=((JI, JXP), np.int32), pts_c=((JI, JXP, W), np.int32), pts_len=((JI,), np.int32),
                      pts_raw=((), np.int32), nph_raw=((), np.int32))
        for k, (shp, dt) in shapes.items():
            new = np.zeros((cap + 1,) + shp, dt)       # row `cap` stays all-zero: the "no album" row
            if k in self.t:
                new[:self.n] = self.t[k][:self.n]
            self.t[k] = new
        self.cap = cap

    def rows_of(self, albums, aids):
        """album ids -> rows, converting the albums seen for the first time"""
        out = []
        for aid in aids:
            r = self.row.get(aid)
            if r is None:
                self._grow(self.n + 1)
                r = self.row[aid] = self.n
                self.n += 1
                self._convert(albums[aid], r)
            out.append(r)
        return out

    def _convert(self, alb, r):
        t, memo = self.t, self.memo
        for name, key, _ in self.FIELDS:
            words, chars, cap = alb[key], alb[key + "_c"], self.caps[name]
            n = min(len(words), cap)
            if n:
                t[name][r, :n] = memo.word_ids(words[:n])
            nc = min(len(chars), cap)
            if nc:
                t[name + "_c"][r, :nc] = memo.char_rows(chars[:nc])
            t[name + "_len"][r], t[name + "_raw"][r] = n, len(words)
        titles, titles_c = alb["photo_titles"], alb["photo_titles_c"]
        t["nph_raw"][r] = len(alb["photo_ids"])
        t["pts_raw"][r] = max(map(len, titles), default=0)
        for k, title in enumerate(titles[:self.cap_JI]):
            n = min(len(title), self.cap_JXP)
            if n:
                t["pts"][r, k, :n] = memo.word_ids(title[:n])
            t["pts_len"][r, k] = n
        for k, title_c in enumerate(titles_c[:self.cap_JI]):
            n = min(len(title_c), self.cap_JXP)
            if n:
                t["pts_c"][r, k, :n] = memo.char_rows(title_c[:n])


def _album_fields_from_tables(f, batch, g, N, tables):
    """at / ad / when / where / pts (+ _c, _mask) of the feed by gathering album rows; returns False when the batch does
    not carry album ids (a hand-made batch): the caller then walks the nested lists instead."""
    d, sh = batch.data, batch.shared
    aids = d.get("aid")
    if aids is None or "albums" not in sh or any(a not in sh["albums"] for s_ in aids for a in s_):
        return False
    cap_M = g("max_num_albums")
    flat = tables.rows_of(sh["albums"], [a for s_ in aids for a in s_])
    t = tables.t
    zero = tables.cap                                               # the all-zero row
    counts = [len(s_) for s_ in aids]
    allrows = np.asarray(flat, np.int64)
    one = lambda v: int(v) if v else 1
    M = min(cap_M, one(max(counts, default=0)))
    idx = np.full((N, M), zero, np.int64)
    pos = 0
    for i, c in enumerate(counts):
        m = min(c, M)
        idx[i, :m] = flat[pos:pos + m]
        pos += c
    raw_max = lambda k: int(t[k][allrows].max()) if len(allrows) else 0
    for name, _, cap in AlbumTables.FIELDS:
        J = min(tables.caps[name], one(raw_max(name + "_raw")))
        f[name] = t[name][idx][:, :, :J]
        f[name + "_c"] = t[name + "_c"][idx][:, :, :J]
        f[name + "_mask"] = np.arange(J) < t[name + "_len"][idx][..., None]
    JI = min(tables.cap_JI, one(raw_max("nph_raw")))
    JXP = min(tables.cap_JXP, one(raw_max("pts_raw")))
    f["pts"] = np.ascontiguousarray(t["pts"][idx][:, :, :JI, :JXP])
    f["pts_c"] = np.ascontiguousarray(t["pts_c"][idx][:, :, :JI, :JXP])
    f["pts_mask"] = np.arange(JXP) < t["pts_len"][idx][:, :, :JI, None]
    for name, _, _ in AlbumTables.FIELDS:
        f[name], f[name + "_c"] = np.ascontiguousarray(f[name]), np.ascontiguousarray(f[name + "_c"])
    return True


def build_feed_dict(config, batch, is_train=False, num_choice=4, memo=None, use_tables=True):
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
    tables = None
    if use_tables:
        tables = getattr(memo, "tables", None)
        if tables is None or tables.key[1:] != AlbumTables(g, memo).key[1:]:
            tables = memo.tables = AlbumTables(g, memo)
        if not _album_fields_from_tables(f, batch, g, N, tables):
            tables = None
    shapes = dict(at=(N, M, JXA), ad=(N, M, JD), when=(N, M, JT), where=(N, M, JG), pts=(N, M, JI, JXP), q=(N, JQ),
                  choices=(N, num_choice, JA))
    for name in TEXT_FIELDS:
        if tables is not None and name in ("at", "ad", "when", "where", "pts"):
            assert f[name].shape == shapes[name], (name, f[name].shape, shapes[name])
            continue
        f[name] = np.zeros(shapes[name], np.int32)
        f[name + "_c"] = np.zeros(shapes[name] + (W,), np.int32)
        f[name + "_mask"] = np.zeros(shapes[name], bool)
    f["pis"], f["pis_mask"] = np.zeros((N, M, JI), np.int32), np.zeros((N, M, JI), bool)
    f["is_train"] = is_train
    f["image_emb_mat"], f["existing_emb_mat"] = d["pidx2feat"], sh["existing_emb_mat"]

    # choices: the correct answer goes to a random slot when training (y marks it), to `yidx` when evaluating
    C, Cc = [list(ci) for ci in d["cs"]], [list(ci) for ci in d["ccs"]]    # per-QA lists are copied (the answer is inserted into
    #                                                                        them), their sentences are shared: same effect
    #                                                                        as the reference's deepcopy (:1249-1250)
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

    if tables is None:
        text("at", d["album_title"], d["album_title_c"], (None, cap_M), g("max_sent_album_title_size"))
        text("ad", d["album_description"], d["album_description_c"], (None, cap_M), g("max_sent_des_size"))
        text("when", d["when"], d["when_c"], (None, cap_M), g("max_when_size"))
        text("where", d["where"], d["where_c"], (None, cap_M), g("max_where_size"))
        text("pts", d["photo_titles"], d["photo_titles_c"], (None, cap_M, cap_JI), g("max_sent_photo_title_size"))
    text("choices", C, Cc, (None, None), JA)
    text("q", d["q"], d["cq"], (None,), None)                               # the question is never clipped (:1525)
    return f, memo
