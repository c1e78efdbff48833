"""Host batch assembly: fvta_memexqa_amd.feed.build_feed_dict vs the per-word / per-character loops of the reference's
get_feed_dict (restated in oracle/feed_literal.py), on a synthetic MemexQA-shaped dataset at the README training sizes
(batch 64 here; 4 albums x 8 photos, 8-word photo titles, 16-character words).  CPU only."""
import os, sys, time, random
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import numpy as np
from make_feed_golden import make_dataset, Config, MAXMETA
from fvta_memexqa_amd import utils as U
from fvta_memexqa_amd.feed import build_feed_dict
from oracle import feed_literal as FL

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data, shared = make_dataset(3, n_qa=4 * N, n_albums=200)
thres = dict(sent_album_title_size_thres=8, sent_photo_title_size_thres=8, sent_des_size_thres=10, sent_when_size_thres=4,
             sent_where_size_thres=4, answer_size_thres=5, question_size_thres=25, num_photos_thres=8, num_albums_thres=4,
             word_size_thres=16)
config = Config(batch_size=N, is_train=True, showspecs=False, maxmeta=MAXMETA, **thres)
ds = U.Dataset(data, "train", shared=shared)
U.update_config(config, [ds])
random.seed(0)
batches = [b for _, b in ds.get_batches(N, 4, shuffle=True)]
def run(fn, reps=3):
    ts = []
    for _ in range(reps):
        np.random.seed(0)
        t0 = time.perf_counter()
        for b in batches:
            out = fn(b)
        ts.append((time.perf_counter() - t0) / len(batches))
    return min(ts) * 1e3, out
memo = [None]
def fast(b):
    f, memo[0] = build_feed_dict(config, b, True, 4, memo[0])
    return f
t_fast, f1 = run(fast)
t_lit, f2 = run(lambda b: FL.get_feed_dict(config, b, True, 4))
assert all(np.array_equal(f1[k], f2[k]) for k in f1 if hasattr(f1[k], "shape"))
tok = sum(int(f1[k].sum()) for k in f1 if k.endswith("_mask"))
print("batch %d: %d valid tokens, pts_c %s" % (N, tok, f1["pts_c"].shape))
print("literal loops   %.2f ms / batch" % t_lit)
print("build_feed_dict %.2f ms / batch  (%.1fx)" % (t_fast, t_lit / t_fast))
