"""End-to-end step time of the reference's own entry (Dataset.get_batches -> Trainer.step) on a synthetic
MemexQA-shaped dataset at the README training sizes (hidden 50, char CNN 100, time warp 5, question attention),
split into host batch assembly, host->device load, and device step."""
import os, sys, time, random
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import numpy as np, torch
from make_feed_golden import make_dataset, Config, MAXMETA
from fvta_memexqa_amd import utils as U
from fvta_memexqa_amd.model_v2 import get_model
from fvta_memexqa_amd.trainer import Trainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data, shared = make_dataset(3, n_qa=8 * N, n_albums=300, idim=2537, wdim=100)
thres = dict(sent_album_title_size_thres=8, sent_photo_title_size_thres=8, sent_des_size_thres=10, sent_when_size_thres=4,
             sent_where_size_thres=4, answer_size_thres=5, question_size_thres=25, num_photos_thres=8, num_albums_thres=4,
             word_size_thres=16)
config = Config(batch_size=N, is_train=True, showspecs=False, maxmeta=MAXMETA, hidden_size=50, simiMatrix=2, add_tanh=True,
                use_question_att=True, use_time_warp=True, warp_type=5, use_char=True, char_emb_size=100, char_out_size=100,
                image_feat_dim=2537, use_image_trans=True, image_trans_dim=100, init_lr=0.5, precision="f32", **thres)
ds = U.Dataset(data, "train", shared=shared)
U.update_config(config, [ds])
config.word_vocab_size = len(shared["word2idx"]) + 2
config.char_vocab_size = len(shared["char2idx"]) + 2
model = get_model(config)
trainer = Trainer(model, config)
random.seed(0)
batches = list(ds.get_batches(N, 24, shuffle=True))
def sync(): torch.cuda.synchronize()
for b in batches[:8]:
    trainer.step(None, b)
sync()
t_feed = t_load = t_dev = 0.0
for b in batches[8:]:
    t0 = time.perf_counter(); feed = model.get_feed_dict(b[1], is_train=True)
    t1 = time.perf_counter(); L = model.load_inputs(feed, training=True); sync()
    t2 = time.perf_counter(); trainer.step_device(L); sync()
    t3 = time.perf_counter()
    t_feed += t1 - t0; t_load += t2 - t1; t_dev += t3 - t2
n = len(batches) - 8
print("batch %d: get_feed_dict %.2f ms, load_inputs (host->device, layout) %.2f ms, device step %.2f ms  -> %.0f QA-pairs/s end to end"
      % (N, t_feed / n * 1e3, t_load / n * 1e3, t_dev / n * 1e3, N / ((t_feed + t_load + t_dev) / n)))
print("distinct layouts cached:", len(model._layouts))
