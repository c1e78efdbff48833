"""Host-side mirror of the reference's `model_v2.py` surface for the FVTA hot path.

What the reference builds as a TF-1 graph (model_v2.py:361-1096) is here a thin
Python object that owns device buffers and calls the HIP kernels of
libfvta_hip.so in order.  No arithmetic of the path runs in Python/PyTorch:
torch owns memory, streams and `torch.distributed`.

Surface kept from the reference:
  get_model(config) -> Model                      model_v2.py:12-17
  Model.loss / .yp / .logits / .global_step       model_v2.py:366, 1082-1095
  Model.att_logits / .q_att_logits / .hall        model_v2.py:914, 1022, 1045 (vis)
  parameter names of the TF checkpoint / weights.npz (main.py:578-588)

Three entry levels, all through load_inputs():
  * the reference's own feed (Dataset mini-batch -> get_feed_dict -> id / char / photo-index arrays,
    model_v2.py:1099-1565), with the embedding front-end (524-645) inside the step -- needs the vocabulary sizes in
    the config;
  * the same index arrays as an `inputs` tree (synth.make_token_inputs);
  * the ENCODER INPUTS x* of model_v2.py:680-688 (embedded tokens / photo features + masks) -- the bench headline.
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from ._lib import F32, BF16, BF16X3

SUPPORTED_W = (64, 128, 256, 512, 1024, 2048)


def get_model(config):
    """model_v2.py:12-17."""
    return Model(config, "model_%s" % getattr(config, "modelname", "fvta"))


def _cfg(config, name, default):
    if isinstance(config, dict):
        return config.get(name, default)
    return getattr(config, name, default)


def padded_hidden(d):
    """Hidden size the kernels run at: w = 2*d_pad must be a supported width.
    Zero-padded units are exact: their gates see z = 0 -> c = h = 0, and every
    gradient into a padded parameter is 0."""
    for w in SUPPORTED_W:
        if 2 * d <= w:
            return w // 2
    raise ValueError("hidden_size %d too large (max %d)" % (d, SUPPORTED_W[-1] // 2))


class ParamStore:
    """All trainables in ONE flat fp32 device buffer (+ one flat gradient buffer):
    a single RCCL all-reduce bucket and a single optimiser launch per step."""

    def __init__(self, specs, device, late=()):
        """`late`: names whose gradients are final only at the very end of the backward pass (the text cell and what
        hangs off its dx); they are laid out LAST, so that [0, early_numel) -- scorer, attention, photo cell -- is one
        contiguous bucket whose all-reduce can start while the text cell's recurrence still runs."""
        order = [n for n in specs if n not in late] + [n for n in specs if n in late]
        self.specs = {n: specs[n] for n in order}  # name -> shape (padded)
        self.offsets = {}
        off = 0
        self.early_numel = 0
        for name, shape in self.specs.items():
            self.offsets[name] = off
            off += (int(np.prod(shape)) + 63) // 64 * 64  # 256-byte aligned slices
            if name not in late:
                self.early_numel = off
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)

    def view(self, name, grad=False, flat=None):
        """`flat`: another buffer with this store's layout (an optimiser slot)"""
        off, shape = self.offsets[name], self.specs[name]
        buf = flat if flat is not None else (self.grad if grad else self.flat)
        return buf[off:off + int(np.prod(shape))].view(*shape)


class _Layout:
    """Per batch-shape cache: arenas, offset tables and kernel handles."""
    pass


class Model:
    # reference checkpoint names
    N_TEXT_K = "reader/text/utext/%s/basic_lstm_cell/kernel"
    N_TEXT_B = "reader/text/utext/%s/basic_lstm_cell/bias"
    N_IMG_K = "reader/image/uimage/%s/basic_lstm_cell/kernel"
    N_IMG_B = "reader/image/uimage/%s/basic_lstm_cell/bias"
    N_ATT_W, N_ATT_B = "attention/all/att_logits/W", "attention/all/att_logits/b"
    N_QATT_W, N_QATT_B = "question_emb/question_att/att_logits/W", "question_emb/question_att/att_logits/b"
    N_OUT_W, N_OUT_B = "output/choicelogits/W", "output/choicelogits/b"
    N_TW_WH_W, N_TW_WH_B = "time_warp/WH/W", "time_warp/WH/b"
    N_TW_WC_W, N_TW_WC_B = "time_warp/WC/W", "time_warp/WC/b"
    # embedding front-end (model_v2.py:524-645), present when the config carries the vocabulary sizes
    N_CHAR_EMB = "emb/var/char_emb"
    N_CONV_F, N_CONV_B = "emb/conv/conv1d/filter", "emb/conv/conv1d/bias"
    N_WORD_EMB = "emb/word/var/word_emb_mat"
    N_IMGT_W = "emb/image/image_transform/image_trans_linear/W"
    N_IMGT_B = "emb/image/image_transform/image_trans_linear/b"

    def __init__(self, config, scope="model", text_in=None, img_in=None, device=None):
        self.scope = scope
        self.config = config
        self.dev = device or ops.require_gpu()
        self.d = int(_cfg(config, "hidden_size", 100))
        self.dp = padded_hidden(self.d)
        self.w, self.wp = 2 * self.d, 2 * self.dp
        self.N = _cfg(config, "batch_size", None)
        self.num_choice = int(_cfg(config, "num_choice", 4))      # model_v2.py:381 reads an undefined flag; model.py:353 says 4
        self.simi = int(_cfg(config, "simiMatrix", 1))
        self.add_tanh = bool(_cfg(config, "add_tanh", False))
        self.scorer_tanh = self.add_tanh     # model_v2.py:1073 (with use_eu_output); model.py's scorer has none
        self.use_question_att = bool(_cfg(config, "use_question_att", False))
        self.use_eu_output = bool(_cfg(config, "use_eu_output", False))
        self.share_fw_bw = bool(_cfg(config, "share_fw_bw", True))
        self.precision = {"f32": F32, "bf16": BF16, "bf16x3": BF16X3}[_cfg(config, "precision", "f32")]
        # shadow rows (bf16 engine): the context tensor hall (model_v2.py:863-914) is never stored in fp32 -- the focal
        # attention reads the bf16 rows the encoders keep for their own recurrence (ops.BiLstm.shadow_rows).  Per layout,
        # where the shadow kernels cover the shape (_shadow_covers); config "shadow_rows", else FVTA_SHADOW_ROWS, else ON
        sr = _cfg(config, "shadow_rows", None)
        self.shadow_rows = bool(int(os.environ.get("FVTA_SHADOW_ROWS", "1"))) if sr is None else bool(sr)
        self.wd = float(_cfg(config, "wd", None) or 0.0)                # --wd (main.py:105); None / 0.0: no l2 terms
        # d logits of softmax_cross_entropy_with_logits (model_v2.py:1088): True = what TF-1's kernel returns,
        # softmax - labels on every row, all-False label rows (padded rows of a short batch, :1270) included
        self.tf_xent_grad = bool(_cfg(config, "tf_xent_grad", True))
        # --keep_prob (main.py:102): DropoutWrapper(cell, input_keep_prob) on both cells while training (model_v2.py:657-661);
        # evaluation layouts never drop (main.py:191 forces 1.0, and :657 gates on is_train)
        self.keep_prob = float(_cfg(config, "keep_prob", 1.0))
        if not 0.0 < self.keep_prob <= 1.0:
            raise ValueError("keep_prob %r is not in (0, 1]" % self.keep_prob)
        self.dropout_seed = int(_cfg(config, "dropout_seed", 0))
        self._dropout_calls = 0      # persisted by Trainer.save / restore: a resumed run continues the mask sequence
        from . import dist as _dist   # data-parallel ranks draw DIFFERENT masks for their shards
        self._dropout_rank_salt = (_dist.world()[1] * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        self.use_time_warp = bool(_cfg(config, "use_time_warp", False))
        self.warp_type = int(_cfg(config, "warp_type", 1))
        self.window_t = float(_cfg(config, "window_t", 3.0))     # time_warp_window_t, init 3.0 (model_v2.py:333); no gradient
        if self.use_time_warp and self.warp_type not in (1, 2, 3, 4, 5):
            raise Exception("time warping type not implemented")    # model_v2.py:341
        # model_v2.py:1020 passes C (defined only inside the use_time_warp block, :995) to attention_3d: without the
        # time warp the reference's graph construction fails on the undefined name
        self.use_time_warp_att = bool(_cfg(config, "use_time_warp_att", False))
        if self.use_time_warp_att and not self.use_time_warp:
            raise NameError("name 'C' is not defined (use_time_warp_att needs use_time_warp, model_v2.py:995, 1020)")
        if _cfg(config, "use_bidirection", False):
            # model_v2.py:1020 hands bidirect=True to attention_3d, whose branch (:281-292) concatenates h_a [N,w] with
            # q_a [N,T,w]: TF refuses the graph.  (The 1-D `attention(..., bidirect=True)` of model.py's baselines is in
            # functional.py.)
            raise ValueError("Shape must be rank 2 but is rank 3: attention_3d's bidirect branch (model_v2.py:281-292) "
                             "cannot be built; use_bidirection only works with model.py's 1-D attentions")
        if self.simi not in (1, 2, 3, 4):
            raise ValueError("similarity matrix not implemented")    # model_v2.py:255-257 (sys.exit there)
        self.text_in = int(text_in if text_in is not None else _cfg(config, "text_in", 200))
        self.img_in = int(img_in if img_in is not None else _cfg(config, "img_in", 100))
        # token-id entry (the reference's own feed: word / char ids + photo indices, model_v2.py:524-645): enabled by
        # the vocabulary sizes in the config; the encoder input widths then follow from the embedding sizes
        self.token_mode = _cfg(config, "word_vocab_size", None) is not None
        if self.token_mode:
            self.VW = int(_cfg(config, "word_vocab_size", 0))
            self.wdim = int(_cfg(config, "word_emb_size", 100))
            self.use_char = bool(_cfg(config, "use_char", False))
            self.VC = int(_cfg(config, "char_vocab_size", 0))
            self.Wc = int(_cfg(config, "max_word_size", 16))
            self.cdim = int(_cfg(config, "char_emb_size", 8))
            self.cwdim = int(_cfg(config, "char_out_size", 100)) if self.use_char else 0
            self.idim = int(_cfg(config, "image_feat_dim", 2537))
            self.use_image_trans = bool(_cfg(config, "use_image_trans", False))
            self.tdim = int(_cfg(config, "image_trans_dim", 100)) if self.use_image_trans else self.idim
            self.text_in, self.img_in = self.wdim + self.cwdim, self.tdim
            self.existing_emb_mat = torch.zeros(0, self.wdim, dtype=torch.float32, device=self.dev)
        # encoder input widths are zero padded to a multiple of 8 (16-byte bf16 operand pieces); exact, like
        # the hidden padding: padded input columns are 0 and their kernel rows stay 0
        self.text_in_p = (self.text_in + 7) // 8 * 8
        self.img_in_p = (self.img_in + 7) // 8 * 8
        self.global_step = 0                                            # model_v2.py:366
        self.loss = self.yp = self.logits = None
        self.att_logits = self.q_att_logits = None
        self._hall_layout = None
        self._layouts = {}
        self.max_layouts = int(_cfg(config, "max_cached_layouts", 4))
        # the side stream (photo cell, small launches that run beside the text cell) must own a hardware queue of
        # its own; which torch stream does is measured once here against the current stream (ops.pick_side_stream)
        # HIGH priority: the photo cell is a chain of tiny launches (a handful of workgroups each) that must never queue
        # behind the text cell's chip-filling kernels -- at normal priority its 40 backward launches averaged 199 us
        # against 70 us unloaded and the photo cell, not the text cell, ended the step (profiles/README.md, r02 timeline)
        try:
            hi = torch.cuda.Stream.priority_range()[1]
        except Exception:
            hi = 0
        self._side, self.side_stream_ratio = ops.pick_side_stream(self.dev, priority=int(_cfg(config, "side_stream_priority", hi)))
        # the photo cell's FORWARD on the main stream, in front of the text cell, instead of beside it (its backward chain
        # stays on the side stream): the weights-in-registers step kernel takes a whole CU per workgroup, so two cells'
        # launches side by side only take CUs from each other
        self.serial_photo_forward = bool(_cfg(config, "serial_photo_forward", False))
        dp, wp = self.dp, self.wp
        F = {1: 3 * wp, 2: 2 * wp, 3: 4 * wp, 4: 0}[self.simi]
        dirs = ["fw"] if self.share_fw_bw else ["fw", "bw"]
        specs = {}
        for dr in dirs:
            specs[self.N_TEXT_K % dr] = (self.text_in_p + dp, 4 * dp)
            specs[self.N_TEXT_B % dr] = (4 * dp,)
        for dr in dirs:
            specs[self.N_IMG_K % dr] = (self.img_in_p + dp, 4 * dp)
            specs[self.N_IMG_B % dr] = (4 * dp,)
        specs.update(self._attention_param_specs(F))
        if self.use_time_warp:     # model_v2.py:986-989: WH [4d -> 2d], WC [2d -> 1]
            specs[self.N_TW_WH_W], specs[self.N_TW_WH_B] = (2 * wp, wp), (wp,)
            specs[self.N_TW_WC_W], specs[self.N_TW_WC_B] = (wp,), (1,)
        specs[self.N_OUT_W] = ((7 if self.use_eu_output else 5) * self._scorer_width(),)
        specs[self.N_OUT_B] = (1,)
        self._plain = set()     # parameters stored exactly in the reference's shape (no hidden-size padding)
        if self.token_mode:
            specs[self.N_WORD_EMB] = (max(self.VW, 1), self.wdim)
            if self.cwdim:
                specs[self.N_CHAR_EMB] = (self.VC, self.cdim)
                specs[self.N_CONV_F] = (5, self.cdim, self.cwdim)      # the reference's [1, 5, cdim, cwdim]
                specs[self.N_CONV_B] = (self.cwdim,)
            if self.use_image_trans:
                specs[self.N_IMGT_W], specs[self.N_IMGT_B] = (self.idim, self.tdim), (self.tdim,)
            self._plain = {self.N_WORD_EMB, self.N_CHAR_EMB, self.N_CONV_F, self.N_CONV_B, self.N_IMGT_W, self.N_IMGT_B}
        late = {n for n in specs if "/utext/" in n} | self._plain      # text cell; embeddings and photo transform (off dx)
        self.params = ParamStore(specs, self.dev, late=late)
        self.early_work = None      # pending all-reduce of the early gradient bucket (data parallelism), see backward()
        self.early_allreduce = bool(_cfg(config, "early_allreduce", False))
        self.init_parameters(int(_cfg(config, "weight_seed", 42)))
        self._loss_buf = torch.zeros(1, dtype=torch.float32, device=self.dev)

    def _scorer_width(self):
        """width of the three vectors the scorer combines (padded layout)"""
        return self.wp

    def _attention_param_specs(self, F):
        """name -> shape of the attention block's variables (F = features per att_logits/W; 0: cosine, none).  The
        model.py-style subclass (fvta_memexqa_amd/model.py) has its own set."""
        specs = {}
        if F:
            specs[self.N_ATT_W], specs[self.N_ATT_B] = (F,), (1,)
            if self.use_question_att:
                specs[self.N_QATT_W], specs[self.N_QATT_B] = (F,), (1,)
        return specs

    def wd_multipliers(self):
        """How many add_wd calls cover each variable (model_v2.py:347-354 adds one l2 term per trainable of the
        CURRENT scope per call): the reader scope once (:835-836: both LSTM cells), attention/all and
        question_emb/question_att once each (:295-296, :198-199), image_trans_linear once (:96-97 via :645), and the
        shared char-CNN filter/bias once per conv1d call -- seven (:564-571).  Embedding tables, the scorer and the time
        warp carry no wd argument."""
        out = {}
        for name in self.params.specs:
            if name.startswith("reader/") or name.startswith("attention/all/") or name.startswith("question_emb/question_att/") \
                    or name.startswith("emb/image/image_transform/image_trans_linear/"):
                out[name] = 1
            elif name.startswith("emb/conv/conv1d/"):
                out[name] = 7
        return out

    def _apply_wd(self, grads, loss):
        """l2 terms of the "losses" collection: into params.grad (grads=True) and/or added to the loss scalar"""
        for name, mult in self.wd_multipliers().items():
            ops.weight_decay(self.params.view(name).reshape(-1), self.params.view(name, True).reshape(-1) if grads else None,
                             self.wd * mult, loss)

    # ------------------------------------------------------------ parameters
    def _pad_kernel(self, k, din, dinp):
        """reference [din+d, 4d] -> padded [dinp+dp, 4dp] (input rows, zero rows, hidden rows; gate blocks
        i,j,f,o kept apart)."""
        d, dp = self.d, self.dp
        out = torch.zeros(dinp + dp, 4 * dp, dtype=torch.float32)
        for g in range(4):
            out[:din, g * dp:g * dp + d] = k[:din, g * d:(g + 1) * d]
            out[dinp:dinp + d, g * dp:g * dp + d] = k[din:, g * d:(g + 1) * d]
        return out

    def _unpad_kernel(self, kp, din, dinp):
        d, dp = self.d, self.dp
        out = torch.zeros(din + d, 4 * d, dtype=torch.float32)
        for g in range(4):
            out[:din, g * d:(g + 1) * d] = kp[:din, g * dp:g * dp + d]
            out[din:, g * d:(g + 1) * d] = kp[dinp:dinp + d, g * dp:g * dp + d]
        return out

    def _din(self, name):
        return (self.text_in, self.text_in_p) if "utext" in name else (self.img_in, self.img_in_p)

    def _pad_blocks(self, v, nblk, blk, blkp):
        out = torch.zeros(nblk * blkp, dtype=torch.float32)
        for g in range(nblk):
            out[g * blkp:g * blkp + blk] = v[g * blk:(g + 1) * blk]
        return out

    def _pad_feat(self, v):
        """[F*w] feature vectors (att W, scorer W): each w-block = [fw d | bw d] -> [fw dp | bw dp]."""
        d, dp = self.d, self.dp
        return self._pad_blocks(v.reshape(-1), v.numel() // d, d, dp)

    def _unpad_feat(self, v):
        d, dp = self.d, self.dp
        n = v.numel() // dp
        return torch.cat([v[g * dp:g * dp + d] for g in range(n)])

    def _pad_feat2d(self, m):
        """[R*d, C*d] matrix whose rows AND columns are d-blocks (time_warp/WH/W) -> dp-blocks."""
        d, dp = self.d, self.dp
        R, C = m.shape[0] // d, m.shape[1] // d
        out = torch.zeros(R * dp, C * dp, dtype=torch.float32)
        for r in range(R):
            for c in range(C):
                out[r * dp:r * dp + d, c * dp:c * dp + d] = m[r * d:(r + 1) * d, c * d:(c + 1) * d]
        return out

    def _unpad_feat2d(self, m):
        d, dp = self.d, self.dp
        R, C = m.shape[0] // dp, m.shape[1] // dp
        return torch.cat([torch.cat([m[r * dp:r * dp + d, c * dp:c * dp + d] for c in range(C)], 1) for r in range(R)], 0)

    def _is_w2d(self, name):
        """linear weights whose rows AND columns are hidden-size blocks ([R*d, C*d], stored [R*dp, C*dp])"""
        return name == self.N_TW_WH_W

    def _is_bfeat(self, name):
        """bias vectors in the feature layout ([C*d] stored [C*dp])"""
        return name == self.N_TW_WH_B

    def set_weights(self, weights, flat=None):
        """weights: dict reference-name -> array in the REFERENCE's shapes
        (main.py:578-588 `weights.npz` layout).  `flat`: write into that buffer (an optimiser slot with the parameter
        store's layout) instead of the parameters."""
        for name, val in weights.items():
            if name not in self.params.specs:
                continue
            t = torch.as_tensor(np.asarray(val), dtype=torch.float32)
            if name in self._plain:
                pass
            elif name.endswith("basic_lstm_cell/kernel"):
                t = self._pad_kernel(t, *self._din(name))
            elif name.endswith("basic_lstm_cell/bias"):
                t = self._pad_blocks(t, 4, self.d, self.dp)
            elif self._is_w2d(name):
                t = self._pad_feat2d(t)
            elif name.endswith("/W") or self._is_bfeat(name):
                t = self._pad_feat(t)
            self.params.view(name, flat=flat).copy_(t.reshape(self.params.specs[name]).to(self.dev))

    def get_weights(self, grad=False, flat=None):
        """-> dict reference-name -> numpy array in the reference's shapes (`flat`: read that buffer instead)."""
        out = {}
        for name in self.params.specs:
            t = self.params.view(name, grad=grad, flat=flat).detach().cpu()
            if name in self._plain:
                t = t.reshape((1,) + tuple(t.shape)) if name == self.N_CONV_F else t
            elif name.endswith("basic_lstm_cell/kernel"):
                t = self._unpad_kernel(t, *self._din(name))
            elif name.endswith("basic_lstm_cell/bias"):
                t = torch.cat([t[g * self.dp:g * self.dp + self.d] for g in range(4)])
            elif self._is_w2d(name):
                t = self._unpad_feat2d(t)
            elif self._is_bfeat(name):
                t = self._unpad_feat(t)
            elif name.endswith("/W"):
                t = self._unpad_feat(t).reshape(-1, 1)
            out[name] = t.numpy()
        return out

    N_TW_WINDOW = "time_warp/time_warp_C/time_warp_window_t"            # model_v2.py:334

    def save_weights(self, weights_path):
        """main.py:578-588 (`--is_save_weights`): every trainable variable under its TF name ("<scope>/<name>:0",
        scope = "model_<modelname>", model_v2.py:15) in `weights.npz`, plus the `all.txt` name/shape listing."""
        import os
        os.makedirs(weights_path, exist_ok=True)
        out = {"%s/%s:0" % (self.scope, k): v for k, v in self.get_weights().items()}
        if self.use_time_warp and self.warp_type == 5:
            out["%s/%s:0" % (self.scope, self.N_TW_WINDOW)] = np.float32(self.window_t)
        out["%s/global_step:0" % self.scope] = np.int64(self.global_step)           # model_v2.py:366; not a trainable
        with open(os.path.join(weights_path, "all.txt"), "w") as f:
            for k, v in out.items():
                f.writelines("%s %s\n" % (k, str(tuple(int(x) for x in np.shape(v)))))
        np.savez(os.path.join(weights_path, "weights.npz"), **out)
        return os.path.join(weights_path, "weights.npz")

    def load_weights(self, path):
        """Inverse of save_weights: `path` is the weights directory or the .npz itself.  Keys may carry any
        "<scope>/" prefix and the ":0" suffix; a variable of this model that the file lacks is an error, like the
        reference's restore by variable name (main.py:640-665)."""
        import os
        if os.path.isdir(path):
            path = os.path.join(path, "weights.npz")
        if not os.path.exists(path):
            raise Exception("Model not exists")                         # main.py:665
        with np.load(path) as z:
            arrays = {key: z[key] for key in z.files}
        self._assign_named(arrays, "weights file %s" % path)

    OPTIMIZER_SLOTS = ("Adadelta", "Adadelta_1", "Adam", "Adam_1")       # tf.train.*Optimizer slot variable suffixes

    def load_tf_checkpoint(self, path, verify=True):
        """main.py:640-665 (`saver.restore(sess, ckpt.model_checkpoint_path)`): `path` is the reference's save directory
        (its `checkpoint` state file names the latest prefix), a checkpoint prefix or its .index file.  Restores every
        variable of this model by name plus global_step; returns the optimiser's slot variables found in the file,
        {slot suffix: {variable name: array}} (Trainer.restore_tf_checkpoint puts them back).  The file format is
        restated in tf_checkpoint.py (unpinned: no TensorFlow here to write a reference file)."""
        from .tf_checkpoint import read_checkpoint
        arrays = read_checkpoint(path, verify=verify)      # verify: every tensor's crc32c (about 70 MB/s; False skips it)
        slots, model_vars = {}, {}
        for key, val in arrays.items():
            base, _, last = key.rpartition("/")
            if last in self.OPTIMIZER_SLOTS and base:
                slots.setdefault(last, {})[base] = val
            elif last in ("beta1_power", "beta2_power"):
                slots.setdefault(last, {})[""] = val
            else:
                model_vars[key] = val
        self._assign_named(model_vars, "checkpoint %s" % path)
        return slots

    def _assign_named(self, arrays, what):
        """{variable name (any "<scope>/" prefix, optional ":0"): array} -> parameters, global_step, the warp window.
        Strict both ways: a variable this model lacks in `arrays`, or one of `arrays` it has no place for, is an error."""
        known = list(self.params.specs) + [self.N_TW_WINDOW, "global_step"]
        got, unmatched = {}, []
        for key, val in arrays.items():
            k = key[:-2] if key.endswith(":0") else key
            for name in known:
                if k == name or k.endswith("/" + name):
                    got[name] = val
                    break
            else:
                unmatched.append(key)
        missing = [n for n in self.params.specs if n not in got]
        if missing:
            raise KeyError("%s lacks %s" % (what, ", ".join(missing)))
        # a variable of the FILE that this model has no place for is an error too: silently dropping e.g. the
        # .../bw/basic_lstm_cell/{kernel,bias} of a TF-1.0-style checkpoint into a share_fw_bw=True model would load a
        # different network (switches that legitimately remove variables: share_fw_bw, use_question_att, use_time_warp, ...)
        if unmatched:
            hint = " (the file has separate backward-direction cells: build the model with share_fw_bw=False)" \
                if any("/bw/basic_lstm_cell/" in k for k in unmatched) and self.share_fw_bw else ""
            raise KeyError("%s holds variables this model does not have: %s%s" % (what, ", ".join(unmatched), hint))
        if "global_step" in got:
            self.global_step = int(got.pop("global_step"))
        if self.N_TW_WINDOW in got:
            self.window_t = float(got.pop(self.N_TW_WINDOW))
            self._layouts.clear()                                       # the window is baked into the warp descriptors
        self.set_weights(got)

    def init_parameters(self, seed=42):
        """Reference initialisers: linear W ~ truncated_normal(0.1), b = 0
        (model_v2.py:88-89); LSTM kernels Glorot-uniform, zero bias [TF default]."""
        from .synth import _glorot, _trunc_normal
        g = torch.Generator().manual_seed(seed)
        wts = {}
        for name, shape in self.params.specs.items():
            if name in self._plain:
                if name == self.N_WORD_EMB:         # main.py:308: N(0, I) rows for the non-GloVe words
                    wts[name] = torch.randn(shape, generator=g)
                elif name == self.N_IMGT_W:          # linear: truncated_normal(0.1), b = 0
                    wts[name] = _trunc_normal(g, shape)
                elif name == self.N_IMGT_B:
                    wts[name] = torch.zeros(shape)
                else:                               # tf.get_variable default: Glorot uniform
                    fan = (shape[0] * shape[1], shape[2]) if len(shape) == 3 else (shape[0], shape[-1])
                    lim = (6.0 / (fan[0] + fan[1])) ** 0.5
                    wts[name] = (torch.rand(shape, generator=g) * 2 - 1) * lim
            elif name.endswith("basic_lstm_cell/kernel"):
                din = self.text_in if "utext" in name else self.img_in
                wts[name] = _glorot(g, din + self.d, 4 * self.d)
            elif self._is_w2d(name):
                wts[name] = _trunc_normal(g, (shape[0] // self.dp * self.d, shape[1] // self.dp * self.d))
            elif name.endswith("/W"):
                wts[name] = _trunc_normal(g, (shape[0] // self.dp * self.d, 1))
        self.set_weights(wts)

    def _oracle_key_map(self):
        """oracle short key -> reference variable name"""
        return {"text_kernel": self.N_TEXT_K % "fw", "text_bias": self.N_TEXT_B % "fw",
                "text_kernel_bw": self.N_TEXT_K % "bw", "text_bias_bw": self.N_TEXT_B % "bw",
                "image_kernel": self.N_IMG_K % "fw", "image_bias": self.N_IMG_B % "fw",
                "image_kernel_bw": self.N_IMG_K % "bw", "image_bias_bw": self.N_IMG_B % "bw",
                "att_W": self.N_ATT_W, "att_b": self.N_ATT_B, "qatt_W": self.N_QATT_W, "qatt_b": self.N_QATT_B,
                "out_W": self.N_OUT_W, "out_b": self.N_OUT_B, "WH_W": self.N_TW_WH_W, "WH_b": self.N_TW_WH_B,
                "WC_W": self.N_TW_WC_W, "WC_b": self.N_TW_WC_B, "word_emb": self.N_WORD_EMB, "char_emb": self.N_CHAR_EMB,
                "conv_filter": self.N_CONV_F, "conv_bias": self.N_CONV_B, "img_W": self.N_IMGT_W, "img_b": self.N_IMGT_B}

    def set_oracle_params(self, p):
        """Parameters in the oracle's short-key format (fvta_memexqa_amd.synth.make_params)."""
        m = self._oracle_key_map()
        if "existing_emb_mat" in p:
            self.set_existing_emb(p["existing_emb_mat"])
        if "window_t" in p:
            self.window_t = float(p["window_t"])
        self.set_weights({m[k]: v for k, v in p.items() if k in m})

    def _oracle_names(self):
        return {v: k for k, v in self._oracle_key_map().items()}

    def get_oracle_grads(self):
        m = self._oracle_names()
        return {m[k]: v for k, v in self.get_weights(grad=True).items()}

    def get_oracle_params(self):
        """current parameters in the oracle's short-key format (torch fp32, CPU), incl. the frozen word table"""
        m = self._oracle_names()
        out = {m[k]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in self.get_weights().items()}
        if self.token_mode:
            out["existing_emb_mat"] = self.existing_emb_mat.cpu()
        return out

    def unpad_w(self, t):
        """[..., 2*d_pad] kernel layout (fw half | bw half, each zero padded) -> the reference's [..., 2d]"""
        if self.dp == self.d:
            return t
        return torch.cat([t[..., :self.d], t[..., self.dp:self.dp + self.d]], -1)

    def set_existing_emb(self, mat):
        """the frozen pre-trained word vectors the reference feeds as `existing_emb_mat` (model_v2.py:470, 590):
        word ids >= word_vocab_size index into it."""
        self.existing_emb_mat = torch.as_tensor(np.asarray(mat), dtype=torch.float32).to(self.dev).contiguous()

    # ---------------------------------------------------------------- layout
    def _layout(self, shapes, training):
        key = (tuple(shapes["ctx"]), shapes["q"], shapes["choices"], bool(training), int(self.existing_emb_mat.shape[0])
               if self.token_mode else -1)
        if key in self._layouts:
            self._layouts[key] = self._layouts.pop(key)      # most recently used last
            return self._layouts[key]
        while len(self._layouts) >= self.max_layouts:         # real batches change shape (per-batch maxima): bounded cache
            old = self._layouts.pop(next(iter(self._layouts)))
            if self._hall_layout is old:                      # (an evicted layout's arena / saved buffers go with it)
                self._hall_layout = None
        L = _Layout()
        dev, wp, dp = self.dev, self.wp, self.dp
        N, JQ = shapes["q"]
        _, C, JA = shapes["choices"]
        ctx = shapes["ctx"]  # tuple of (cell, dims) with dims = (N,M,J) or (N,M,JI,J)
        K = len(ctx)
        M = ctx[0][1][1]
        per_album = [int(np.prod(dims[2:])) for _, dims in ctx]
        JMAX = max(per_album)                                   # model_v2.py:869
        T = M * JMAX
        L.N, L.K, L.M, L.JMAX, L.T, L.JQ, L.C, L.JA = N, K, M, JMAX, T, JQ, C, JA
        seq_rows = self._plan_arena(L, ctx, training)
        L.shadow = self._shadow_covers(L)
        L.hall_fresh = True
        if L.shadow:
            L.shadow_tab = torch.zeros(2, L.row_hq, dtype=torch.int64, device=dev)
            L.zero_half = torch.zeros(dp, dtype=torch.bfloat16, device=dev)     # the rows no encoder writes
        # ---- sequence groups (text cell / image cell)
        groups = {"text": [], "image": []}
        groups["text"].append(dict(name="q", count=N, J=JQ, rows=L.row_hq + torch.arange(N) * JQ))
        groups["text"].append(dict(name="choices", count=N * C, J=JA, rows=L.row_hch + torch.arange(N * C) * JA))
        L.ctx_slots = []
        for k, (cell, dims) in enumerate(ctx):
            J = dims[-1]
            cnt = int(np.prod(dims[:-1]))
            groups[cell].append(dict(name="ctx%d" % k, count=cnt, J=J, rows=seq_rows(k, dims)))
            L.ctx_slots.append((cell, len(groups[cell]) - 1, dims))
        L.groups = {}
        for cell, segs in groups.items():
            if not segs:
                continue
            din = self.text_in_p if cell == "text" else self.img_in_p
            G = SimpleNamespace(segs=segs, din=din)
            B = sum(s["count"] for s in segs)
            Jmax = max(s["J"] for s in segs)
            x_off, out_off, seq_J, pos = [], [], [], 0
            s0 = 0
            for s in segs:
                s["x_elem0"] = pos
                s["s0"] = s0
                x_off.append(pos + torch.arange(s["count"], dtype=torch.int64) * s["J"] * din)
                out_off.append(s["rows"].to(torch.int64) * wp)
                seq_J.append(torch.full((s["count"],), s["J"], dtype=torch.int32))
                pos += s["count"] * s["J"] * din
                s0 += s["count"]
            G.B, G.J = B, Jmax
            G.x = torch.zeros(pos, dtype=torch.float32, device=dev)
            G.dx = None
            # input dropout: each direction of the cell runs on its own dropped copy of G.x (x2 = [fw | bw]); their
            # gradients come back as dx2 = [fw | bw] and are folded through the same masks (ops.dropout_pair_*)
            G.dropout = bool(training) and self.keep_prob < 1.0
            G.x2 = torch.zeros(2 * pos, dtype=torch.float32, device=dev) if G.dropout else None
            G.dx2 = None
            G.drop_seed = 0
            if self.token_mode:
                # one row per sequence position (padding positions included, as the reference embeds them too)
                t0 = 0
                offs = []
                for s in segs:
                    s["tok0"] = t0
                    n = s["count"] * s["J"]
                    offs.append(s["x_elem0"] + torch.arange(n, dtype=torch.int64) * din)
                    t0 += n
                G.ntok = t0
                G.tok_off = torch.cat(offs).to(dev)
                if cell == "text":
                    G.word_ids = torch.zeros(t0, dtype=torch.int32, device=dev)
                    G.char_ids = torch.zeros(t0, self.Wc, dtype=torch.int32, device=dev) if self.cwdim else None
                    G.embed = ops.TokenEmbed(t0, self.Wc, self.cdim, self.cwdim, self.wdim, self.VW,
                                             self.VW + int(self.existing_emb_mat.shape[0]), max(self.VC, 1))
                else:
                    G.pidx = torch.zeros(t0, dtype=torch.int32, device=dev)
                    G.embed = ops.ImageTrans(t0, self.idim, self.tdim, self.add_tanh)
            G.lens = torch.zeros(B, dtype=torch.int32, device=dev)
            G.op = ops.BiLstm(B, Jmax, din, dp, torch.cat(x_off), torch.cat(out_off), torch.cat(seq_J), wp,
                              share_fw_bw=self.share_fw_bw, precision=self.precision, training=training,
                              prof_tag=0 if cell == "text" else 1, x_bw_delta=pos if G.dropout else 0,
                              dx_overwrite=True,   # backward() writes dx: no zero fill per step
                              out_pads_persist=True,   # L.arena is written by this op only
                              out_skip=L.row_hq * wp if L.shadow else 0)
            L.groups[cell] = G
        L.q_mask = torch.zeros(N, JQ, dtype=torch.uint8, device=dev)
        self._build_attention(L, training)
        L.y = torch.zeros(N, C, dtype=torch.uint8, device=dev)
        self._layouts[key] = L
        return L

    SHADOW_OK = True     # (model.py's attentions read the fp32 rows)

    def _shadow_covers(self, L):
        """the shapes fvta_attn_fwd_shadow / fvta_attn_bwd_shadow take (include/fvta_hip.h)"""
        # (with the time warp: fvta_timewarp_fwd_shadow / _bwd_shadow warp the bf16 rows; time_warp_att keeps the fp32 rows)
        return (self.SHADOW_OK and self.shadow_rows and self.precision == BF16 and
                (not self.use_time_warp or (not self.use_time_warp_att and L.K <= 8)) and
                self.simi in (1, 2, 3) and self.wp in (512, 1024) and L.JQ <= 32)

    @property
    def hall(self):
        """the context tensor of the last forward (vis output, model_v2.py:914): under shadow rows it is filled in here"""
        L = self._hall_layout
        if L is None:
            return None
        ev = getattr(L, "fwd_done", None)
        if ev is not None:       # ordered behind the forward that produced the rows, whatever stream the reader runs on
            torch.cuda.current_stream().wait_event(ev)
        if L.shadow and not L.hall_fresh:
            # (valid until the next forward on this layout overwrites the shadow rows: hall_fresh is reset there)
            ops.rows_from_shadow(L.shadow_tab, L.row_hq, self.dp, self.wp, L.arena)
            L.hall_fresh = True
        return L.hall

    @property
    def warp_h(self):
        """the warped context tensor of the last forward (vis output, model_v2.py:1009); under shadow rows the bf16 rows are
        converted into an fp32 buffer here, on demand"""
        if getattr(self, "_warp_h", None) is not None:
            return self._warp_h
        L = getattr(self, "_warp_layout", None)
        if L is None:
            return None
        ev = getattr(L, "fwd_done", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        if getattr(L, "warp_f32", None) is None:
            L.warp_f32 = torch.zeros(L.N, L.K, L.T, self.wp, dtype=torch.float32, device=self.dev)
        ops.rows_from_shadow(L.warp_tab, L.row_hq, self.dp, self.wp, L.warp_f32)
        return L.warp_f32

    def _plan_arena(self, L, ctx, training):
        """Rows of the encoder output arena: [hall N*K*T | hq N*JQ | hchoices N*C*JA], the context tensor of
        model_v2.py:863-914 (every stream padded to JMAX rows per album, stacked on K).  Returns seq_rows(k, dims): the
        first arena row of every sequence of context stream k."""
        dev, wp = self.dev, self.wp
        N, K, M, JMAX, T, JQ, C, JA = L.N, L.K, L.M, L.JMAX, L.T, L.JQ, L.C, L.JA
        L.row_hall, L.row_hq = 0, N * K * T
        L.row_hch = L.row_hq + N * JQ
        L.rows = L.row_hch + N * C * JA
        L.arena = torch.zeros(L.rows, wp, dtype=torch.float32, device=dev)
        L.d_arena = torch.zeros(L.rows, wp, dtype=torch.float32, device=dev) if training else None
        L.hall = L.arena[:L.row_hq].view(N, K, T, wp)
        L.hq = L.arena[L.row_hq:L.row_hch].view(N, JQ, wp)
        L.hch = L.arena[L.row_hch:].view(N, C, JA, wp)
        L.hall_mask = torch.zeros(N, K, M, JMAX, dtype=torch.uint8, device=dev)

        def seq_rows(k, dims):
            n = torch.arange(N).view(N, 1, 1)
            m = torch.arange(M).view(1, M, 1)
            if len(dims) == 3:
                ji = torch.zeros(1, 1, 1, dtype=torch.int64)
                step = 0
            else:
                ji = torch.arange(dims[2]).view(1, 1, -1)
                step = dims[3]
            return (((n * K + k) * M + m) * JMAX + ji * step).reshape(-1)
        return seq_rows

    def _build_attention(self, L, training):
        """kernel handles and buffers of the block between the encoders and the scorer"""
        dev, wp = self.dev, self.wp
        N, K, T, JQ, C = L.N, L.K, L.T, L.JQ, L.C
        L.ones_mask = torch.ones(N, 1, dtype=torch.uint8, device=dev)
        L.att = ops.FocalAttention(N, K, T, JQ, wp, self.simi, self.add_tanh)
        L.qatt = ops.FocalAttention(N, 1, JQ, 1, wp, self.simi, self.add_tanh) if self.use_question_att else None
        L.lq = torch.zeros(N, wp, dtype=torch.float32, device=dev)
        L.lch = torch.zeros(N, C, wp, dtype=torch.float32, device=dev)
        if self.use_time_warp:
            L.tw = ops.TimeWarp(N, K, T, wp, self.warp_type, self.window_t)
            if L.shadow:     # the warped rows as bf16, read by the attention through a static table of their addresses
                L.warp_b = torch.zeros(N * K * T, wp, dtype=torch.bfloat16, device=dev)
                base = L.warp_b.data_ptr() + torch.arange(N * K * T, dtype=torch.int64) * (wp * 2)
                L.warp_tab = torch.stack([base, base + self.dp * 2]).to(dev).contiguous()
                L.warp_h = None          # (the fp32 vis tensor is filled in on demand: Model.warp_h)
            else:
                L.warp_h = torch.zeros(N, K, T, wp, dtype=torch.float32, device=dev)
            L.d_warp = torch.zeros(N, K, T, wp, dtype=torch.float32, device=dev) if training else None
            L.d_lq = torch.zeros(N, wp, dtype=torch.float32, device=dev) if training else None
            L.d_tscale = torch.zeros(N, T, dtype=torch.float32, device=dev) if (training and self.use_time_warp_att) else None

    def _host_ctx_mask(self, L):
        return np.zeros((L.N, L.K, L.M, L.JMAX), np.uint8)

    def _put_ctx_mask(self, L, buf, k, m):
        """stream k's mask m [N,M,rows per album] into the context mask (`buf`: L.hall_mask or its host staging array)"""
        buf[:, k, :, :m.shape[2]] = m

    def seg_x(self, L, cell, si):
        """arena view of a segment's encoder input, shaped [count, J, in]"""
        G = L.groups[cell]
        s = G.segs[si]
        n = s["count"] * s["J"] * G.din
        return G.x[s["x_elem0"]:s["x_elem0"] + n].view(s["count"], s["J"], G.din)

    def get_input_grads(self, L):
        """d loss / d encoder inputs after backward(L, need_dx=True): one [count, J, in] view per stream, context
        streams in their order, then the question, then the choices (the gradients the embedding front-end consumes)"""
        def seg(cell, si):
            G = L.groups[cell]
            s = G.segs[si]
            n = s["count"] * s["J"] * G.din
            din = self.text_in if cell == "text" else self.img_in
            return G.dx[s["x_elem0"]:s["x_elem0"] + n].view(s["count"], s["J"], G.din)[:, :, :din]
        return [seg(cell, si) for cell, si, _ in L.ctx_slots] + [seg("text", 0), seg("text", 1)]

    @staticmethod
    def _lead(st):
        """leading (sequence) shape of a stream in either entry form: encoder inputs `x`, word ids, photo indices"""
        if "x" in st:
            return tuple(st["x"].shape[:-1])
        return tuple(st["ids"].shape) if "ids" in st else tuple(st["pis"].shape)

    @classmethod
    def shapes_of(cls, inputs):
        return dict(ctx=tuple((st.get("cell", "text"), cls._lead(st)) for st in inputs["ctx"]),
                    q=cls._lead(inputs["q"]), choices=cls._lead(inputs["choices"]))

    def get_feed_dict(self, batch, is_train=False):
        """model_v2.py:1099-1565.  `batch` is a utils.Dataset mini-batch (what Dataset.get_batches yields): returns the
        reference's feed arrays keyed by placeholder name (`at`, `at_c`, `at_mask`, ..., `y`, `image_emb_mat`,
        `existing_emb_mat`), built by feed.build_feed_dict.  A dict that already holds encoder inputs or index arrays
        (the oracle `inputs` format) passes through."""
        if isinstance(batch, dict) and ("ctx" in batch or "at" in batch):
            return batch
        if not (hasattr(batch, "data") and hasattr(batch, "shared")):
            raise TypeError("get_feed_dict wants a utils.Dataset mini-batch or an inputs / feed dict")
        from .feed import build_feed_dict
        feed, self._vocab_memo = build_feed_dict(self.config, batch, is_train, self.num_choice,
                                                 getattr(self, "_vocab_memo", None))
        return feed

    def _no_photo(self):
        """--no_photo (main.py:42): model_v2.py:864, 904 leave the photo stream out of the context tensor"""
        return bool(_cfg(self.config, "no_photo", False))

    def inputs_from_feed(self, feed):
        """feed dict (placeholder names) -> the `inputs` tree load_inputs takes; context streams in the reference's
        stacking order at, ad, when, where, pts, pis (model_v2.py:905-912; `no_photo` drops pis)."""
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))

        def text(n):
            return dict(ids=t(feed[n]), chars=t(feed[n + "_c"]), mask=t(feed[n + "_mask"]), cell="text")

        ctx = [text(n) for n in ("at", "ad", "when", "where", "pts")]
        if not self._no_photo():
            ctx.append(dict(pis=t(feed["pis"]), mask=t(feed["pis_mask"]), cell="image"))
        emb = feed.get("existing_emb_mat")
        if emb is not None and emb is not getattr(self, "_fed_emb", None):     # fed every step, uploaded when it changes
            self.set_existing_emb(emb)
            self._fed_emb = emb
        return dict(ctx=ctx, q=text("q"), choices=text("choices"), y=t(feed["y"]) if feed.get("y") is not None else None,
                    image_emb_mat=feed["image_emb_mat"])

    def load_inputs(self, inputs, training=False):
        """Copy encoder inputs (oracle `inputs` format, any device) into the arenas.
        Host-side plumbing: masks -> lengths (model_v2.py:667-678) and the mask
        pad/stack of model_v2.py:890-912."""
        if "at" in inputs:
            inputs = self.inputs_from_feed(inputs)
        L = self._layout(self.shapes_of(inputs), training)
        dev = self.dev

        L.token = "x" not in inputs["q"]
        if L.token and not self.token_mode:
            raise ValueError("token-id inputs need a model built with word_vocab_size & co. in its config")
        if L.token:
            L.image_emb_mat = torch.as_tensor(np.asarray(inputs["image_emb_mat"]) if not torch.is_tensor(
                inputs["image_emb_mat"]) else inputs["image_emb_mat"]).to(dev, torch.float32).contiguous()

        # Host-resident index inputs (the feed path: NumPy arrays / CPU tensors) are packed on the host and go up as ONE
        # copy per device buffer -- word ids, char ids, photo indices, lengths, context mask -- instead of one small copy
        # per stream and field (~45 per step, which at README batch sizes cost as much host time as the device step).
        host = L.token and all(not (torch.is_tensor(v) and v.is_cuda)
                               for st in [inputs["q"], inputs["choices"]] + list(inputs["ctx"]) for v in st.values()
                               if torch.is_tensor(v))
        if host:
            npy = lambda v: v.numpy() if torch.is_tensor(v) else np.asarray(v)
            parts = {c: dict(ids=[None] * len(G.segs), chars=[None] * len(G.segs), lens=[None] * len(G.segs))
                     for c, G in L.groups.items()}
            hall_mask = self._host_ctx_mask(L)

            def stage(cell, si, st):
                mask = npy(st["mask"])
                pp = parts[cell]
                pp["lens"][si] = mask.reshape(-1, mask.shape[-1]).sum(1).astype(np.int32)
                if cell == "text":
                    pp["ids"][si] = npy(st["ids"]).reshape(-1).astype(np.int32, copy=False)
                    if L.groups[cell].char_ids is not None:
                        pp["chars"][si] = npy(st["chars"]).reshape(-1, self.Wc).astype(np.int32, copy=False)
                else:
                    pp["ids"][si] = npy(st["pis"]).reshape(-1).astype(np.int32, copy=False)
                return mask

            stage("text", 0, inputs["q"])
            stage("text", 1, inputs["choices"])
            for k, st in enumerate(inputs["ctx"]):
                cell, si, dims = L.ctx_slots[k]
                self._put_ctx_mask(L, hall_mask, k, stage(cell, si, st).reshape(L.N, L.M, -1))
            up = lambda dst, arr: dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)))
            for cell, G in L.groups.items():
                pp = parts[cell]
                up(G.lens, np.concatenate(pp["lens"]))
                G.op.set_active_hint(np.concatenate(pp["lens"]))     # the host knows the lengths: see BiLstm.set_active_hint
                if cell == "text":
                    up(G.word_ids, np.concatenate(pp["ids"]))
                    if G.char_ids is not None:
                        up(G.char_ids, np.concatenate(pp["chars"]))
                else:
                    up(G.pidx, np.concatenate(pp["ids"]))
            up(L.hall_mask, hall_mask)
            up(L.q_mask, npy(inputs["q"]["mask"]).astype(np.uint8))
        else:
            def put(cell, si, st):
                G = L.groups[cell]
                s = G.segs[si]
                mask = st["mask"]
                if "x" in st:
                    x = st["x"]
                    self.seg_x(L, cell, si)[:, :, :x.shape[-1]].copy_(x.reshape(-1, x.shape[-2], x.shape[-1]).to(dev, torch.float32))
                else:
                    n = s["count"] * s["J"]
                    if cell == "text":
                        G.word_ids[s["tok0"]:s["tok0"] + n] = st["ids"].reshape(-1).to(dev, torch.int32)
                        if G.char_ids is not None:
                            G.char_ids[s["tok0"]:s["tok0"] + n] = st["chars"].reshape(n, -1).to(dev, torch.int32)
                    else:
                        G.pidx[s["tok0"]:s["tok0"] + n] = st["pis"].reshape(-1).to(dev, torch.int32)
                G.lens[s["s0"]:s["s0"] + s["count"]] = mask.reshape(-1, mask.shape[-1]).to(dev).sum(1).to(torch.int32)
                if not mask.is_cuda:      # masks that live on the host: their row sums are the lengths hint (no device sync)
                    host_lens.setdefault(cell, {})[si] = mask.reshape(-1, mask.shape[-1]).sum(1).numpy()

            host_lens = {}
            put("text", 0, inputs["q"])
            put("text", 1, inputs["choices"])
            L.q_mask.copy_(inputs["q"]["mask"].to(dev, torch.uint8))
            L.hall_mask.zero_()
            for k, st in enumerate(inputs["ctx"]):
                cell, si, dims = L.ctx_slots[k]
                put(cell, si, st)
                self._put_ctx_mask(L, L.hall_mask, k, st["mask"].to(dev, torch.uint8).reshape(L.N, L.M, -1))
            for cell, G in L.groups.items():
                got = host_lens.get(cell, {})
                G.op.set_active_hint(np.concatenate([got[i] for i in range(len(G.segs))]) if len(got) == len(G.segs) else None)
        if inputs.get("y") is not None:
            y = inputs["y"]
            L.y.copy_((y if torch.is_tensor(y) else torch.from_numpy(np.ascontiguousarray(y))).to(torch.uint8))
            L.has_y = True
        else:
            L.has_y = False
        return L

    # --------------------------------------------------------------- forward
    def _cell_params(self, cell, grad=False):
        kn, bn = (self.N_TEXT_K, self.N_TEXT_B) if cell == "text" else (self.N_IMG_K, self.N_IMG_B)
        v = lambda n: self.params.view(n, grad)
        if self.share_fw_bw:
            return v(kn % "fw"), v(bn % "fw"), None, None
        return v(kn % "fw"), v(bn % "fw"), v(kn % "bw"), v(bn % "bw")

    def forward(self, L, want_logits=False):
        """model_v2.py:649-1096 on the loaded batch.  Returns yp (device tensor)."""
        P = self.params
        main = torch.cuda.current_stream()
        if getattr(L, "token", False):                                      # model_v2.py:524-645
            T_, I_ = L.groups["text"], L.groups.get("image")
            cw = self.cwdim
            if cw:                                                          # conv1d's dropout (model_v2.py:58-62), training only
                if T_.dropout:
                    self._dropout_calls += 1
                    T_.char_drop_seed = ((self.dropout_seed * 0x9E3779B1 + self._dropout_calls * 2 + 0x5851F42D) ^ self._dropout_rank_salt) & (2 ** 64 - 1)
                    T_.embed.set_dropout(self.keep_prob, T_.char_drop_seed)
                else:
                    T_.embed.set_dropout(1.0, 0)
            T_.embed.forward(T_.word_ids, T_.char_ids, T_.tok_off, P.view(self.N_WORD_EMB), self.existing_emb_mat,
                             P.view(self.N_CHAR_EMB) if cw else None, P.view(self.N_CONV_F) if cw else None,
                             P.view(self.N_CONV_B) if cw else None, T_.x)
            if I_ is not None:
                it = self.use_image_trans
                I_.embed.forward(I_.pidx, I_.tok_off, L.image_emb_mat, P.view(self.N_IMGT_W) if it else None,
                                 P.view(self.N_IMGT_B) if it else None, I_.x)
        # the photo cell (few rows, one short latency-bound launch per photo) runs beside the text cell on a side
        # HIP stream: the two write disjoint rows of the arena and meet again before the attention.  It is
        # enqueued FIRST -- a side stream that waits for main after the text launches are queued runs after them.
        if L.shadow:
            L.shadow_tab.fill_(L.zero_half.data_ptr())
            L.hall_fresh = False
        for cell, G in sorted(L.groups.items(), key=lambda kv: kv[0] != "image"):
            side = self._side if (cell == "image" and "text" in L.groups and not self.serial_photo_forward) else None
            if side is not None:
                side.wait_stream(main)
            with torch.cuda.stream(side if side is not None else main):
                G.op.make_plan(G.lens)
                kf, bf, kb, bb = self._cell_params(cell)
                if G.dropout:                                              # model_v2.py:657-661
                    self._dropout_calls += 1
                    G.drop_seed = ((self.dropout_seed * 0x9E3779B1 + self._dropout_calls * 2 + (cell == "image")) ^ self._dropout_rank_salt) & (2 ** 64 - 1)
                    ops.dropout_pair_fwd(G.x, G.x2, self.keep_prob, G.drop_seed)
                G.op.forward(G.x2 if G.dropout else G.x, L.arena, kf, bf, kb, bb)   # encoders + context tensor
                if L.shadow:
                    G.op.shadow_rows(L.shadow_tab, L.row_hq)
        main.wait_stream(self._side)
        att, qatt = self._attend(L, want_logits)
        L.logits, L.yp, L.loss_t = ops.scorer_ce_fwd(L.gq, L.g1, L.lch, P.view(self.N_OUT_W), P.view(self.N_OUT_B),
                                                     L.y if L.has_y else None, self.use_eu_output, self.scorer_tanh)
        if self.wd and L.has_y:                                              # :1094-1095: loss = add_n("losses")
            self._apply_wd(False, L.loss_t)
        self.logits, self.yp, self.loss = L.logits, L.yp, L.loss_t
        self._hall_layout = L
        if getattr(L, "fwd_done", None) is None:
            L.fwd_done = torch.cuda.Event()
        L.fwd_done.record(main)       # Model.hall waits for it (a reader on another stream)
        if want_logits:
            self.att_logits, self.q_att_logits = att, qatt
        return L.yp

    def _attend(self, L, want_logits):
        """The block between the encoders and the scorer (model_v2.py:953-1050): time warp, attention_3d, question
        attention.  Sets L.g1, L.gq, L.lch (= gchoices); returns the two logit tensors (None unless wanted)."""
        P = self.params
        T = L.groups["text"]
        T.op.last_state(L.arena, T.segs[1]["s0"], T.segs[1]["count"], L.lch)   # lchoices :807-812
        W = P.view(self.N_ATT_W) if self.simi != 4 else None
        b = P.view(self.N_ATT_B) if self.simi != 4 else None
        ctx = L.hall
        tab = L.shadow_tab if L.shadow else None       # the rows the attention reads, as an address table (shadow rows)
        if self.use_time_warp:                                              # :953-1009, WQ = lq (:970)
            T.op.last_state(L.arena, T.segs[0]["s0"], T.segs[0]["count"], L.lq)
            tw_p = (P.view(self.N_TW_WH_W), P.view(self.N_TW_WH_B), P.view(self.N_TW_WC_W), P.view(self.N_TW_WC_B))
            if L.shadow:     # bf16 rows in, warped bf16 rows out: the attention reads THOSE through their own table
                L.tw.forward_shadow(L.shadow_tab, L.lq, *tw_p, L.warp_b)
                tab, ctx = L.warp_tab, None
                self._warp_h, self._warp_layout = None, L
            else:
                L.tw.forward(L.hall, L.lq, *tw_p, L.warp_h)
                ctx = self._warp_h = L.warp_h
            self.C = L.tw.c                                                 # c[n,t]; the reference's C[n,t,t'] = c[n,t]
        L.ctx = ctx
        # :1020; time_warp_att: the softmax over t runs on amax * sum_t' C[n,t,t'] = amax * c[n,t] cnt(t) (:269-275)
        L.tscale = L.tw.scale if self.use_time_warp_att else None
        if L.shadow and not want_logits:
            L.g1, att = L.att.forward_shadow(tab, L.hq, L.hall_mask.view(L.N, L.K, L.T), L.q_mask, W, b), None
        else:
            if L.shadow:     # the full logit tensor is an inspection output of the fp32-row kernel: fill the rows in
                self._hall_layout = L
                ctx = self.warp_h if self.use_time_warp else self.hall
            L.g1, att = L.att.forward(ctx, L.hq, L.hall_mask.view(L.N, L.K, L.T), L.q_mask, W, b, want_logits, tscale=L.tscale)
        if self.use_question_att:                                           # :1044
            Wq = P.view(self.N_QATT_W) if self.simi != 4 else None
            bq = P.view(self.N_QATT_B) if self.simi != 4 else None
            L.gq, qatt = L.qatt.forward(L.hq, L.g1.view(L.N, 1, self.wp), L.q_mask, L.ones_mask, Wq, bq, want_logits)
        else:
            T.op.last_state(L.arena, T.segs[0]["s0"], T.segs[0]["count"], L.lq)  # lq :697
            L.gq, qatt = L.lq, None
        return att, qatt

    # -------------------------------------------------------------- backward
    def backward(self, L, loss_scale=1.0, need_dx=False):
        """Gradient of the mean cross-entropy into params.grad (accumulated)."""
        P = self.params
        dgq, dg1, dgch = ops.scorer_ce_bwd(L.gq, L.g1, L.lch, P.view(self.N_OUT_W), P.view(self.N_OUT_B), L.y, L.logits,
                                           L.yp, loss_scale, P.view(self.N_OUT_W, True), P.view(self.N_OUT_B, True),
                                           self.use_eu_output, self.scorer_tanh, self.tf_xent_grad)
        self._attend_bwd(L, dgq, dg1, dgch)
        L.dg1 = dg1        # d loss / d g1 as the focal attention's backward saw it (scorer + question attention); inspection only
        main = torch.cuda.current_stream()
        token = getattr(L, "token", False)
        need_dx = need_dx or token          # the embedding parameters are trained through the encoder inputs
        for cell, G in sorted(L.groups.items(), key=lambda kv: kv[0] != "image"):   # side-stream cell first, see forward
            kf, bf, kb, bb = self._cell_params(cell)
            dkf, dbf, dkb, dbb = self._cell_params(cell, grad=True)
            if need_dx and G.dx is None:
                G.dx = torch.zeros_like(G.x)
            side = self._side if (cell == "image" and "text" in L.groups) else None
            if side is not None:
                side.wait_stream(main)      # d_arena is complete
            with torch.cuda.stream(side if side is not None else main):
                # (dx / dx2 are WRITTEN by the op -- desc.dx_overwrite, zeros at padded positions -- not added to)
                if G.dropout:
                    if need_dx and G.dx2 is None:
                        G.dx2 = torch.zeros_like(G.x2)
                    G.op.backward(G.x2, L.arena, L.d_arena, kf, kb, G.dx2 if need_dx else None, dkf, dbf, dkb, dbb)
                    if need_dx:
                        ops.dropout_pair_bwd(G.dx2, G.dx, self.keep_prob, G.drop_seed)
                else:
                    G.op.backward(G.x, L.arena, L.d_arena, kf, kb, G.dx if need_dx else None, dkf, dbf, dkb, dbb)
                if side is not None and not self.wd and self.early_allreduce:
                    # data parallelism, `early_allreduce=True` only: everything in [0, early_numel) of the flat gradient --
                    # scorer, attention(s), time warp, photo cell -- is final in THIS stream's order now; start its
                    # all-reduce on RCCL's stream while the text cell's recurrence (enqueued next, on the main stream)
                    # runs.  OFF by default: measured with one RCCL rank (tools/r04_rccl_probe.py), a collective in flight
                    # beside the recurrence costs the step 0.7 ms (14.0 vs 13.3 ms) -- more than the whole 12 MB bucket
                    # costs on the wire -- while ONE all-reduce of the whole bucket after the backward costs nothing
                    # measurable (13.33 vs 13.32 ms with the collectives stubbed out)
                    from . import dist
                    self.early_work = dist.allreduce_async(P.grad[:P.early_numel])
        main.wait_stream(self._side)            # both cells' gradients are in params.grad
        if self.wd:
            self._apply_wd(True, None)
        if token:
            T_, I_ = L.groups["text"], L.groups.get("image")
            cw = self.cwdim
            g = lambda n: P.view(n, True)
            T_.embed.backward(T_.word_ids, T_.char_ids, T_.tok_off, P.view(self.N_CHAR_EMB) if cw else None,
                              P.view(self.N_CONV_F) if cw else None, T_.dx, g(self.N_WORD_EMB),
                              g(self.N_CHAR_EMB) if cw else None, g(self.N_CONV_F) if cw else None,
                              g(self.N_CONV_B) if cw else None)
            if I_ is not None and self.use_image_trans:
                I_.embed.backward(I_.pidx, I_.tok_off, L.image_emb_mat, I_.x, I_.dx, g(self.N_IMGT_W), g(self.N_IMGT_B))

    def _attend_bwd(self, L, dgq, dg1, dgch):
        """Backward of _attend: the scorer's three input gradients into every row of L.d_arena and the attention
        block's parameter gradients."""
        P = self.params
        cos = self.simi == 4      # cosine similarity has no att_logits/{W,b}
        aW, ab = (None, None) if cos else (P.view(self.N_ATT_W), P.view(self.N_ATT_B))
        daW, dab = (None, None) if cos else (P.view(self.N_ATT_W, True), P.view(self.N_ATT_B, True))
        if self.use_question_att and not cos:
            qW, qb, dqW, dqb = P.view(self.N_QATT_W), P.view(self.N_QATT_B), P.view(self.N_QATT_W, True), P.view(self.N_QATT_B, True)
        else:
            qW = qb = dqW = dqb = None
        L.d_arena[L.row_hq:].zero_()   # hq / hchoices gradient rows; the hall rows are written by the attention backward
        d_hall = L.d_arena[:L.row_hq].view(L.N, L.K, L.T, self.wp)
        d_hq = L.d_arena[L.row_hq:L.row_hch].view(L.N, L.JQ, self.wp)
        T = L.groups["text"]
        if self.use_question_att:
            # g1 feeds both the scorer and the question attention: accumulate its second gradient in place
            # (the hq region of d_arena is still zero here, so accumulate mode is exact for it too)
            L.qatt.backward(L.hq, L.g1.view(L.N, 1, self.wp), L.q_mask, L.ones_mask, qW, qb, dgq,
                            d_hq.view(L.N, 1, L.JQ, self.wp), dg1.view(L.N, 1, self.wp), dqW, dqb, accumulate=True)
        else:
            T.op.last_state_bwd(dgq, T.segs[0]["s0"], T.segs[0]["count"], L.d_arena)
        T.op.last_state_bwd(dgch.view(-1, self.wp), T.segs[1]["s0"], T.segs[1]["count"], L.d_arena)
        if self.use_time_warp:
            # attention gradient w.r.t. the warped tensor (masked rows zeroed: the warp backward walks every row),
            # then through the warp into the hall rows of the arena and into lq -> the question encoder's last state
            tw_p = (P.view(self.N_TW_WH_W), P.view(self.N_TW_WH_B), P.view(self.N_TW_WC_W), P.view(self.N_TW_WC_B))
            tw_g = (P.view(self.N_TW_WH_W, True), P.view(self.N_TW_WH_B, True), P.view(self.N_TW_WC_W, True), P.view(self.N_TW_WC_B, True))
            L.d_lq.zero_()
            if L.shadow:
                L.att.backward_shadow(L.warp_tab, L.hq, L.hall_mask.view(L.N, L.K, L.T), L.q_mask, aW, ab, dg1, L.d_warp, d_hq,
                                      daW, dab, accumulate=3)
                L.tw.backward_shadow(L.shadow_tab, L.lq, *tw_p, L.d_warp, d_hall, L.d_lq, *tw_g)
            else:
                if L.d_tscale is not None:
                    L.d_tscale.zero_()
                L.att.backward(L.ctx, L.hq, L.hall_mask.view(L.N, L.K, L.T), L.q_mask, aW, ab, dg1, L.d_warp, d_hq, daW, dab,
                               accumulate=3, tscale=L.tscale, d_tscale=L.d_tscale)
                L.tw.backward(L.hall, L.lq, *tw_p, L.d_warp, d_hall, L.d_lq, *tw_g, d_scale_att=L.d_tscale)
            T.op.last_state_bwd(L.d_lq, T.segs[0]["s0"], T.segs[0]["count"], L.d_arena)
        elif L.shadow:
            L.att.backward_shadow(L.shadow_tab, L.hq, L.hall_mask.view(L.N, L.K, L.T), L.q_mask, aW, ab, dg1, d_hall, d_hq,
                                  daW, dab, accumulate=2)
        else:
            L.att.backward(L.hall, L.hq, L.hall_mask.view(L.N, L.K, L.T), L.q_mask, aW, ab, dg1, d_hall, d_hq, daW, dab,
                           accumulate=2)

    def zero_grad(self):
        self.params.grad.zero_()

