"""model.py of the reference (`--modelname` without FVTA): the soft-attention baselines the paper compares against.

Same encoders, scorer, loss, feed and checkpoint surface as model_v2 (the two files share model.py:460-800 /
model_v2.py:490-836 up to whitespace); what differs is the block between the encoders and the scorer
(model.py:831-983), built here from the SAME HIP kernels as the FVTA path:

  use_ml_att        one 1-D `attention` (model.py:117-186) per context stream            :834-850
  (default)         mean over the albums of the streams' last LSTM states                :868-885
  use_mm_att        `attention` over the K per-stream vectors, else their mean           :901-909
  use_direct_links  `attention` over ALL context rows, added to (or replacing) the above :916-953
  use_choices_att   `attention_keeprank1` of every choice against the question           :966-968
  use_question_att  `attention` of the question against the K per-stream vectors         :977-978

Every 1-D attention is fvta_attn_fwd / fvta_attn_bwd with K = 1 (feat_order 1: model.py:149's simiMatrix-2 feature
order).  The encoder output arena is laid out [N][stream 0 rows | stream 1 rows | ...] WITHOUT the JMAX padding of the
FVTA context tensor, so the direct-links attention reads it as one dense [N, V, w] tensor and each per-stream attention
reads its slice through fvta_attn_desc.hinfo_stride -- no concat, no copy.  attention_keeprank1 is the same kernel at
batch N * num_choice with the question tiled per choice (fvta_rows_broadcast; its gradient comes back through
fvta_rows_reduce).

use_bidirection (model.py:169-177, 297-307, 895-897, 905-906, 968-969, 979-980): every attention that takes the flag also
returns the question attended by each of its rows, averaged over the rows; [h_a ; q_a] goes through a `bidrection_squash`
linear.  The max-pooled half keeps the streaming backward, the reversed half is dense in (row, question position): its
gradient runs through fvta_attn_qside_bwd + fvta_attn_logits_bwd (csrc/attn_dense.hip).  concat (:888-889, :987-991):
the K per-stream vectors side by side instead of attended / averaged, choices and question lifted to that width by two
linears.  Flag sets the reference's own graph construction rejects raise ValueError here (use_bidirection + use_ml_att:
tf.stack of [N,4d] and [N,2d]; concat + use_question_att: `g1` undefined; concat + use_direct_links: [N,2d] + [N,12d]).

use_tgif_ml_att (:851-866, attention_tgif :210-245 per stream): two linears into an mlp_dim = d space, a third down to a
score, softmax over the stream's rows, exp_mask applied to the PROBABILITIES (as written, :234), weighted sum,
tanh(linear) + lq -- composed from fvta_linear_fwd/bwd (over the stream's rows in place: the _blk forms), fvta_softmax_fwd/bwd,
fvta_exp_mask, fvta_wsum_fwd/bwd.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import model_v2, ops
from .model_v2 import _cfg


def get_model(config):
    """model.py:12-16."""
    return Model(config, "model_%s" % getattr(config, "modelname", "memexqa"))


def _stream_masked(cell, dims):
    """model.py:838-850: at / ad / when / where ([N,M,J] text streams) are attended under (stream mask & q_mask) with
    config.simiMatrix; pts (:849) and pis (:850) are called without hq_mask and without simiMatrix: NO mask is applied
    (:137 wants both) and the similarity is the default 1."""
    return cell == "text" and len(dims) == 3


class Model(model_v2.Model):
    STREAMS = (("at", "text", 3), ("ad", "text", 3), ("when", "text", 3), ("where", "text", 3), ("pts", "text", 4),
               ("pis", "image", 3))
    N_ML_W, N_ML_B = "attention/multi_layer_attention/%s/att_logits/W", "attention/multi_layer_attention/%s/att_logits/b"
    N_MM_W, N_MM_B = "attention/multi_modal_attention/mm_att/att_logits/W", "attention/multi_modal_attention/mm_att/att_logits/b"
    N_FULL_W, N_FULL_B = "attention/direct_links/full_att/att_logits/W", "attention/direct_links/full_att/att_logits/b"
    N_CATT_W, N_CATT_B = "choices_emb/choices_att/att_logits/W", "choices_emb/choices_att/att_logits/b"
    # bidrection_squash linears by call site (model.py:897, 906, 969, 980) and the concat linears (:990-991)
    N_SQ = {"g1": "attention/bidrection_squash/%s", "mm": "attention/multi_modal_attention/bidrection_squash/%s",
            "catt": "choices_emb/bidrection_squash/%s", "qatt": "question_emb/bidrection_squash/%s"}
    N_CC = {"ch": "output/gchoice_trans_concat/%s", "q": "output/gq_trans_concat/%s"}
    N_TG = "attention/multi_layer_attention/%s/%s/%s"          # stream, one of mlp_q / mlp_h / preatt / final, W / b
    SHADOW_OK = False            # the 1-D attentions / direct links below read the fp32 rows of the arena
    HAS_VIS_TENSORS = False      # no C / C_win / warp_h / hall here: Tester.step_vis fails as it does on the reference's model.py

    def __init__(self, config, scope="model", text_in=None, img_in=None, device=None):
        self.bi = bool(_cfg(config, "use_bidirection", False))
        self.concat = bool(_cfg(config, "concat", False))
        if self.bi and _cfg(config, "use_ml_att", False):
            raise ValueError("use_bidirection + use_ml_att: tf.stack of [N,4d] (at .. where) with [N,2d] (pts, pis: "
                             "model.py:849-850 pass no bidirect) cannot be built")
        if self.concat and (_cfg(config, "use_question_att", False) or _cfg(config, "use_direct_links", False)):
            raise ValueError("concat: `g1` is undefined for question_att (model.py:978); full_a [N,2d] cannot be added to / "
                             "scored against the [N,12d] concat (:952, :1013)")
        if int(_cfg(config, "simiMatrix", 1)) not in (1, 2, 3):
            raise ValueError("similarity matrix not implemented")            # model.py:152-154 (sys.exit there)
        self.use_ml_att = bool(_cfg(config, "use_ml_att", False))
        self.use_tgif = bool(_cfg(config, "use_tgif_ml_att", False)) and not self.use_ml_att     # :834 / :851 if / elif
        self.use_mm_att = bool(_cfg(config, "use_mm_att", False)) and not self.concat     # (:888-909: concat skips it)
        self.use_direct_links = bool(_cfg(config, "use_direct_links", False))
        self.direct_links_only = self.use_direct_links and bool(_cfg(config, "direct_links_only", False))
        self.use_choices_att = bool(_cfg(config, "use_choices_att", False))
        # context streams in stacking order (model.py:892): (name, cell, rank of the mask).  The reference has these six;
        # `ctx_streams` in the config overrides them (synthetic shapes with another K).
        self.streams = tuple(tuple(s) for s in (_cfg(config, "ctx_streams", None) or self.STREAMS))
        cfg = dict(config) if isinstance(config, dict) else dict(vars(config))
        cfg.update(use_time_warp=False, use_time_warp_att=False, use_bidirection=False)   # no time warp; bidirection: here
        super().__init__(cfg, scope, text_in, img_in, device)
        self.config = config
        self.scorer_tanh = False             # model.py:1011-1013: linear(...) without add_tanh
        self.ml_att_logits = self.mm_att_logits = None

    def _no_photo(self):
        return False                         # model.py never reads --no_photo: its graph always has the six streams

    @staticmethod
    def streams_of(inputs):
        """`ctx_streams` config entry for an oracle-format inputs dict"""
        return tuple(("ctx%d" % k, st.get("cell", "text"), st["mask"].dim()) for k, st in enumerate(inputs["ctx"]))

    # ------------------------------------------------------------ parameters
    def _simi_of(self, k):
        _, cell, rank = self.streams[k]
        return self.simi if _stream_masked(cell, (0,) * rank) else 1

    def _attention_param_specs(self, F):
        wp = self.wp
        feat = {1: 3 * wp, 2: 2 * wp, 3: 4 * wp}
        specs = {}
        if self.use_ml_att:
            for k, (name, _, _) in enumerate(self.streams):
                specs[self.N_ML_W % name], specs[self.N_ML_B % name] = (feat[self._simi_of(k)],), (1,)
        if self.use_tgif:
            dp = self.dp
            for name, _, _ in self.streams:
                for lin, shp in (("mlp_q", (wp, dp)), ("mlp_h", (wp, dp)), ("preatt", (dp,)), ("final", (wp, wp))):
                    specs[self.N_TG % (name, lin, "W")] = shp
                    specs[self.N_TG % (name, lin, "b")] = (shp[1],) if len(shp) == 2 else (1,)
        if self.use_mm_att:
            specs[self.N_MM_W], specs[self.N_MM_B] = (F,), (1,)
        if self.use_direct_links:
            specs[self.N_FULL_W], specs[self.N_FULL_B] = (F,), (1,)
        if self.use_choices_att:
            specs[self.N_CATT_W], specs[self.N_CATT_B] = (F,), (1,)
        if self.use_question_att:
            specs[self.N_QATT_W], specs[self.N_QATT_B] = (F,), (1,)
        if self.bi and not self.concat:
            specs[self.N_SQ["g1"] % "W"], specs[self.N_SQ["g1"] % "b"] = (wp, wp), (wp,)
            for key, on in (("mm", self.use_mm_att), ("qatt", self.use_question_att)):
                if on:
                    specs[self.N_SQ[key] % "W"], specs[self.N_SQ[key] % "b"] = (2 * wp, wp), (wp,)
        if self.bi and self.use_choices_att:
            specs[self.N_SQ["catt"] % "W"], specs[self.N_SQ["catt"] % "b"] = (2 * wp, wp), (wp,)
        if self.concat:
            K = len(self.streams)
            for key in ("ch", "q"):
                specs[self.N_CC[key] % "W"], specs[self.N_CC[key] % "b"] = (wp, K * wp), (K * wp,)
        return specs

    def _scorer_width(self):
        return len(self.streams) * self.wp if self.concat else self.wp

    def _is_w2d(self, name):
        return name.endswith(("bidrection_squash/W", "_trans_concat/W", "/mlp_q/W", "/mlp_h/W", "/final/W"))

    def _is_bfeat(self, name):
        return name.endswith(("bidrection_squash/b", "_trans_concat/b", "/mlp_q/b", "/mlp_h/b", "/final/b"))

    def wd_multipliers(self):
        """model.py's add_wd sites (:320-327): the reader scope (:802), every attention / attention_keeprank1 call's own
        scope (:184, :315), image_trans_linear (:95 via :611), conv1d once per call -- seven (:534-540)."""
        out = {}
        for name in self.params.specs:
            if "bidrection_squash/" in name:                # linear(...) without wd (:897, 906, 969, 980)
                continue
            if name.startswith(("reader/", "attention/", "choices_emb/", "question_emb/",
                                "emb/image/image_transform/image_trans_linear/")):
                out[name] = 1
            elif name.startswith("emb/conv/conv1d/"):
                out[name] = 7
        return out

    def _oracle_key_map(self):
        m = super()._oracle_key_map()
        for k in ("att_W", "att_b", "WH_W", "WH_b", "WC_W", "WC_b"):
            m.pop(k)
        for k, (name, _, _) in enumerate(self.streams):
            m["ml%d_W" % k], m["ml%d_b" % k] = self.N_ML_W % name, self.N_ML_B % name
        m.update(mm_W=self.N_MM_W, mm_b=self.N_MM_B, full_W=self.N_FULL_W, full_b=self.N_FULL_B, catt_W=self.N_CATT_W,
                 catt_b=self.N_CATT_B)
        for k, (name, _, _) in enumerate(self.streams):
            for key, lin in (("q", "mlp_q"), ("h", "mlp_h"), ("p", "preatt"), ("f", "final")):
                m["tg%d_%s_W" % (k, key)], m["tg%d_%s_b" % (k, key)] = self.N_TG % (name, lin, "W"), self.N_TG % (name, lin, "b")
        for key, name in self.N_SQ.items():
            m["sq_%s_W" % key], m["sq_%s_b" % key] = name % "W", name % "b"
        for key, name in self.N_CC.items():
            m["cc_%s_W" % key], m["cc_%s_b" % key] = name % "W", name % "b"
        return m

    # ---------------------------------------------------------------- layout
    def _plan_arena(self, L, ctx, training):
        """[context N * V | hq N*JQ | hchoices N*C*JA] with V = sum over the streams of M * (rows per album): row
        (n, stream k, album m, j) = n * V + off_k + m * per_k + j  (model.py:923-936's `full`, in place)."""
        if len(ctx) != len(self.streams) or any(c != s[1] or len(d) != s[2] for (c, d), s in zip(ctx, self.streams)):
            raise ValueError("context streams %s do not match the model's %s" % (ctx, self.streams))
        dev, wp = self.dev, self.wp
        N, M, JQ, C, JA = L.N, L.M, L.JQ, L.C, L.JA
        L.per = [int(np.prod(dims[2:])) for _, dims in ctx]
        L.Vk = [M * p for p in L.per]
        L.off = [int(x) for x in np.concatenate([[0], np.cumsum(L.Vk)[:-1]])]
        L.V = int(sum(L.Vk))
        L.row_hall, L.row_hq = 0, N * L.V
        L.row_hch = L.row_hq + N * JQ
        L.rows = L.row_hch + N * C * JA
        L.arena = torch.zeros(L.rows, wp, dtype=torch.float32, device=dev)
        L.d_arena = torch.zeros(L.rows, wp, dtype=torch.float32, device=dev) if training else None
        L.hall = L.arena[:L.row_hq].view(N, L.V, wp)
        L.hq = L.arena[L.row_hq:L.row_hch].view(N, JQ, wp)
        L.hch = L.arena[L.row_hch:].view(N, C, JA, wp)
        # context masks, stream-major: stream k's [N, V_k] block starts at N * off_k
        L.hall_mask = torch.zeros(N * L.V, dtype=torch.uint8, device=dev)

        def seq_rows(k, dims):
            n = torch.arange(N).view(N, 1, 1)
            m = torch.arange(M).view(1, M, 1)
            ji = torch.zeros(1, 1, 1, dtype=torch.int64) if len(dims) == 3 else torch.arange(dims[2]).view(1, 1, -1) * dims[3]
            return (n * L.V + L.off[k] + m * L.per[k] + ji).reshape(-1)
        return seq_rows

    def _host_ctx_mask(self, L):
        return np.zeros(L.N * L.V, np.uint8)

    def _put_ctx_mask(self, L, buf, k, m):
        o = L.N * L.off[k]
        buf[o:o + L.N * L.Vk[k]].reshape(L.N, L.M, L.per[k])[...] = m

    def _stream_mask(self, L, k):
        o = L.N * L.off[k]
        return L.hall_mask[o:o + L.N * L.Vk[k]]

    def _build_attention(self, L, training):
        dev, wp = self.dev, self.wp
        N, K, JQ, C, JA = L.N, L.K, L.JQ, L.C, L.JA
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        att = lambda n, t, jq, simi, stride=0: ops.FocalAttention(n, 1, t, jq, wp, simi, False, feat_order=1, hinfo_stride=stride)
        L.g1s, L.d_g1s = z(N, K, wp), (z(N, K, wp) if training else None)        # model.py:892  g1 [N,K,2d]
        L.g1_a, L.g1, L.dg_k = z(N, wp), z(N, wp), z(N, wp)
        L.lq, L.lch = z(N, wp), z(N, C, wp)
        L.masked = [_stream_masked(cell, dims) for cell, _, dims in L.ctx_slots]
        if self.use_ml_att:
            L.ml = [att(N, L.Vk[k], JQ, self._simi_of(k), L.V * wp) for k in range(K)]
        elif self.use_tgif:
            dp = self.dp
            L.d_lq = z(N, wp)
            L.tg = [SimpleNamespace(q_in=z(N, dp), pre=z(N, L.Vk[k], dp), score=z(N, L.Vk[k]), p=z(N, L.Vk[k]), att=z(N, L.Vk[k]),
                                    attended=z(N, wp), fin=z(N, wp), d_attended=z(N, wp), d_att=z(N, L.Vk[k]),
                                    d_score=z(N, L.Vk[k]), d_pre=z(N, L.Vk[k], dp) if training else None, d_q_in=z(N, dp))
                    for k in range(K)]
        else:
            L.cnt = [L.groups[cell].segs[si]["count"] // N for cell, si, _ in L.ctx_slots]    # sequences per example
            L.last = [z(N * c, wp) for c in L.cnt]
            L.d_last = [z(N * c, wp) for c in L.cnt] if training else None
        L.mm = att(N, K, JQ, self.simi) if self.use_mm_att else None
        L.full = att(N, L.V, JQ, self.simi) if self.use_direct_links else None
        L.qatt = att(N, JQ, K, self.simi) if self.use_question_att else None
        if self.use_choices_att:
            L.catt = att(N * C, JA, JQ, self.simi)
            L.hq4, L.d_hq4 = z(N * C, JQ, wp), (z(N * C, JQ, wp) if training else None)
            L.cmask = torch.zeros(N * C, JA, dtype=torch.uint8, device=dev)
            L.qmask4 = torch.zeros(N, C, JQ, dtype=torch.uint8, device=dev)
        # use_bidirection: per call site the [h_a ; q_a] pair, its squashed output and the dense logit gradient
        L.bi = {}
        if self.bi:
            for key, on, R, V, J in (("mm", self.use_mm_att, N, K, JQ), ("catt", self.use_choices_att, N * C, JA, JQ),
                                     ("qatt", self.use_question_att, N, JQ, K)):
                if on:
                    L.bi[key] = SimpleNamespace(R=R, V=V, JQ=J, cat=z(R, 2 * wp), q_a=z(R, wp), out=z(R, wp), lg=None,
                                                d_cat=z(R, 2 * wp), d_ha=z(R, wp), d_qa=z(R, wp), dA=z(R, V, J))
            if not self.concat:
                L.g1s_raw, L.d_g1s_raw = z(N, K, wp), z(N, K, wp)                # the stack before its squash (:897)
        if self.concat:                                                          # :987-991
            Kw = K * wp
            L.gch_in, L.gq_cc, L.gch_cc = z(N, C, wp), z(N, Kw), z(N, C, Kw)
            L.d_gch_in, L.d_gq_in = z(N, C, wp), z(N, wp)

    def load_inputs(self, inputs, training=False):
        if "at" in inputs:
            inputs = self.inputs_from_feed(inputs)
        L = super().load_inputs(inputs, training)
        if self.use_choices_att:                                               # choices_mask, q_mask tiled per choice
            m = inputs["choices"]["mask"]
            m = m if torch.is_tensor(m) else torch.from_numpy(np.ascontiguousarray(m))
            L.cmask.copy_(m.reshape(L.N * L.C, L.JA).to(torch.uint8))
            L.qmask4.copy_(L.q_mask[:, None, :].expand(L.N, L.C, L.JQ))
        return L

    # --------------------------------------------------------------- forward
    def _pv(self, name, grad=False):
        return self.params.view(name, grad)

    def _stream_ptr(self, L, k, grad=False):
        """the arena from stream k's first row on (the strided attention starts there)"""
        return (L.d_arena if grad else L.arena).view(-1)[L.off[k] * self.wp:]

    def _att_fwd(self, L, key, op, hinfo, hq, hm, qm, Wn, Bn, want_logits):
        """one attention call site: (h_a, logits); under use_bidirection [h_a ; q_a] through its bidrection_squash"""
        W, b = self._pv(Wn), self._pv(Bn)
        if not self.bi:
            return op.forward(hinfo, hq, hm, qm, W, b, want_logits)
        st, wp = L.bi[key], self.wp
        h_a, st.lg = op.forward(hinfo, hq, hm, qm, W, b, True)
        ops.attn_qside_fwd(st.lg, hq, st.q_a, st.R, st.V, st.JQ, wp)            # model.py:169-177 / 297-307
        ops.rows_reduce(h_a, st.cat, st.R, 1, wp, 2 * wp)                       # tf.concat([h_a, q_a])
        ops.rows_reduce(st.q_a, st.cat.view(-1)[wp:], st.R, 1, wp, 2 * wp)
        ops.linear_fwd(st.cat, self._pv(self.N_SQ[key] % "W"), self._pv(self.N_SQ[key] % "b"), st.out, st.R, 2 * wp, wp)
        return st.out, st.lg

    def _att_bwd(self, L, key, op, hinfo, hq, hm, qm, Wn, Bn, d_out, d_hinfo, d_hq, accumulate):
        W, b, dW, db = self._pv(Wn), self._pv(Bn), self._pv(Wn, True), self._pv(Bn, True)
        if not self.bi:
            op.backward(hinfo, hq, hm, qm, W, b, d_out, d_hinfo, d_hq, dW, db, accumulate=accumulate)
            return
        st, wp = L.bi[key], self.wp
        sW, sb = self.N_SQ[key] % "W", self.N_SQ[key] % "b"
        ops.linear_bwd(st.cat, self._pv(sW), None, d_out, st.d_cat, self._pv(sW, True), self._pv(sb, True), st.R, 2 * wp, wp)
        ops.rows_broadcast(st.d_cat, st.d_ha, st.R, 1, wp, 2 * wp)
        ops.rows_broadcast(st.d_cat.view(-1)[wp:], st.d_qa, st.R, 1, wp, 2 * wp)
        op.backward(hinfo, hq, hm, qm, W, b, st.d_ha, d_hinfo, d_hq, dW, db, accumulate=accumulate)   # the max-pooled half
        ops.attn_qside_bwd(st.lg, hq, st.d_qa, st.dA, d_hq, st.R, st.V, st.JQ, wp)                    # the reversed half
        op.logits_bwd(hinfo, hq, W, st.dA, d_hinfo, d_hq, dW, db)

    def _attend(self, L, want_logits):
        """model.py:831-991.  Sets L.g1 (g1_all), L.gq, L.lch (gchoices)."""
        N, K, JQ, C, wp = L.N, L.K, L.JQ, L.C, self.wp
        T = L.groups["text"]
        ml_logits = []
        stack = L.g1s_raw if (self.bi and not self.concat) else L.g1s
        if self.use_tgif or not self.use_question_att:
            T.op.last_state(L.arena, T.segs[0]["s0"], T.segs[0]["count"], L.lq)      # lq :663
        for k, (cell, si, dims) in enumerate(L.ctx_slots):
            if self.use_tgif:                                                   # :851-866, attention_tgif :210-245
                self._tgif_fwd(L, k, stack)
            elif self.use_ml_att:                                               # :834-850
                name = self.streams[k][0]
                hm, qm = (self._stream_mask(L, k), L.q_mask) if L.masked[k] else (None, None)
                g, lg = L.ml[k].forward(self._stream_ptr(L, k), L.hq, hm, qm, self._pv(self.N_ML_W % name),
                                        self._pv(self.N_ML_B % name), want_logits)
                ml_logits.append(lg)
                ops.rows_reduce(g, stack.view(-1)[k * wp:], N, 1, wp, K * wp)   # tf.stack slot k (:892)
            else:                                                               # :868-885: reduce_mean of the last states
                G, seg = L.groups[cell], L.groups[cell].segs[si]
                G.op.last_state(L.arena, seg["s0"], seg["count"], L.last[k])
                ops.rows_reduce(L.last[k], stack.view(-1)[k * wp:], N, L.cnt[k], wp, K * wp, 1.0 / L.cnt[k])
        mm_lg = att_lg = q_lg = None
        if self.concat:                                                         # :889: [N,K,w] side by side IS the concat
            L.g1 = L.g1s.view(N, K * wp)
        else:
            if self.bi:                                                         # :896-897
                ops.linear_fwd(L.g1s_raw, self._pv(self.N_SQ["g1"] % "W"), self._pv(self.N_SQ["g1"] % "b"), L.g1s, N * K, wp, wp)
            if self.use_mm_att:                                                 # :901-906 (hinfo_mask None: no mask)
                L.g1_a, mm_lg = self._att_fwd(L, "mm", L.mm, L.g1s, L.hq, None, None, self.N_MM_W, self.N_MM_B, want_logits)
            else:
                ops.rows_reduce(L.g1s, L.g1_a, N, K, wp, wp, 1.0 / K)           # :909
            if self.use_direct_links:                                           # :916-953
                full_a, att_lg = L.full.forward(L.hall, L.hq, None, None, self._pv(self.N_FULL_W), self._pv(self.N_FULL_B),
                                                want_logits)
                ops.rows_broadcast(full_a, L.g1, N, 1, wp)
                if not self.direct_links_only:
                    ops.rows_broadcast(L.g1_a, L.g1, N, 1, wp, accumulate=True)
            else:
                ops.rows_broadcast(L.g1_a, L.g1, N, 1, wp)
        if self.use_choices_att:                                                # :966-969
            ops.rows_broadcast(L.hq, L.hq4, N, C, JQ * wp)                      # the tile of hq per choice (:262)
            gch, _ = self._att_fwd(L, "catt", L.catt, L.hch, L.hq4, L.cmask, L.qmask4, self.N_CATT_W, self.N_CATT_B, False)
            gch = gch.view(N, C, wp)
        else:
            T.op.last_state(L.arena, T.segs[1]["s0"], T.segs[1]["count"], L.lch)     # lchoices :971
            gch = L.lch
        if self.use_question_att:                                               # :977-980 (hq_mask None: no mask)
            L.gq, q_lg = self._att_fwd(L, "qatt", L.qatt, L.hq, L.g1s, None, None, self.N_QATT_W, self.N_QATT_B, want_logits)
        else:
            L.gq = L.lq                                                         # :982
        if self.concat:                                                         # :987-991: lift both to the concat width
            Kw = K * wp
            L.gch_in, L.gq_in = gch, L.gq
            ops.linear_fwd(gch, self._pv(self.N_CC["ch"] % "W"), self._pv(self.N_CC["ch"] % "b"), L.gch_cc, N * C, wp, Kw)
            ops.linear_fwd(L.gq, self._pv(self.N_CC["q"] % "W"), self._pv(self.N_CC["q"] % "b"), L.gq_cc, N, wp, Kw)
            gch, L.gq = L.gch_cc, L.gq_cc
        L.lch = gch
        if want_logits:
            self.ml_att_logits, self.mm_att_logits = ml_logits, mm_lg
        return att_lg, q_lg

    # -------------------------------------------------------------- backward
    def _attend_bwd(self, L, dgq, dg1, dgch):
        N, K, JQ, C, JA, wp = L.N, L.K, L.JQ, L.C, L.JA, self.wp
        T = L.groups["text"]
        # the context rows of the gradient arena are cleared only when their first writer ADDS (last-state scatter, the
        # TGIF weighted sum); an attention over every row of a stream / of the arena writes them with plain stores
        # (accumulate mode 2: valid rows only, the encoders never read the others) and later writers accumulate
        hall_written = self.use_direct_links or self.use_ml_att
        if hall_written and not (self.use_tgif and not self.use_direct_links):
            L.d_arena[L.row_hq:].zero_()
        else:
            L.d_arena.zero_()
        hall_written = self.use_direct_links
        L.d_g1s.zero_()
        d_hall = L.d_arena[:L.row_hq]
        d_hq = L.d_arena[L.row_hq:L.row_hch]
        d_hch = L.d_arena[L.row_hch:]
        g = lambda n: self._pv(n, True)
        if self.concat:
            Kw = K * wp
            cw, cb, qw, qb = self.N_CC["ch"] % "W", self.N_CC["ch"] % "b", self.N_CC["q"] % "W", self.N_CC["q"] % "b"
            ops.linear_bwd(L.gch_in, self._pv(cw), None, dgch, L.d_gch_in, g(cw), g(cb), N * C, wp, Kw)
            ops.linear_bwd(L.gq_in, self._pv(qw), None, dgq, L.d_gq_in, g(qw), g(qb), N, wp, Kw)
            dgch, dgq = L.d_gch_in, L.d_gq_in
        if self.use_choices_att:
            self._att_bwd(L, "catt", L.catt, L.hch, L.hq4, L.cmask, L.qmask4, self.N_CATT_W, self.N_CATT_B,
                          dgch.view(N * C, wp), d_hch, L.d_hq4, 0)
            ops.rows_reduce(L.d_hq4, d_hq, N, C, JQ * wp, accumulate=True)      # gradient of the tile
        else:
            T.op.last_state_bwd(dgch.view(-1, wp), T.segs[1]["s0"], T.segs[1]["count"], L.d_arena)
        if self.use_question_att:
            self._att_bwd(L, "qatt", L.qatt, L.hq, L.g1s, None, None, self.N_QATT_W, self.N_QATT_B, dgq, d_hq, L.d_g1s, 1)
        else:
            T.op.last_state_bwd(dgq, T.segs[0]["s0"], T.segs[0]["count"], L.d_arena)
        if self.concat:
            ops.rows_broadcast(dg1, L.d_g1s, N * K, 1, wp, accumulate=True)      # d g1_a [N,K*w] = d stack
        else:
            if self.use_direct_links:
                L.full.backward(L.hall, L.hq, None, None, self._pv(self.N_FULL_W), self._pv(self.N_FULL_B), dg1, d_hall, d_hq,
                                g(self.N_FULL_W), g(self.N_FULL_B), accumulate=2)      # first writer of every context row
            if not self.direct_links_only:
                if self.use_mm_att:
                    self._att_bwd(L, "mm", L.mm, L.g1s, L.hq, None, None, self.N_MM_W, self.N_MM_B, dg1, L.d_g1s, d_hq, 1)
                else:
                    ops.rows_broadcast(dg1, L.d_g1s, N, K, wp, scale=1.0 / K, accumulate=True)
            elif not self.use_question_att:
                return                                                          # nothing reaches the per-stream vectors
        dstack = L.d_g1s
        if self.bi and not self.concat:                                         # through the stack's squash (:897)
            sW, sb = self.N_SQ["g1"] % "W", self.N_SQ["g1"] % "b"
            ops.linear_bwd(L.g1s_raw, self._pv(sW), None, L.d_g1s, L.d_g1s_raw, g(sW), g(sb), N * K, wp, wp)
            dstack = L.d_g1s_raw
        if self.use_tgif:
            L.d_lq.zero_()
        for k, (cell, si, dims) in enumerate(L.ctx_slots):
            ops.rows_broadcast(dstack.view(-1)[k * wp:], L.dg_k, N, 1, wp, K * wp)          # d g1[:, k, :], dense
            if self.use_tgif:
                self._tgif_bwd(L, k)
            elif self.use_ml_att:
                name = self.streams[k][0]
                hm, qm = (self._stream_mask(L, k), L.q_mask) if L.masked[k] else (None, None)
                L.ml[k].backward(self._stream_ptr(L, k), L.hq, hm, qm, self._pv(self.N_ML_W % name),
                                 self._pv(self.N_ML_B % name), L.dg_k, self._stream_ptr(L, k, True), d_hq,
                                 g(self.N_ML_W % name), g(self.N_ML_B % name), accumulate=1 if hall_written else 2)
            else:
                G, seg = L.groups[cell], L.groups[cell].segs[si]
                ops.rows_broadcast(L.dg_k, L.d_last[k], N, L.cnt[k], wp, scale=1.0 / L.cnt[k])
                G.op.last_state_bwd(L.d_last[k], seg["s0"], seg["count"], L.d_arena)
        if self.use_tgif:                                                       # lq of every stream's mlp_q and "+ lq"
            T.op.last_state_bwd(L.d_lq, T.segs[0]["s0"], T.segs[0]["count"], L.d_arena)

    # ---------------------------------------------------- attention_tgif (model.py:210-245)
    def _tg(self, k, lin, which, grad=False):
        return self._pv(self.N_TG % (self.streams[k][0], lin, which), grad)

    def _tgif_fwd(self, L, k, stack):
        N, K, wp, dp, Vk = L.N, L.K, self.wp, self.dp, L.Vk[k]
        st, hk, blk = L.tg[k], self._stream_ptr(L, k), (Vk, L.V * wp)
        ops.linear_fwd(L.lq, self._tg(k, "mlp_q", "W"), self._tg(k, "mlp_q", "b"), st.q_in, N, wp, dp)                # :225
        ops.linear_fwd(hk, self._tg(k, "mlp_h", "W"), self._tg(k, "mlp_h", "b"), st.pre, N * Vk, wp, dp, blk=blk)     # :226
        ops.rows_broadcast(st.q_in, st.pre, N, Vk, dp, accumulate=True)                                               # :227-228
        ops.linear_fwd(st.pre, self._tg(k, "preatt", "W"), self._tg(k, "preatt", "b"), st.score, N * Vk, dp, 1)        # :229
        ops.softmax_fwd(st.score, st.p, N, Vk)                                                                        # :231
        ops.exp_mask(st.p, self._stream_mask(L, k), st.att, N * Vk)                                                   # :233 (on the probabilities)
        ops.wsum_fwd(hk, st.att, st.attended, N, Vk, wp, target_ld=L.V * wp)                                          # :234-235
        ops.linear_fwd(st.attended, self._tg(k, "final", "W"), self._tg(k, "final", "b"), st.fin, N, wp, wp, add_tanh=True)
        ops.rows_reduce(st.fin, stack.view(-1)[k * wp:], N, 1, wp, K * wp)                                            # :236  tanh(.) + lq
        ops.rows_reduce(L.lq, stack.view(-1)[k * wp:], N, 1, wp, K * wp, accumulate=True)

    def _tgif_bwd(self, L, k):
        N, wp, dp, Vk = L.N, self.wp, self.dp, L.Vk[k]
        st, hk, dhk, blk = L.tg[k], self._stream_ptr(L, k), self._stream_ptr(L, k, True), (Vk, L.V * wp)
        g = lambda lin, which: self._tg(k, lin, which, True)
        ops.rows_broadcast(L.dg_k, L.d_lq, N, 1, wp, accumulate=True)                                                 # "+ lq"
        ops.linear_bwd(st.attended, self._tg(k, "final", "W"), st.fin, L.dg_k, st.d_attended, g("final", "W"), g("final", "b"),
                       N, wp, wp, add_tanh=True)
        ops.wsum_bwd(hk, st.att, st.d_attended, st.d_att, dhk, N, Vk, wp, target_ld=L.V * wp)
        ops.softmax_bwd(st.p, st.d_att, st.d_score, N, Vk)                      # (exp_mask adds a constant: d p = d att)
        ops.linear_bwd(st.pre, self._tg(k, "preatt", "W"), None, st.d_score, st.d_pre, g("preatt", "W"), g("preatt", "b"),
                       N * Vk, dp, 1)
        ops.rows_reduce(st.d_pre, st.d_q_in, N, Vk, dp)                         # gradient of the tile of q_in
        ops.linear_bwd(hk, self._tg(k, "mlp_h", "W"), None, st.d_pre, dhk, g("mlp_h", "W"), g("mlp_h", "b"), N * Vk, wp, dp,
                       accumulate_dx=True, blk=blk)
        ops.linear_bwd(L.lq, self._tg(k, "mlp_q", "W"), None, st.d_q_in, L.d_lq, g("mlp_q", "W"), g("mlp_q", "b"), N, wp, dp,
                       accumulate_dx=True)
