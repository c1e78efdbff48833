"""Counterpart of the reference's tester.py (Tester.step 16-25, step_vis 30-43)."""
import numpy as np


class Tester:
    def __init__(self, model, config, sess=None):
        self.config = config
        self.model = model

    def step(self, sess, batch):
        """tester.py:16-25: returns yp[:num_examples] as a numpy array."""
        batchIdxs, batch_data = batch
        feed_dict = self.model.get_feed_dict(batch_data, is_train=False)
        layout = self.model.load_inputs(feed_dict, training=False)
        yp = self.model.forward(layout).cpu().numpy()
        return yp[:self._num_examples(batch_data, yp)]

    @staticmethod
    def _num_examples(batch_data, yp):
        n = getattr(batch_data, "num_examples", None)
        if n is None and isinstance(batch_data, dict):
            n = batch_data.get("num_examples", yp.shape[0])
        return n

    def trim(self, input_s, num):
        """tester.py:27-28"""
        return [one[:num] if isinstance(one, np.ndarray) else -1 for one in input_s]

    def step_vis(self, sess, batch):
        """tester.py:30-43.  With a Dataset batch (the reference's feed): the reference's 28-tuple, same order --
        yp, C, C_win, att_logits, q_att_logits, the six context masks + q_mask, the six `h*_len`, JXP, warp_h, hall, and
        the id arrays at, ad, when, where, pts, pis, q; yp and (C, C_win, att_logits, q_att_logits, at_mask, pts_mask,
        pis_mask, q_mask) trimmed to num_examples (a non-array becomes -1, as `trim` does there).  With an `inputs`
        dict (no id arrays): the subset the hot path owns, (yp, att_logits, q_att_logits, hall)."""
        batchIdxs, batch_data = batch
        m = self.model
        if not getattr(m, "HAS_VIS_TENSORS", True):
            # tester.py:37 fetches self.model.C, C_win, warp_h, hall: only model_v2.py's Model defines them
            raise AttributeError("Model instance has no attribute 'C' (step_vis needs the FVTA model's tensors)")
        feed = m.get_feed_dict(batch_data, is_train=False)
        L = m.load_inputs(feed, training=False)
        yp = m.forward(L, want_logits=True).cpu().numpy()
        n = self._num_examples(batch_data, yp)
        cpu = lambda t: None if t is None else t.cpu().numpy()
        if "at" not in feed:
            return yp[:n], cpu(m.att_logits)[:n], None if m.q_att_logits is None else cpu(m.q_att_logits)[:n], cpu(m.hall)[:n]
        N, K, M, JMAX, T = L.N, L.K, L.M, L.JMAX, L.T
        ctx5 = lambda t: m.unpad_w(t).reshape(N, K, M, JMAX, m.w).cpu().numpy()
        hall = ctx5(m.hall)
        warp_h = C = C_win = None
        if m.use_time_warp:                                            # model_v2.py:995-1009
            warp_h = ctx5(m.warp_h)
            C = np.broadcast_to(cpu(m.C).reshape(N, T, 1), (N, T, T))   # C_logits[n,t,t'] = c[n,t]
            C_win = np.float32(m.window_t) if m.warp_type == 5 else None
        else:
            warp_h = None
        lens = lambda k: feed[k + "_mask"].sum(-1).astype("int32")    # model_v2.py:667-678
        att, qatt = cpu(m.att_logits), cpu(m.q_att_logits)
        C, C_win, att, qatt, at_mask, pts_mask, pis_mask, q_mask = self.trim(
            [C, C_win, att, qatt, feed["at_mask"], feed["pts_mask"], feed["pis_mask"], feed["q_mask"]], n)
        return (yp[:n], C, C_win, att, qatt, at_mask, feed["ad_mask"], feed["when_mask"], feed["where_mask"], pts_mask,
                pis_mask, q_mask, lens("at"), lens("ad"), lens("when"), lens("where"), lens("pts"), lens("pis"),
                feed["pts"].shape[3], warp_h, hall, feed["at"], feed["ad"], feed["when"], feed["where"], feed["pts"],
                feed["pis"], feed["q"])
