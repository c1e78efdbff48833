"""Counterpart of the reference's tester.py (Tester.step 16-25, step_vis 30-43)."""


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
        n = getattr(batch_data, "num_examples", None)
        if n is None and isinstance(batch_data, dict):
            n = batch_data.get("num_examples", yp.shape[0])
        return yp[:n]

    def step_vis(self, sess, batch):
        """tester.py:30-43 subset the hot path owns: yp, att_logits, q_att_logits, hall."""
        batchIdxs, batch_data = batch
        feed_dict = self.model.get_feed_dict(batch_data, is_train=False)
        layout = self.model.load_inputs(feed_dict, training=False)
        yp = self.model.forward(layout, want_logits=True).cpu().numpy()
        n = batch_data.get("num_examples", yp.shape[0]) if isinstance(batch_data, dict) else batch_data.num_examples
        m = self.model
        cpu = lambda t: None if t is None else t.cpu().numpy()[:n]
        return yp[:n], cpu(m.att_logits), cpu(m.q_att_logits), cpu(m.hall)
