"""Counterpart of the reference's trainer.py (Trainer.__init__ 10-27, step 30-40)."""
import torch

from . import dist, ops


class AdadeltaOptimizer:
    """tf.train.AdadeltaOptimizer(init_lr) as used at trainer.py:16 (rho 0.95, eps 1e-8)."""

    def __init__(self, learning_rate, rho=0.95, epsilon=1e-8):
        self.lr, self.rho, self.eps = float(learning_rate), rho, epsilon
        self.state = None

    def apply(self, params, grad_scale):
        if self.state is None:
            self.state = (torch.zeros_like(params.flat), torch.zeros_like(params.flat))
        ops.adadelta_step(params.flat, params.grad, self.state[0], self.state[1], self.lr, self.rho, self.eps, grad_scale)


class AdamOptimizer:
    """tf.train.AdamOptimizer(init_lr), the commented-out alternative of trainer.py:17."""

    def __init__(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), beta1, beta2, epsilon
        self.state, self.t = None, 0

    def apply(self, params, grad_scale):
        if self.state is None:
            self.state = (torch.zeros_like(params.flat), torch.zeros_like(params.flat))
        self.t += 1
        ops.adam_step(params.flat, params.grad, self.state[0], self.state[1], self.t, self.lr, self.b1, self.b2, self.eps,
                      grad_scale)


class Trainer:
    def __init__(self, model, config):
        self.config = config
        self.model = model
        name = getattr(config, "optimizer", "adadelta") if not isinstance(config, dict) else config.get("optimizer", "adadelta")
        lr = getattr(config, "init_lr", 0.5) if not isinstance(config, dict) else config.get("init_lr", 0.5)
        self.opt = AdamOptimizer(lr) if name == "adam" else AdadeltaOptimizer(lr)   # trainer.py:16-17
        self.need_dx = False   # gradients into the encoder inputs (for the embedding front-end)

    def step_device(self, layout):
        """One fwd+bwd+update on an already loaded batch; returns the loss as a DEVICE tensor
        (no host sync).  Under data parallelism: one all-reduce of the flat gradient bucket."""
        m = self.model
        m.zero_grad()
        m.forward(layout)
        m.backward(layout, loss_scale=1.0, need_dx=self.need_dx)
        # one flat gradient bucket; its early part (scorer / attention / photo cell) may already be on the wire
        early, m.early_work = getattr(m, "early_work", None), None
        scale = dist.allreduce_grads(m.params.grad, getattr(m.params, "early_numel", 0), early)
        self.opt.apply(m.params, scale)
        m.global_step += 1
        return m.loss

    def step(self, sess, batch, get_summary=False):
        """trainer.py:30-40: batch = (batchIdx, batch_data); returns (loss, summary, train_op)."""
        batchIdx, batch_data = batch
        feed_dict = self.model.get_feed_dict(batch_data, is_train=True)
        layout = self.model.load_inputs(feed_dict, training=True)
        loss = self.step_device(layout)
        if dist.is_dist():          # report the global-batch mean, like the single-process reference's loss
            loss = dist.mean_over_ranks(loss.clone())
        return float(loss.item()), None, None
