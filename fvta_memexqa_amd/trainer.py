"""Counterpart of the reference's trainer.py (Trainer.__init__ 10-27, step 30-40)."""
import os

import numpy as np
import torch

from . import dist, ops


class AdadeltaOptimizer:
    """tf.train.AdadeltaOptimizer(init_lr) as used at trainer.py:16 (rho 0.95, eps 1e-8)."""

    def __init__(self, learning_rate, rho=0.95, epsilon=1e-8):
        self.lr, self.rho, self.eps = float(learning_rate), rho, epsilon
        self.state = None

    SLOTS = ("Adadelta", "Adadelta_1")      # TF's slot variable names: <var>/Adadelta (accum), <var>/Adadelta_1 (accum_update)

    def apply(self, params, grad_scale):
        if self.state is None:
            self.state = (torch.zeros_like(params.flat), torch.zeros_like(params.flat))
        ops.adadelta_step(params.flat, params.grad, self.state[0], self.state[1], self.lr, self.rho, self.eps, grad_scale)


class AdamOptimizer:
    """tf.train.AdamOptimizer(init_lr), the commented-out alternative of trainer.py:17."""

    def __init__(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), beta1, beta2, epsilon
        self.state, self.t = None, 0

    SLOTS = ("Adam", "Adam_1")              # <var>/Adam (m), <var>/Adam_1 (v); beta powers follow from the step count

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
        # a generation-2 pass of CPython's collector over torch's / numpy's heaps is a host pause of tens of milliseconds --
        # several device steps.  Once the model and the first batch exist there is nothing left for it to find, so the
        # first step freezes what is alive and switches the collector off (`gc_freeze`, default on; release_host() undoes it)
        gcf = getattr(config, "gc_freeze", True) if not isinstance(config, dict) else config.get("gc_freeze", True)
        self.gc_freeze, self._owns_host = bool(gcf), False

    def own_host(self):
        """gc.collect(); gc.freeze(); gc.disable() -- once, at the first step (or by hand before an inference loop)."""
        if not self._owns_host:
            import gc
            gc.collect()
            gc.freeze()
            gc.disable()
            self._owns_host = True

    def release_host(self):
        if self._owns_host:
            import gc
            gc.enable()
            gc.unfreeze()
            self._owns_host = False

    def __del__(self):
        try:
            self.release_host()
        except Exception:
            pass

    def step_device(self, layout):
        """One fwd+bwd+update on an already loaded batch; returns the loss as a DEVICE tensor
        (no host sync).  Under data parallelism: one all-reduce of the flat gradient bucket."""
        m = self.model
        if self.gc_freeze and not self._owns_host:
            self.own_host()
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

    # ---- checkpoint / resume.  The reference's tf.train.Saver (main.py:296, 430-440) writes every global variable: the
    # trainables, global_step AND the optimiser's slot variables, so a restored run continues the Adadelta averages.
    # Here: model.save_weights' weights.npz (main.py:578-588 layout) plus `optimizer.npz` beside it, slots keyed
    # "<scope>/<variable>/<slot>:0" in the reference's shapes.
    def save(self, path):
        m = self.model
        m.save_weights(path)
        out = {"optimizer": np.array(type(self.opt).__name__)}
        if self.opt.state is not None:
            for slot, flat in zip(self.opt.SLOTS, self.opt.state):
                for k, v in m.get_weights(flat=flat).items():
                    out["%s/%s/%s:0" % (m.scope, k, slot)] = v
        if hasattr(self.opt, "t"):
            out["step_count"] = np.int64(self.opt.t)
        out["dropout_calls"] = np.array(int(getattr(m, "_dropout_calls", 0)))   # the dropout mask sequence continues on resume
        np.savez(os.path.join(path, "optimizer.npz"), **out)
        return path

    def restore(self, path):
        """weights + global_step (model.load_weights) and, when `optimizer.npz` is there, the optimiser slots"""
        m = self.model
        m.load_weights(path)
        f = os.path.join(path if os.path.isdir(path) else os.path.dirname(path), "optimizer.npz")
        if not os.path.exists(f):
            return False
        with np.load(f) as z:
            if str(z["optimizer"]) != type(self.opt).__name__:
                raise ValueError("checkpoint holds %s slots, the trainer runs %s" % (z["optimizer"], type(self.opt).__name__))
            state = (torch.zeros_like(m.params.flat), torch.zeros_like(m.params.flat))
            have = False
            for slot, flat in zip(self.opt.SLOTS, state):
                got = {}
                for key in z.files:
                    if key.endswith("/%s:0" % slot):
                        name = key[:-len("/%s:0" % slot)]
                        for known in m.params.specs:
                            if name == known or name.endswith("/" + known):
                                got[known] = z[key]
                                break
                if got:
                    missing = [n for n in m.params.specs if n not in got]
                    if missing:
                        raise KeyError("optimizer.npz lacks the %s slot of %s" % (slot, ", ".join(missing)))
                    m.set_weights(got, flat=flat)
                    have = True
            if have:
                self.opt.state = state
            if "step_count" in z.files and hasattr(self.opt, "t"):
                self.opt.t = int(z["step_count"])
            if "dropout_calls" in z.files:
                m._dropout_calls = int(z["dropout_calls"])
        return True

    def restore_tf_checkpoint(self, path, verify=True):
        """main.py:640-665: restore a checkpoint the reference's `saver.save` wrote -- the model's variables by name,
        global_step and, when present, this optimiser's slot variables (the Saver stores them with the trainables)."""
        m = self.model
        slots = m.load_tf_checkpoint(path, verify=verify)
        state = (torch.zeros_like(m.params.flat), torch.zeros_like(m.params.flat))
        found, partial = 0, []
        for slot, flat in zip(self.opt.SLOTS, state):
            named = {}
            for key, val in slots.get(slot, {}).items():
                for known in m.params.specs:
                    if key == known or key.endswith("/" + known):
                        named[known] = val
                        break
            if len(named) == len(m.params.specs):
                m.set_weights(named, flat=flat)
                found += 1
            elif named:
                partial.append("%s (%d of %d variables)" % (slot, len(named), len(m.params.specs)))
        # all slots -> resume; none -> a weights-only file, the optimiser starts fresh (returns False); anything in
        # between is a file written for another model or another optimiser: refuse rather than resume half a state
        if partial or 0 < found < len(self.opt.SLOTS):
            raise KeyError("checkpoint %s holds part of this optimiser's state only: %s" %
                           (path, ", ".join(partial) or "%d of %d slots" % (found, len(self.opt.SLOTS))))
        if found == len(self.opt.SLOTS):
            self.opt.state = state
            if hasattr(self.opt, "t"):
                self.opt.t = m.global_step           # beta powers = beta ** global_step (one apply per step)
                # AdamOptimizer initialises beta1_power to beta1 and multiplies it once per apply (_finish): after k
                # applies the file holds beta1 ** (k + 1).  AdamW.step does t += 1 before it uses beta ** t, so the
                # resumed count is k = (exponent - 1).  A float32 power below ~1e-37 is denormal (no longer a clean
                # power of beta1): keep global_step there.
                b1p = slots.get("beta1_power", {}).get("")
                if b1p is not None and 1e-37 < float(b1p) < 1.0:
                    import math
                    e = int(round(math.log(float(b1p)) / math.log(self.opt.b1)))
                    if e < 1 or abs(self.opt.b1 ** e - float(b1p)) > 1e-3 * float(b1p):
                        raise ValueError("checkpoint %s: beta1_power %g is not a power of beta1 = %g" %
                                         (path, float(b1p), self.opt.b1))
                    self.opt.t = e - 1
        return found == len(self.opt.SLOTS)
