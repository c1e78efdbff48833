"""Counterpart of the reference's attention_gru_cell.py: `AttentionGRUCell(num_units)`; one call = one
step (attention_gru_cell.py:50-70).  The TF RNNCell creates its variables on first call under scopes
gates/{weights,biases}, candidate/weights, input/{weights,biases}; here they are passed explicitly."""
import torch

from . import ops


class AttentionGRUCell:
    def __init__(self, num_units, input_size=None, activation=None):
        self._num_units = num_units

    @property
    def state_size(self):
        return self._num_units

    @property
    def output_size(self):
        return self._num_units

    def init_params(self, device, seed=0):
        g = torch.Generator().manual_seed(seed)
        d = self._num_units
        lim = lambda fi, fo: (6.0 / (fi + fo)) ** 0.5
        mk = lambda fi, fo: ((torch.rand(fi, fo, generator=g) * 2 - 1) * lim(fi, fo)).to(device)
        return {"gates/weights": mk(2 * d, d), "gates/biases": torch.zeros(d, device=device),   # bias_start 0.0 (:72)
                "candidate/weights": mk(d, d), "input/weights": mk(d, d), "input/biases": torch.zeros(d, device=device)}

    def __call__(self, inputs, state, params, scope=None):
        """inputs [B, num_units+1] (word vector ++ scalar attention), state [B, num_units] -> (new_h, new_h)."""
        if inputs.shape[-1] != self._num_units + 1:
            raise ValueError("Input should be passed as word input concatenated with 1D attention on end axis")  # :54-55
        new_h, self.saved = ops.attgru_fwd(inputs, state, params["gates/weights"], params["gates/biases"],
                                           params["candidate/weights"], params["input/weights"], params["input/biases"])
        return new_h, new_h
