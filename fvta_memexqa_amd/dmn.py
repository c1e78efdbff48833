"""The DMN+ episodic memory the AttentionGRUCell lives in (reference model_dmnplus.py:503-516 around
`_generate_episode` :113-136 and `_get_attention` :89-111), forward AND backward:

    prev_memory = gq
    for i in range(num_hops):
        episode = _generate_episode(prev_memory, gq, facts, facts_length, i, d)       # attention MLP -> softmax over
        prev_memory = relu(dense(concat([prev_memory, episode, gq], 1), d))           # the facts -> AttentionGRU over them
    output = prev_memory

Every hop shares the attention MLP and the AttentionGRU (the reference's reuse flags, :117, :123); each hop has its own
dense layer ("hop_%d").  All arithmetic runs in the HIP library (feature vector, linears, softmax, the gated recurrence
one fact per launch, relu); torch only owns the buffers.  What the reference builds AROUND this module in the DMN+
model -- uni-directional encoders, the bi-GRU fusion of the facts, its own output layer (model_dmnplus.py:300-500,
:518-560) -- is not built: DESIGN.md section 7.

Variable names are TensorFlow's (tf.contrib.layers.fully_connected: weights/biases; tf.layers.dense: kernel/bias;
dynamic_rnn adds "rnn"): memory/attention/fc{1,2}/{weights,biases},
memory/attention_gru/rnn/attention_gru_cell/{gates,candidate,input}/{weights[,biases]}, memory/hop_<i>/dense/{kernel,bias}.
"""
import torch

from . import _lib, ops
from .ops import check, ptr, stream_ptr

CELL = "memory/attention_gru/rnn/attention_gru_cell/"
CELL_VARS = (("gates/weights", lambda d: (2 * d, d)), ("gates/biases", lambda d: (d,)), ("candidate/weights", lambda d: (d, d)),
             ("input/weights", lambda d: (d, d)), ("input/biases", lambda d: (d,)))


class EpisodicMemory:
    def __init__(self, hidden_size, num_hops, device="cuda", seed=0):
        ops.require_gpu()
        self.d, self.num_hops, self.device = int(hidden_size), int(num_hops), torch.device(device)
        d = self.d
        g = torch.Generator().manual_seed(seed)

        def glorot(fi, fo):
            lim = (6.0 / (fi + fo)) ** 0.5
            return ((torch.rand(fi, fo, generator=g) * 2 - 1) * lim).to(self.device)

        zeros = lambda *s: torch.zeros(*s, device=self.device)
        p = {"memory/attention/fc1/weights": glorot(4 * d, d), "memory/attention/fc1/biases": zeros(d),
             "memory/attention/fc2/weights": glorot(d, 1), "memory/attention/fc2/biases": zeros(1)}
        for name, shape in CELL_VARS:
            sh = shape(d)
            p[CELL + name] = zeros(*sh) if name.endswith("biases") else glorot(*sh)   # bias_start 0.0 (attention_gru_cell.py:72)
        for i in range(self.num_hops):
            p["memory/hop_%d/dense/kernel" % i] = glorot(3 * d, d)
            p["memory/hop_%d/dense/bias" % i] = zeros(d)
        self.params = p
        self.grads = {k: torch.zeros_like(v) for k, v in p.items()}
        self._saved = None

    # ------------------------------------------------------------------ weights
    def set_weights(self, weights):
        for k, v in weights.items():
            if k not in self.params:
                raise KeyError(k)
            t = torch.as_tensor(v, dtype=torch.float32).reshape(self.params[k].shape)
            self.params[k].copy_(t)

    def zero_grad(self):
        for gr in self.grads.values():
            gr.zero_()

    # ------------------------------------------------------------------ forward
    def forward(self, gq, facts, facts_length):
        """gq [N,d] (the question vector, also the first memory), facts [N,F,d], facts_length [N] -> output [N,d]."""
        lib, p, d = _lib.load(), self.params, self.d
        gq = gq.to(self.device, torch.float32).contiguous()
        facts = facts.to(self.device, torch.float32).contiguous()
        N, F, _ = facts.shape
        if gq.shape != (N, d) or facts.shape[2] != d:
            raise ValueError("EpisodicMemory: gq %s / facts %s do not fit hidden size %d" % (tuple(gq.shape), tuple(facts.shape), d))
        live = (torch.arange(F, device=self.device)[None, :] < torch.as_tensor(facts_length).to(self.device)[:, None])
        live = live.to(torch.float32).contiguous()                                  # dynamic_rnn's sequence_length (:130)
        new = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        hops, prev = [], gq
        for i in range(self.num_hops):
            h = {"prev": prev}
            # ---- _get_attention for all facts at once (:89-111), softmax over ALL F facts (:120, no mask)
            feats = new(N, F, 4 * d)
            check(lib.fvta_dmn_features(ptr(facts), ptr(gq), ptr(prev), ptr(feats), N, F, d, stream_ptr()), "fvta_dmn_features")
            h["a1"] = new(N * F, d)
            ops.linear_fwd(feats, p["memory/attention/fc1/weights"], p["memory/attention/fc1/biases"], h["a1"], N * F, 4 * d, d, True)
            logit = new(N * F, 1)
            ops.linear_fwd(h["a1"], p["memory/attention/fc2/weights"], p["memory/attention/fc2/biases"], logit, N * F, d, 1)
            h["att"] = new(N, F)
            ops.softmax_fwd(logit, h["att"], N, F)
            # ---- the gated recurrence (:123-134): beyond a row's length dynamic_rnn copies the state through, which the
            # cell does by itself for a gate of 0
            g = new(N * F, 1)
            ops.wsum_fwd(h["att"], live, g, N * F, 1, 1)
            h["gru_in"] = torch.cat([facts, g.view(N, F, 1)], 2).contiguous()       # tf.concat([fact_vecs, attentions], 2)
            states, saved = [torch.zeros(N, d, device=self.device)], []
            for t in range(F):
                x_t = h["gru_in"][:, t].contiguous()
                s_t, sv = ops.attgru_fwd(x_t, states[-1], p[CELL + "gates/weights"], p[CELL + "gates/biases"],
                                         p[CELL + "candidate/weights"], p[CELL + "input/weights"], p[CELL + "input/biases"])
                states.append(s_t)
                saved.append(sv)
            h["states"], h["saved"] = states, saved
            # ---- memory update (:510-514)
            h["cat3"] = torch.cat([prev, states[-1], gq], 1).contiguous()
            pre = new(N, d)
            ops.linear_fwd(h["cat3"], p["memory/hop_%d/dense/kernel" % i], p["memory/hop_%d/dense/bias" % i], pre, N, 3 * d, d)
            h["out"] = new(N, d)
            check(lib.fvta_relu_fwd(ptr(pre), ptr(h["out"]), N * d, stream_ptr()), "fvta_relu_fwd")
            prev = h["out"]
            hops.append(h)
        self._saved = (gq, facts, live, hops)
        return prev

    __call__ = forward

    # ----------------------------------------------------------------- backward
    def backward(self, d_output):
        """Gradient of the last forward: returns (d_gq [N,d], d_facts [N,F,d]); parameter gradients are ADDED to self.grads."""
        if self._saved is None:
            raise RuntimeError("EpisodicMemory.backward before forward")
        lib, p, gr, d = _lib.load(), self.params, self.grads, self.d
        gq, facts, live, hops = self._saved
        N, F, _ = facts.shape
        new = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        d_gq = torch.zeros(N, d, device=self.device)
        d_facts = torch.zeros(N, F, d, device=self.device)
        d_mem = d_output.to(self.device, torch.float32).contiguous()
        for i in reversed(range(self.num_hops)):
            h = hops[i]
            # ---- memory update
            d_pre = new(N, d)
            check(lib.fvta_relu_bwd(ptr(h["out"]), ptr(d_mem), ptr(d_pre), N * d, stream_ptr()), "fvta_relu_bwd")
            d_cat3 = new(N, 3 * d)
            ops.linear_bwd(h["cat3"], p["memory/hop_%d/dense/kernel" % i], h["out"], d_pre, d_cat3,
                           gr["memory/hop_%d/dense/kernel" % i], gr["memory/hop_%d/dense/bias" % i], N, 3 * d, d)
            d_prev = d_cat3[:, :d].contiguous()
            d_state = d_cat3[:, d:2 * d].contiguous()
            d_gq += d_cat3[:, 2 * d:]
            # ---- the gated recurrence, last fact first
            d_gru_in = new(N, F, d + 1)
            for t in reversed(range(F)):
                x_t = h["gru_in"][:, t].contiguous()
                d_x, d_state = ops.attgru_bwd(x_t, h["states"][t], p[CELL + "gates/weights"], p[CELL + "candidate/weights"],
                                              p[CELL + "input/weights"], h["saved"][t], d_state, gr[CELL + "gates/weights"],
                                              gr[CELL + "gates/biases"], gr[CELL + "candidate/weights"],
                                              gr[CELL + "input/weights"], gr[CELL + "input/biases"])
                d_gru_in[:, t] = d_x
            # (the initial state is the constant 0: d_state ends here)
            d_facts += d_gru_in[:, :, :d]
            d_g = d_gru_in[:, :, d].contiguous()                                      # [N,F]
            d_att = new(N * F, 1)
            ops.wsum_fwd(d_g, live, d_att, N * F, 1, 1)                               # d att = d g * (t < length)
            d_logit = new(N * F, 1)
            ops.softmax_bwd(h["att"], d_att, d_logit, N, F)
            # ---- attention MLP
            d_a1 = new(N * F, d)
            logit_unused = d_logit                                                    # (no tanh on fc2: y is not read)
            ops.linear_bwd(h["a1"], p["memory/attention/fc2/weights"], logit_unused, d_logit, d_a1,
                           gr["memory/attention/fc2/weights"], gr["memory/attention/fc2/biases"], N * F, d, 1)
            feats = new(N, F, 4 * d)                                                  # recomputed, not kept
            check(lib.fvta_dmn_features(ptr(facts), ptr(gq), ptr(h["prev"]), ptr(feats), N, F, d, stream_ptr()), "fvta_dmn_features")
            d_feats = new(N * F, 4 * d)
            ops.linear_bwd(feats, p["memory/attention/fc1/weights"], h["a1"], d_a1, d_feats,
                           gr["memory/attention/fc1/weights"], gr["memory/attention/fc1/biases"], N * F, 4 * d, d, True)
            check(lib.fvta_dmn_features_bwd(ptr(facts), ptr(gq), ptr(h["prev"]), ptr(d_feats), ptr(d_facts), ptr(d_gq), ptr(d_prev),
                                            N, F, d, stream_ptr()), "fvta_dmn_features_bwd")
            d_mem = d_prev
        d_gq += d_mem                                                                 # the first memory IS gq (:506)
        return d_gq, d_facts
