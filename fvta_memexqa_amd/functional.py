"""The reference's functional graph helpers with their own keyword signatures (SURVEY 8b "functional ops"):

    softmax(logits, scope=None)                                                     model_v2.py:23-28
    softsel(target, logits, hard=False, hardK=None, scope=None)                      model_v2.py:39-48
    linear(x, output_size, scope, add_tanh=False, wd=None)                           model_v2.py:75-100
    exp_mask(val, mask)                                                              utils.py:210-213
    attention(hinfo, hq, hinfo_mask=None, hq_mask=None, simiMatrix=1, wd=None,
              add_tanh=False, bidirect=False, scope=None) -> (h_a, a_logits)         model_v2.py:125-201
    attention_3d(hinfo, hq, hinfo_mask=None, hq_mask=None, simiMatrix=1, wd=None, add_tanh=False,
                 time_warp_att=False, C=None, bidirect=False, scope=None)            model_v2.py:210-298
    attention_keeprank1(hinfo, hq, hinfo_mask=None, hq_mask=None, simiMatrix=1, wd=None,
                        bidirect=False, scope=None) -> h_a [N,M,w]                   model.py:247-314
    attention_tgif(hinfo, lq, hinfo_mask=None, wd=None, mlp_dim=512, scope=None)     model.py:210-244

Tensors are torch CUDA tensors; every op is one call into libfvta_hip.so (forward only -- training goes through
`Model`, whose backward kernels own the gradients).  Where the reference creates TF variables (`linear`'s W / b, the
`att_logits` linear inside the attentions) the variable lives in a module-level store under the same scoped name
(`variable_scope("attention")` + `scope="all"` -> "attention/all/att_logits/W"), initialised like the reference
(truncated normal 0.1 / zeros) and reused on the next call, which is what `tf.get_variable` under reuse does.
`wd` appends the l2 terms to `losses` like `add_wd` (model_v2.py:347-354).
"""
import contextlib
import ctypes
import zlib

import torch

from . import _lib, ops
from ._lib import check, ptr, stream_ptr
from .model_v2 import SUPPORTED_W

variables = {}      # scoped name -> torch tensor (fp32, CUDA)
losses = []         # the "losses" collection: scalar tensors appended by `wd`
_scope = []


@contextlib.contextmanager
def variable_scope(name):
    _scope.append(name)
    try:
        yield
    finally:
        _scope.pop()


def reset_default_graph():
    variables.clear()
    del losses[:]


def _name(*parts):
    return "/".join([p for p in _scope if p] + [p for p in parts if p])


def get_variable(name, shape, init="trunc_normal", seed=None):
    """tf.get_variable under reuse: create on first use, return the same tensor afterwards."""
    if name in variables:
        if tuple(variables[name].shape) != tuple(shape):
            raise ValueError("variable %s exists with shape %s, asked for %s" % (name, tuple(variables[name].shape), tuple(shape)))
        return variables[name]
    dev = ops.require_gpu()
    if init == "zeros":
        v = torch.zeros(*shape, dtype=torch.float32, device=dev)
    elif init == "glorot":          # tf.get_variable's default / xavier_initializer: uniform(+-sqrt(6 / (fan_in + fan_out)))
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) if seed is None else seed)
        lim = (6.0 / (shape[0] + shape[-1])) ** 0.5
        v = ((torch.rand(*shape, generator=g) * 2 - 1) * lim).to(dev)
    else:
        from .synth import _trunc_normal
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) if seed is None else seed)
        v = _trunc_normal(g, tuple(shape)).to(dev)                  # truncated_normal(stddev=0.1), model_v2.py:88
    variables[name] = v
    return v


def _add_wd(names, wd):
    """add_wd (model_v2.py:347-354) for the variables of the current call's scope"""
    if wd is None or wd == 0.0:
        return
    for n in names:
        t = torch.zeros(1, dtype=torch.float32, device=variables[n].device)
        ops.weight_decay(variables[n].reshape(-1), None, wd, t)
        losses.append(t)


def _f32(t):
    return t.to(torch.float32).contiguous()


def softmax(logits, scope=None):
    logits = _f32(logits)
    out = torch.empty_like(logits)
    J = logits.shape[-1]
    check(_lib.load().fvta_softmax_fwd(ptr(logits), ptr(out), logits.numel() // J, J, stream_ptr()), "fvta_softmax_fwd")
    return out


def softsel(target, logits, hard=False, hardK=None, scope=None):
    """target [..., J, d], logits [..., J] -> [..., d]  (`hard` / `hardK` are accepted and unused, as in the reference)"""
    target, logits = _f32(target), _f32(logits)
    J, d = target.shape[-2], target.shape[-1]
    if tuple(logits.shape) != tuple(target.shape[:-1]):
        raise ValueError("softsel: logits %s do not match target %s" % (tuple(logits.shape), tuple(target.shape)))
    out = torch.empty(*target.shape[:-2], d, dtype=torch.float32, device=target.device)
    check(_lib.load().fvta_softsel_fwd(ptr(target), ptr(logits), ptr(out), logits.numel() // J, J, d, stream_ptr()),
          "fvta_softsel_fwd")
    return out


def exp_mask(val, mask):
    val = _f32(val)
    m = ops.as_mask_u8(mask.expand_as(val) if tuple(mask.shape) != tuple(val.shape) else mask)
    out = torch.empty_like(val)
    check(_lib.load().fvta_exp_mask(ptr(val), ptr(m), ptr(out), val.numel(), stream_ptr()), "fvta_exp_mask")
    return out


def linear_raw(x, W, b, add_tanh=False):
    """flatten(x, 1) . W [in, out] + b through fvta_linear_fwd with explicit weights (no variable scope)"""
    x = _f32(x)
    din, dout = x.shape[-1], W.shape[1]
    y = torch.empty(*x.shape[:-1], dout, dtype=torch.float32, device=x.device)
    check(_lib.load().fvta_linear_fwd(ptr(x), ptr(_f32(W)), ptr(b), ptr(y), x.numel() // din, din, dout, int(add_tanh),
                                      stream_ptr()), "fvta_linear_fwd")
    return y


def linear(x, output_size, scope, add_tanh=False, wd=None):
    x = _f32(x)
    din = x.shape[-1]
    with variable_scope(scope):
        wn, bn = _name("W"), _name("b")
        W = get_variable(wn, (din, int(output_size)))
        b = get_variable(bn, (int(output_size),), init="zeros")
        _add_wd([wn, bn], wd)
    y = torch.empty(*x.shape[:-1], int(output_size), dtype=torch.float32, device=x.device)
    check(_lib.load().fvta_linear_fwd(ptr(x), ptr(W), ptr(b), ptr(y), x.numel() // din, din, int(output_size), int(add_tanh),
                                      stream_ptr()), "fvta_linear_fwd")
    return y


def _pad_channels(t, wp):
    w = t.shape[-1]
    if w == wp:
        return t.contiguous()
    out = torch.zeros(*t.shape[:-1], wp, dtype=torch.float32, device=t.device)
    out[..., :w] = t
    return out


def _attention(hinfo, hq, hinfo_mask, hq_mask, simiMatrix, wd, add_tanh, scope, feat_order, tscale=None):
    """hinfo [N,K,T,w] -> (h_a [N,w], a_logits [N,K,T,JQ]) through fvta_attn_fwd; w is zero padded to a kernel width
    (exact: a zero channel adds nothing to any feature of any similarity), W block-wise with it."""
    if simiMatrix not in (1, 2, 3, 4):
        raise ValueError("similarity matrix not implemented")              # model_v2.py:255-257 (sys.exit there)
    hinfo, hq = _f32(hinfo), _f32(hq)
    N, K, T, w = hinfo.shape
    JQ = hq.shape[1]
    wp = next((c for c in SUPPORTED_W if w <= c), None)
    if wp is None:
        raise ValueError("attention: feature width %d too large (max %d)" % (w, SUPPORTED_W[-1]))
    F = {1: 3, 2: 2, 3: 4, 4: 0}[simiMatrix]
    W = b = None
    with variable_scope(scope):
        if F:
            wn, bn = _name("att_logits", "W"), _name("att_logits", "b")
            Wv = get_variable(wn, (F * w, 1))                              # linear(..., output_size=1, scope="att_logits")
            b = get_variable(bn, (1,), init="zeros")
            W = Wv.reshape(F, w)
            if wp != w:
                W = _pad_channels(W, wp)
            W = W.reshape(-1).contiguous()
            _add_wd([wn, bn], wd)
    op = ops.FocalAttention(N, K, T, JQ, wp, simiMatrix, add_tanh, feat_order=feat_order)
    both = hinfo_mask is not None and hq_mask is not None                  # model_v2.py:146 / 233: only when BOTH are given
    hm = ops.as_mask_u8(hinfo_mask.reshape(N, K, T)) if both else None
    qm = ops.as_mask_u8(hq_mask) if both else None
    h_a, a = op.forward(_pad_channels(hinfo, wp), _pad_channels(hq, wp), hm, qm, W, b, want_logits=True, tscale=tscale)
    return h_a[:, :w].contiguous(), a


def attention_3d(hinfo, hq, hinfo_mask=None, hq_mask=None, simiMatrix=1, wd=None, add_tanh=False, time_warp_att=False,
                 C=None, bidirect=False, scope=None):
    """hinfo [N,K,M,JX,w] (or [N,K,T,w]), hq [N,JQ,w], masks [N,K,M,JX] / [N,JQ] -> (h_a [N,w], a_logits [N,K,T,JQ])."""
    if bidirect:
        raise NotImplementedError("bidirect: the 3-D branch cannot run in the reference either (SURVEY 3.5)")
    N, K, w = hinfo.shape[0], hinfo.shape[1], hinfo.shape[-1]
    tscale = None
    if time_warp_att:
        if C is None:
            raise ValueError("time_warp_att needs C [N,T,T] (model_v2.py:269-275)")
        # model_v2.py:269-275: a_logits_maxed[n,k,t] * sum_t' C[n,t,t'] -- the row sums of C [N,T,T] as a linear layer
        # with an all-ones weight (fvta_linear_fwd)
        C = _f32(C)
        T = C.shape[-1]
        tscale = linear_raw(C.reshape(N * T, T), torch.ones(T, 1, dtype=torch.float32, device=C.device), None).reshape(N, T)
    return _attention(hinfo.reshape(N, K, -1, w), hq, hinfo_mask, hq_mask, simiMatrix, wd, add_tanh,
                      scope or "attention_2vector", 0, tscale=tscale)


def _wsum(target, weights):
    """sum_j weights[r,j] * target[r,j,:] -> [rows, d] (fvta_wsum_fwd)"""
    target, weights = _f32(target), _f32(weights)
    rows, J, d = target.shape
    out = torch.empty(rows, d, dtype=torch.float32, device=target.device)
    check(_lib.load().fvta_wsum_fwd(ptr(target), ptr(weights), ptr(out), rows, J, d, stream_ptr()), "fvta_wsum_fwd")
    return out


def _bidirect_q_a(a_logits, hq, lead):
    """the reversed-direction vector of the `bidirect` branch (model_v2.py:184-188, model.py:169-174, 297-304):
    q_a = reduce_mean over the V rows of softsel(q_aug, a_logits), i.e. every row (masked ones included: their logits are
    all -1e30, so they attend the question uniformly) softmaxes its JQ logits and averages the question with them.
    a_logits [*lead, V, JQ], hq [N, JQ, w] -> [*lead, w].  Three launches: softmax over JQ, the mean over V of the
    weights (the average commutes with the weighted sum), the weighted sum of the question vectors."""
    V, JQ = a_logits.shape[-2], a_logits.shape[-1]
    G = int(torch.tensor(lead).prod().item()) if len(lead) else 1
    p = softmax(a_logits)                                                   # [*lead, V, JQ]
    ones = torch.full((G, V), 1.0 / V, dtype=torch.float32, device=p.device)
    pbar = _wsum(p.reshape(G, V, JQ), ones)                                 # [G, JQ]
    N = hq.shape[0]
    per_n = G // N                                                          # (n, m) groups share n's question
    q = _f32(hq)
    if per_n > 1:
        q = q[:, None].expand(N, per_n, JQ, q.shape[-1]).reshape(G, JQ, q.shape[-1])
    return _wsum(q, pbar).reshape(*lead, q.shape[-1])


def attention(hinfo, hq, hinfo_mask=None, hq_mask=None, simiMatrix=1, wd=None, add_tanh=False, bidirect=False, scope=None):
    """hinfo [N,...,w] flattened to [N,V,w] (model_v2.py:133) -> (h_a [N,w], a_logits [N,V,JQ]); with `bidirect`
    h_a is [N,2w] = concat([h_a, q_a]) (model_v2.py:184-192)."""
    N, w = hinfo.shape[0], hinfo.shape[-1]
    h = hinfo.reshape(N, 1, -1, w)
    hm = hinfo_mask.reshape(N, 1, -1) if hinfo_mask is not None else None
    h_a, a = _attention(h, hq, hm, hq_mask, simiMatrix, wd, add_tanh, scope or "attention_2vector", 0)
    a = a.reshape(N, h.shape[2], hq.shape[1])
    if bidirect:
        h_a = torch.cat([h_a, _bidirect_q_a(a, hq, (N,))], 1)              # tf.concat: memory layout, no arithmetic
    return h_a, a


def attention_keeprank1(hinfo, hq, hinfo_mask=None, hq_mask=None, simiMatrix=1, wd=None, bidirect=False, scope=None):
    """model.py:247-314: hinfo [N,M,...,w] -> h_a [N,M,w] ([N,M,2w] with `bidirect`), each album attended on its own (softsel over the rows of
    (n, m) with the max-over-question logits; no softmax over m).  That is the inner stage of attention_3d with K = M:
    one fvta_attn_fwd, then the per-(n,k) result is read back out of the saved state.  model.py's feature order for
    simiMatrix 2 is [(h-q)^2, h*q] (feat_order 1); no tanh on the logits."""
    if simiMatrix not in (1, 2, 3):
        raise ValueError("similarity matrix not implemented")              # model.py:283-285 (sys.exit there)
    hinfo, hq = _f32(hinfo), _f32(hq)
    N, M, w = hinfo.shape[0], hinfo.shape[1], hinfo.shape[-1]
    h = hinfo.reshape(N, M, -1, w)
    V, JQ = h.shape[2], hq.shape[1]
    wp = next((c for c in SUPPORTED_W if w <= c), None)
    if wp is None:
        raise ValueError("attention: feature width %d too large (max %d)" % (w, SUPPORTED_W[-1]))
    F = {1: 3, 2: 2, 3: 4}[simiMatrix]
    with variable_scope(scope or "attention_2vector"):
        wn, bn = _name("att_logits", "W"), _name("att_logits", "b")
        Wv = get_variable(wn, (F * w, 1))
        b = get_variable(bn, (1,), init="zeros")
        W = _pad_channels(Wv.reshape(F, w), wp).reshape(-1).contiguous()
        _add_wd([wn, bn], wd)
    op = ops.FocalAttention(N, M, V, JQ, wp, simiMatrix, False, feat_order=1)
    both = hinfo_mask is not None and hq_mask is not None
    hm = ops.as_mask_u8(hinfo_mask.reshape(N, M, V)) if both else None
    qm = ops.as_mask_u8(hq_mask) if both else None
    _, a = op.forward(_pad_channels(h, wp), _pad_channels(hq, wp), hm, qm, W, b, want_logits=bool(bidirect))
    u = torch.empty(N, M, wp, dtype=torch.float32, device=h.device)
    check(op.lib.fvta_attn_read_u(ctypes.byref(op.desc), ptr(op.saved), ptr(u), stream_ptr()), "fvta_attn_read_u")
    u = u[..., :w].contiguous()
    if bidirect:                                                            # model.py:297-307 -> [N,M,2w]
        u = torch.cat([u, _bidirect_q_a(a, hq, (N, M))], 2)
    return u


def attention_tgif(hinfo, lq, hinfo_mask=None, wd=None, mlp_dim=512, scope=None):
    """model.py:210-244 (the TGIF-QA attention baseline): score = linear(linear(lq)[:,None] + linear(hinfo)) ->
    softmax over the rows -> exp_mask applied to the PROBABILITIES (as the reference does: masked rows get weight
    -1e30, harmless only because their h is 0) -> weighted sum -> tanh(linear) + lq.  Returns (logits [N,2*mlp_dim],
    att [N,V]); needs lq.shape[-1] == 2 * mlp_dim like the reference's tf.add."""
    hinfo, lq = _f32(hinfo), _f32(lq)
    N, w = hinfo.shape[0], hinfo.shape[-1]
    h = hinfo.reshape(N, -1, w)
    V = h.shape[1]
    with variable_scope(scope or "attention_2vector"):
        q_in = linear(lq, mlp_dim, scope="mlp_q", wd=None)
        h_in = linear(h, mlp_dim, scope="mlp_h", wd=None)
        preatt = (h_in + q_in[:, None, :]).contiguous()                      # tf.tile + tf.add (plumbing)
        score = linear(preatt, 1, scope="preatt").reshape(N, V)
        att = softmax(score)
        if hinfo_mask is not None:
            att = exp_mask(att, hinfo_mask.reshape(N, V))
        attended = torch.empty(N, w, dtype=torch.float32, device=h.device)
        check(_lib.load().fvta_wsum_fwd(ptr(h), ptr(att), ptr(attended), N, V, w, stream_ptr()), "fvta_wsum_fwd")
        final = linear(attended, 2 * mlp_dim, scope="final", add_tanh=True)
        if wd is not None:                                                    # add_wd over the whole scope (:241-242)
            _add_wd([n for n in variables if n.startswith(_name("") )], wd)
    return final + lq, att


# ------------------------------------------------------------------ DMN+ episode (model_dmnplus.py:89-136)
def _get_attention(q_vec, prev_memory, fact_vecs, hidden_size):
    """model_dmnplus.py:89-111 for ALL facts at once (the reference unstacks the facts and reuses the two
    fully_connected layers): facts [N,F,d] -> attention logits [N,F].  Variables attention/fc1/{weights,biases}
    (tanh), attention/fc2/{weights,biases} (tf.contrib.layers.fully_connected: Glorot weights, zero biases)."""
    N, F, d = fact_vecs.shape
    feats = torch.empty(N, F, 4 * d, dtype=torch.float32, device=fact_vecs.device)
    check(_lib.load().fvta_dmn_features(ptr(_f32(fact_vecs)), ptr(_f32(q_vec)), ptr(_f32(prev_memory)), ptr(feats), N, F, d,
                                        stream_ptr()), "fvta_dmn_features")
    with variable_scope("attention"):
        W1 = get_variable(_name("fc1", "weights"), (4 * d, hidden_size), init="glorot")
        b1 = get_variable(_name("fc1", "biases"), (hidden_size,), init="zeros")
        W2 = get_variable(_name("fc2", "weights"), (hidden_size, 1), init="glorot")
        b2 = get_variable(_name("fc2", "biases"), (1,), init="zeros")
    a1 = linear_raw(feats, W1, b1, add_tanh=True)
    return linear_raw(a1, W2, b2).reshape(N, F)


def generate_episode(memory, q_vec, fact_vecs, fact_vecs_length, hop_index, hidden_size, scope=None):
    """model_dmnplus.py:113-136 `_generate_episode`: attention over the facts from (question, previous memory), softmax
    over ALL F facts (no mask in the reference), then the AttentionGRUCell (attention_gru_cell.py:50-70) run by
    dynamic_rnn over the facts with sequence_length: the state is carried past a row's length -- which the cell does by
    itself when the attention gate is 0, so the gates of steps >= length are zeroed.  Returns the episode [N,d].
    Variables live under the current scope (hop_index > 0 reuses them, as the reference's reuse flags do)."""
    from .attention_gru_cell import AttentionGRUCell
    fact_vecs = _f32(fact_vecs)
    N, F, d = fact_vecs.shape
    with variable_scope(scope):
        att = softmax(_get_attention(q_vec, memory, fact_vecs, hidden_size))                  # [N,F]
        live = (torch.arange(F, device=att.device)[None, :] < fact_vecs_length.to(att.device)[:, None]).to(torch.float32)
        g = _wsum(att.reshape(N * F, 1, 1), live.reshape(N * F, 1)).reshape(N, F, 1)          # att * (t < length)
        gru_inputs = torch.cat([fact_vecs, g], 2).contiguous()                               # [N,F,d+1] (tf.concat)
        with variable_scope("attention_gru"):
            cell = AttentionGRUCell(hidden_size)
            names = ("gates/weights", "gates/biases", "candidate/weights", "input/weights", "input/biases")
            shapes = ((2 * d, d), (d,), (d, d), (d, d), (d,))
            params = {n_: get_variable(_name("attention_gru_cell", n_), sh, init="zeros" if n_.endswith("biases") else "glorot")
                      for n_, sh in zip(names, shapes)}
        state = torch.zeros(N, d, dtype=torch.float32, device=att.device)
        for t in range(F):
            state, _ = cell(gru_inputs[:, t].contiguous(), state, params)
    return state
