"""CPU ORACLE (literal) -- TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH.

PARITY UNPINNED: the reference (JunweiLiang/FVTA_MemexQA) is Python-2 / TensorFlow-1
and can be neither imported nor executed in this environment, and it ships no
tests, golden vectors or fixtures for this path.  This
file is therefore a *restatement* of the reference math, op for op, in NumPy.
It is pinned only by (1) agreement with the independent fused restatement in
`oracle/fvta_fused.py`, (2) hand-derivable known-answer cases
(tests/test_oracle_known_answers.py) and (3) the algebraic identities checked
in tests/test_oracle_identities.py.

Only `tests/`, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may
import this module.  The product package `fvta_memexqa_amd` never does.

Every function cites the reference file:line it follows (paths relative to
the reference repo root).  TensorFlow-internal semantics that are not in the
reference tree (BasicLSTMCell gate order, dynamic_rnn masking, tf.nn.softmax,
reduce_max, l2_normalize, Adadelta) are restated from TF 1.4's documented
behaviour and marked [TF-internal].
"""
import math

import numpy as np

VERY_NEGATIVE_NUMBER = -1e30  # utils.py:205, model_v2.py:10


# ----------------------------------------------------------------------------
# utils.py:210-244
# ----------------------------------------------------------------------------
def exp_mask(val, mask):
    """utils.py:210-213: val + (1 - float(mask)) * -1e30 (in val's dtype)."""
    val = np.asarray(val)
    m = np.asarray(mask).astype(val.dtype)
    return val + (1 - m) * np.asarray(VERY_NEGATIVE_NUMBER, dtype=val.dtype)


def flatten(tensor, keep):
    """utils.py:218-231."""
    shape = tensor.shape
    start = len(shape) - keep
    left = int(np.prod(shape[:start])) if start > 0 else 1
    return tensor.reshape((left,) + tuple(shape[start:]))


def reconstruct(tensor, ref, keep):
    """utils.py:233-244."""
    pre = ref.shape[: len(ref.shape) - keep]
    kept = tensor.shape[len(tensor.shape) - keep:]
    return tensor.reshape(tuple(pre) + tuple(kept))


# ----------------------------------------------------------------------------
# model_v2.py:23-100 helper ops
# ----------------------------------------------------------------------------
def softmax(logits):
    """model_v2.py:23-28 -> tf.nn.softmax over the last axis [TF-internal:
    subtract the row max, exponentiate, divide by the sum]."""
    logits = np.asarray(logits)
    mx = logits.max(axis=-1, keepdims=True)
    e = np.exp(logits - mx)
    return e / e.sum(axis=-1, keepdims=True)


def softsel(target, logits):
    """model_v2.py:39-48: sum over the second-to-last axis of
    softmax(logits)[..., None] * target."""
    a = softmax(logits)
    return (a[..., None] * target).sum(axis=-2)


def linear(x, W, b, add_tanh=False):
    """model_v2.py:75-100: flatten(x,1) @ W + b, optional tanh, reshape back."""
    flat = flatten(x, 1)
    out = flat @ W + b
    if add_tanh:
        out = np.tanh(out)
    return reconstruct(out, x, 1)


def l2_normalize(x, axis=-1, eps=1e-12):
    """[TF-internal] tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), eps))."""
    ss = (x * x).sum(axis=axis, keepdims=True)
    return x / np.sqrt(np.maximum(ss, eps))


def _simi_features(h_aug, q_aug, simiMatrix, feat_order="v2"):
    """Feature concat of model_v2.py:242/245/248 (feat_order 'v2') or
    model.py:146-151 (feat_order 'v1': simiMatrix 2 is [(h-q)^2, h*q])."""
    if simiMatrix == 1:
        return np.concatenate([h_aug, q_aug, h_aug * q_aug], axis=-1)
    if simiMatrix == 2:
        d2 = (h_aug - q_aug) * (h_aug - q_aug)
        if feat_order == "v2":
            return np.concatenate([h_aug * q_aug, d2], axis=-1)
        return np.concatenate([d2, h_aug * q_aug], axis=-1)
    if simiMatrix == 3:
        d2 = (h_aug - q_aug) * (h_aug - q_aug)
        return np.concatenate([h_aug, q_aug, d2, h_aug * q_aug], axis=-1)
    raise ValueError("similarity matrix not implemented")  # model_v2.py:255-257


def attention(hinfo, hq, W=None, b=None, hinfo_mask=None, hq_mask=None,
              simiMatrix=1, add_tanh=False, bidirect=False, feat_order="v2"):
    """model_v2.py:125-201 (and model.py:117-186 with feat_order='v1',
    add_tanh=False).  Returns (h_a, a_logits[N,V,JQ])."""
    N = hinfo.shape[0]
    w = hinfo.shape[-1]
    JQ = hq.shape[1]
    hinfo = hinfo.reshape(N, -1, w)                       # :135
    if hinfo_mask is not None:
        hinfo_mask = np.asarray(hinfo_mask).reshape(N, -1)  # :138
    V = hinfo.shape[1]
    h_aug = np.broadcast_to(hinfo[:, :, None, :], (N, V, JQ, w))  # tile :143
    q_aug = np.broadcast_to(hq[:, None, :, :], (N, V, JQ, w))     # tile :144
    use_mask = (hinfo_mask is not None) and (hq_mask is not None)  # :146
    if use_mask:
        mask = hinfo_mask[:, :, None] & np.asarray(hq_mask)[:, None, :]  # :151
    if simiMatrix == 4:
        a_logits = (l2_normalize(h_aug) * l2_normalize(q_aug)).sum(-1)  # :165-167
    else:
        feat = _simi_features(h_aug, q_aug, simiMatrix, feat_order)
        a_logits = linear(feat, W, b, add_tanh=add_tanh)[..., 0]       # :155-162
    if use_mask:
        a_logits = exp_mask(a_logits, mask)                            # :176
    h_a = softsel(hinfo, a_logits.max(axis=2))                         # :181
    if bidirect:
        q_a = softsel(q_aug, a_logits)        # :186  [N,V,w]
        q_a = q_a.mean(axis=1)                # :189
        h_a = np.concatenate([h_a, q_a], 1)   # :192
    return h_a, a_logits


def attention_3d(hinfo, hq, W=None, b=None, hinfo_mask=None, hq_mask=None,
                 simiMatrix=1, add_tanh=False, time_warp_att=False, C=None,
                 chunk_rows=256):
    """model_v2.py:210-298.  hinfo[N,K,M,JMAX,w] (or [N,K,T,w]); returns
    (h_a[N,w], a_logits[N,K,T,JQ]).  The tile/concat/linear materialisation
    (:230-249) is done literally but chunked along T to bound memory.
    The `bidirect` branch (:281-292) has a rank mismatch in the reference
    and cannot execute for the 3-D case; it is not restated."""
    N, K = hinfo.shape[0], hinfo.shape[1]
    w = hinfo.shape[-1]
    JQ = hq.shape[1]
    h = hinfo.reshape(N, K, -1, w)                              # :222
    T = h.shape[2]
    use_mask = (hinfo_mask is not None) and (hq_mask is not None)  # :233
    if hinfo_mask is not None:
        hm = np.asarray(hinfo_mask).reshape(N, K, -1)            # :225
    a_logits = np.empty((N, K, T, JQ), dtype=h.dtype)
    for t0 in range(0, T, chunk_rows):
        t1 = min(T, t0 + chunk_rows)
        hc = h[:, :, t0:t1]
        h_aug = np.broadcast_to(hc[:, :, :, None, :], (N, K, t1 - t0, JQ, w))       # :230
        q_aug = np.broadcast_to(hq[:, None, None, :, :], (N, K, t1 - t0, JQ, w))    # :231
        if simiMatrix == 4:
            a = (l2_normalize(h_aug) * l2_normalize(q_aug)).sum(-1)                 # :252-254
        else:
            feat = _simi_features(h_aug, q_aug, simiMatrix, "v2")
            a = linear(feat, W, b, add_tanh=add_tanh)[..., 0]                       # :242-249
        a_logits[:, :, t0:t1] = a
    if use_mask:
        mask = hm[:, :, :, None] & np.asarray(hq_mask)[:, None, None, :]            # :238
        a_logits = exp_mask(a_logits, mask)                                         # :264
    a_maxed = a_logits.max(axis=3)                                                  # :268
    if time_warp_att:                                                               # :269-275
        a_maxed = (a_maxed[:, :, :, None] * C[:, None, :, :]).sum(-1)
    u = softsel(h, a_maxed)                      # inner softsel :278  [N,K,w]
    s = a_logits.max(axis=(3, 2))                # reduce_max [3,2]    [N,K]
    h_a = softsel(u, s)                          # outer softsel       [N,w]
    return h_a, a_logits


# ----------------------------------------------------------------------------
# Encoders: model_v2.py:649-833   [TF-internal cell / rnn semantics, SURVEY 3.6]
# ----------------------------------------------------------------------------
def attention_keeprank1(hinfo, hq, W, b, hinfo_mask=None, hq_mask=None, simiMatrix=1, bidirect=False):
    """model.py:247-314: h_a[N,M,w], one softsel per album, max over the question inside; `bidirect` (:297-307)
    appends the question attended by every row and averaged over the rows -> [N,M,2w]."""
    N, M, w = hinfo.shape[0], hinfo.shape[1], hinfo.shape[-1]
    JQ = hq.shape[1]
    hinfo = hinfo.reshape(N, M, -1, w)
    V = hinfo.shape[2]
    h_aug = np.broadcast_to(hinfo[:, :, :, None, :], (N, M, V, JQ, w))
    q_aug = np.broadcast_to(hq[:, None, None, :, :], (N, M, V, JQ, w))
    if simiMatrix == 1:
        feat = np.concatenate([h_aug, q_aug, h_aug * q_aug], 4)
    elif simiMatrix == 2:
        feat = np.concatenate([(h_aug - q_aug) * (h_aug - q_aug), h_aug * q_aug], 4)     # model.py:280 order
    elif simiMatrix == 3:
        feat = np.concatenate([h_aug, q_aug, (h_aug - q_aug) * (h_aug - q_aug), h_aug * q_aug], 4)
    else:
        raise ValueError("similarity matrix not implemented")
    a_logits = linear(feat, W, b)[..., 0]
    if hinfo_mask is not None and hq_mask is not None:
        mask = hinfo_mask.reshape(N, M, V)[..., None] & hq_mask[:, None, None, :]
        a_logits = exp_mask(a_logits, mask)
    h_a = softsel(hinfo, a_logits.max(3))
    if bidirect:
        q_a = softsel(q_aug, a_logits).mean(axis=2)            # :299-302  [N,M,w]
        h_a = np.concatenate([h_a, q_a], 2)                    # :305
    return h_a


def attention_tgif(hinfo, lq, Wq, bq, Wh, bh, Wp, bp, Wf, bf, hinfo_mask=None):
    """model.py:210-244, literally (incl. exp_mask on the softmax OUTPUT)."""
    N, w = hinfo.shape[0], hinfo.shape[-1]
    hinfo = hinfo.reshape(N, -1, w)
    q_in = linear(lq, Wq, bq)
    h_in = linear(hinfo, Wh, bh)
    score = linear(q_in[:, None, :] + h_in, Wp, bp)[..., 0]
    att = softmax(score)
    if hinfo_mask is not None:
        att = exp_mask(att, hinfo_mask.reshape(N, -1))
    attended = (hinfo * att[..., None]).sum(1)
    return np.tanh(linear(attended, Wf, bf)) + lq, att


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def basic_lstm_cell(x, c, h, kernel, bias, forget_bias=1.0):
    """[TF-internal] tf.nn.rnn_cell.BasicLSTMCell (model_v2.py:652-653):
    z=[x,h]@kernel+bias; i,j,f,o=split(z,4); c'=c*sig(f+fb)+sig(i)*tanh(j);
    h'=tanh(c')*sig(o)."""
    z = np.concatenate([x, h], axis=1) @ kernel + bias
    i, j, f, o = np.split(z, 4, axis=1)
    new_c = c * sigmoid(f + forget_bias) + sigmoid(i) * np.tanh(j)
    new_h = np.tanh(new_c) * sigmoid(o)
    return new_c, new_h


def dynamic_rnn(x, seq_len, kernel, bias):
    """[TF-internal] tf.nn.dynamic_rnn with sequence_length: rows with
    t >= seq_len emit zeros and carry their state through.
    x[B,J,in] -> out[B,J,d], (c[B,d], h[B,d])."""
    B, J, _ = x.shape
    d = kernel.shape[1] // 4
    c = np.zeros((B, d), x.dtype)
    h = np.zeros((B, d), x.dtype)
    out = np.zeros((B, J, d), x.dtype)
    seq_len = np.asarray(seq_len)
    for t in range(J):
        nc, nh = basic_lstm_cell(x[:, t], c, h, kernel, bias)
        live = (t < seq_len)[:, None]
        c = np.where(live, nc, c)
        h = np.where(live, nh, h)
        out[:, t] = np.where(live, nh, 0)
    return out, (c, h)


def reverse_sequence(x, seq_len):
    """[TF-internal] tf.reverse_sequence along axis 1: the first seq_len[b]
    entries of row b are reversed, the rest stay in place."""
    out = x.copy()
    for b, L in enumerate(np.asarray(seq_len)):
        L = int(L)
        if L > 1:
            out[b, :L] = x[b, :L][::-1]
    return out


def bidirectional_dynamic_rnn(x, seq_len, kernel_fw, bias_fw, kernel_bw=None, bias_bw=None):
    """[TF-internal] tf.nn.bidirectional_dynamic_rnn (model_v2.py:694 etc.).
    With TF>=1.2 the reference's `cell_text` object passed as both cell_fw and
    cell_bw shares ONE kernel: pass kernel_bw=None for that.
    Returns (out_fw, out_bw), ((c_fw,h_fw),(c_bw,h_bw))."""
    if kernel_bw is None:
        kernel_bw, bias_bw = kernel_fw, bias_fw
    out_fw, st_fw = dynamic_rnn(x, seq_len, kernel_fw, bias_fw)
    xr = reverse_sequence(x, seq_len)
    out_r, st_bw = dynamic_rnn(xr, seq_len, kernel_bw, bias_bw)
    out_bw = reverse_sequence(out_r, seq_len)
    return (out_fw, out_bw), (st_fw, st_bw)


def encode_stream(x, mask, kernel_fw, bias_fw, kernel_bw=None, bias_bw=None):
    """One bi-LSTM call of model_v2.py:694-826: x[..., J, in], mask[..., J]
    bool.  Returns h[..., J, 2d] (concat fw,bw: :696/:735/:792/:826) and
    last[..., 2d] = concat(fw final .h, bw final .h) (:697/:742/:812)."""
    lead = x.shape[:-2]
    J, din = x.shape[-2], x.shape[-1]
    xf = x.reshape((-1, J, din))
    ln = np.asarray(mask).reshape((-1, J)).astype(np.int32).sum(1)   # :667-678
    (of, ob), ((_, hf), (_, hb)) = bidirectional_dynamic_rnn(xf, ln, kernel_fw, bias_fw, kernel_bw, bias_bw)
    hcat = np.concatenate([of, ob], axis=2)
    last = np.concatenate([hf, hb], axis=1)
    return hcat.reshape(lead + (J, hcat.shape[-1])), last.reshape(lead + (last.shape[-1],))


# ----------------------------------------------------------------------------
# Context tensor: model_v2.py:863-914
# ----------------------------------------------------------------------------
def context_tensor(streams, masks):
    """streams: list of K arrays [N,M,J_k,w] (photo titles already reshaped
    [N,M,JI*JXP,w] as :886); masks likewise [N,M,J_k].  Right-pads each to
    JMAX=max J_k (:869-901) and stacks on a new axis 1 (:905-912)."""
    JMAX = max(s.shape[2] for s in streams)
    hs, ms = [], []
    for s, m in zip(streams, masks):
        pad = JMAX - s.shape[2]
        hs.append(np.pad(s, [(0, 0), (0, 0), (0, pad), (0, 0)]))
        ms.append(np.pad(np.asarray(m, bool), [(0, 0), (0, 0), (0, pad)]))
    return np.stack(hs, axis=1), np.stack(ms, axis=1)


# ----------------------------------------------------------------------------
# Time warp: model_v2.py:301-341, 953-1009 (literal O(T^2) form; small T only)
# ----------------------------------------------------------------------------
def time_indication_band(T, warp_type, window_t=3.0, dtype=np.float64):
    """model_v2.py:301-341 multiplicative [T,T] indicator."""
    if warp_type == 1:
        return np.ones((T, T), dtype)
    if warp_type == 2:
        return np.eye(T, dtype=dtype)
    if warp_type == 3:
        return np.tril(np.ones((T, T), dtype))
    if warp_type == 4:
        return np.triu(np.ones((T, T), dtype))
    if warp_type == 5:
        win = int(math.ceil(window_t))                              # :335
        i = np.arange(T)
        return (np.abs(i[:, None] - i[None, :]) <= win).astype(dtype)  # :337
    raise Exception("time warping type not implemented")             # :341


def time_warp_literal(hall, lq, WH_W, WH_b, WC_W, WC_b, warp_type=1, window_t=3.0):
    """model_v2.py:953-1009 literally (hall_t1 and hall_t2 are the SAME
    expression :979-980).  hall[N,K,M,JMAX,w]; returns (hall', C_logits, C)."""
    N, K = hall.shape[:2]
    w = hall.shape[-1]
    hall_t = hall.reshape(N, K, -1, w)                                # :974
    T = hall_t.shape[2]
    t1 = np.broadcast_to(hall_t[:, :, :, None, :], (N, K, T, T, w))   # :979
    t2 = np.broadcast_to(hall_t[:, :, :, None, :], (N, K, T, T, w))   # :980
    WQ_t = lq[:, None, None, None, :]                                 # :983
    WH = linear(np.concatenate([t1 * t2, (t1 - t2) * (t1 - t2)], 4), WH_W, WH_b)  # :986
    C_logits = linear(WH + WQ_t, WC_W, WC_b)[..., 0]                  # :989
    C_logits = np.tanh(C_logits.sum(axis=1))                          # :990 [N,T,T]
    band = time_indication_band(T, warp_type, window_t, hall.dtype)
    C = C_logits * band[None]                                         # :998
    hall_tt = np.broadcast_to(hall_t[:, :, :, None, :], (N, K, T, T, w))  # :1003
    h_warped = (hall_tt * C[:, None, :, :, None]).sum(-2)             # :1004-1005
    return h_warped.reshape(hall.shape), C_logits, C


# ----------------------------------------------------------------------------
# Scorer + loss: model_v2.py:1053-1096
# ----------------------------------------------------------------------------
# ------------------------------------------------- embedding front-end ---
def conv1d(x, filt, bias):
    """model_v2.py:52-70 with keep_prob = 1: x [B, JX, W, cdim], filt [1, height, cdim, cwdim] ->
    relu(conv2d(x, filt, VALID) + bias) [B, JX, W-height+1, cwdim] -> reduce_max over axis 2."""
    B, JX, W, cdim = x.shape
    _, height, _, cw = filt.shape
    P = W - height + 1
    out = np.zeros((B, JX, P, cw), x.dtype)
    for p in range(P):                      # the literal sliding window of conv2d, stride 1
        win = x[:, :, p:p + height, :]      # [B, JX, height, cdim]
        out[:, :, p, :] = np.tensordot(win, filt[0], axes=([2, 3], [0, 1]))
    xc = np.maximum(out + bias, 0.0)
    return xc.max(axis=2)


def embed_tokens(word_ids, char_ids, word_emb_mat, existing_emb_mat, char_emb, filt, bias):
    """model_v2.py:524-620 for ONE text input (at / ad / when / where / pts / q / choices):
    char lookup -> conv1d -> concat([char part, word part]) with word table = concat([trainable, frozen]).
    word_ids [..., J], char_ids [..., J, W]; returns [..., J, cwdim + wdim].  char_emb None: use_char = False."""
    table = np.concatenate([word_emb_mat, existing_emb_mat], 0)      # :590
    A = table[word_ids]                                              # embedding_lookup
    if char_emb is None:
        return A
    Ac = char_emb[char_ids]                                          # [..., J, W, cdim]
    lead = char_ids.shape[:-2]
    J, W = char_ids.shape[-2:]
    xc = conv1d(Ac.reshape(-1, J, W, char_emb.shape[1]), filt, bias).reshape(lead + (J, filt.shape[3]))
    return np.concatenate([xc, A], -1)                               # :611 (char part FIRST)


def image_features(pis, image_emb_mat, W=None, b=None, add_tanh=False):
    """model_v2.py:634-645: embedding_lookup(image_emb_mat, pis) then optionally image_trans_linear."""
    x = image_emb_mat[pis]
    return x if W is None else linear(x, W, b, add_tanh)


def scorer(gq, g1, gch, W, b, use_eu_output=False, add_tanh=False):
    """model_v2.py:1061-1079.  gq[N,w], g1[N,w], gch[N,C,w] -> logits[N,C], yp."""
    C = gch.shape[1]
    g1t = np.broadcast_to(g1[:, None, :], gch.shape)   # :1061
    gqt = np.broadcast_to(gq[:, None, :], gch.shape)   # :1064
    if use_eu_output:                                  # :1073
        feat = np.concatenate([gqt, g1t, gch, g1t * gch, gqt * gch,
                               (g1t - gch) * (g1t - gch), (gqt - gch) * (gqt - gch)], 2)
        logits = linear(feat, W, b, add_tanh=add_tanh)[..., 0]
    else:                                              # :1075
        feat = np.concatenate([gqt, g1t, gch, g1t * gch, gqt * gch], 2)
        logits = linear(feat, W, b)[..., 0]
    return logits, softmax(logits)                     # :1078-1079


def softmax_cross_entropy_mean(logits, y):
    """model_v2.py:1088-1090: mean over ALL N rows of -sum(y*log_softmax)."""
    y = np.asarray(y).astype(logits.dtype)
    mx = logits.max(axis=1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(logits - mx).sum(1))
    losses = (y * (lse[:, None] - logits)).sum(1)
    return losses.mean()


def softmax_cross_entropy_mean_grad(logits, y):
    """[TF-internal] d(mean loss)/d logits as TF-1 computes it for model_v2.py:1088-1090: the
    SoftmaxCrossEntropyWithLogits kernel returns backprop = softmax(logits) - labels next to the loss and the op's
    gradient is grad_loss[:, None] * backprop (no assumption that a row of labels sums to 1); the mean over all N rows
    contributes 1/N.  A row of all-False labels (padded batch row, model_v2.py:1270) has loss 0 and gradient
    softmax/N."""
    y = np.asarray(y).astype(logits.dtype)
    return (softmax(logits) - y) / logits.shape[0]


# ----------------------------------------------------------------------------
# AttentionGRUCell: attention_gru_cell.py:50-70
# ----------------------------------------------------------------------------
def attention_gru_cell(inputs, state, Wg, bg, Wc, Wi, bi):
    """attention_gru_cell.py:50-70.  inputs[B,d+1] (last column = gate g),
    state[B,d].  Wg[2d,d]+bg (gates, :63), Wc[d,d] no bias (candidate, :66),
    Wi[d,d]+bi (input, :68).  new_h=(1-g)*h+g*tanh(r*(h@Wc)+x@Wi+bi)."""
    d = state.shape[1]
    x, g = inputs[:, :d], inputs[:, d:d + 1]
    r = sigmoid(np.concatenate([x, state], 1) @ Wg + bg)
    r = r * (state @ Wc)
    xx = x @ Wi + bi
    h_hat = np.tanh(r + xx)
    return (1 - g) * state + g * h_hat


# ----------------------------------------------------------------------------
# Optimisers: trainer.py:16-17  [TF-internal update rules]
# ----------------------------------------------------------------------------
def dmn_generate_episode(memory, q_vec, fact_vecs, fact_vecs_length, p):
    """model_dmnplus.py:89-136: `_get_attention` per fact (features [f*q, f*m, |f-q|, |f-m|] -> fc1 tanh -> fc2), softmax over
    ALL facts, then dynamic_rnn(AttentionGRUCell) over [fact, attention] with sequence_length (state carried past the
    length; the episode is the final state).  p: fc1_W [4d,H], fc1_b, fc2_W [H,1], fc2_b, Wg, bg, Wc, Wi, bi."""
    N, F, d = fact_vecs.shape
    atts = []
    for i in range(F):                                     # tf.unstack(fact_vecs, axis=1), :116-118
        fv = fact_vecs[:, i]
        feat = np.concatenate([fv * q_vec, fv * memory, np.abs(fv - q_vec), np.abs(fv - memory)], 1)   # :93-98
        a = np.tanh(feat @ p["fc1_W"] + p["fc1_b"])       # :100-104
        atts.append((a @ p["fc2_W"] + p["fc2_b"])[:, 0])   # :106-109
    att = softmax(np.stack(atts).T)                        # :120-121
    state = np.zeros((N, d), fact_vecs.dtype)
    for t in range(F):                                     # dynamic_rnn, :130-134
        inp = np.concatenate([fact_vecs[:, t], att[:, t:t + 1]], 1)
        new = attention_gru_cell(inp, state, p["Wg"], p["bg"], p["Wc"], p["Wi"], p["bi"])
        live = (t < np.asarray(fact_vecs_length))[:, None]
        state = np.where(live, new, state)
    return state


def adadelta_step(var, grad, accum, accum_update, lr, rho=0.95, eps=1e-8):
    """[TF-internal] tf.train.AdadeltaOptimizer(lr) (trainer.py:16)."""
    accum = rho * accum + (1 - rho) * grad * grad
    update = np.sqrt(accum_update + eps) / np.sqrt(accum + eps) * grad
    accum_update = rho * accum_update + (1 - rho) * update * update
    return var - lr * update, accum, accum_update


def adam_step(var, grad, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """[TF-internal] tf.train.AdamOptimizer(lr) (trainer.py:17, commented out):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); var -= lr_t*m/(sqrt(v)+eps)."""
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    return var - lr_t * m / (np.sqrt(v) + eps), m, v


# ----------------------------------------------------------------------------
# Whole hot path: model_v2.py:649-1096 from encoder inputs to loss
# ----------------------------------------------------------------------------
def fvta_forward(params, inputs, cfg):
    """Forward of the FVTA model from the encoder inputs on.

    inputs: dict with
      'ctx': list of K context streams, each dict(x=[N,M,(JI,)J,in], mask=[N,M,(JI,)J],
             cell='text'|'image'); 4-D-leading streams (photo titles, :789) are
             flattened to [N,M,JI*J] after encoding (:886).
      'q': dict(x=[N,JQ,in], mask=[N,JQ]); 'choices': dict(x=[N,C,JA,in], mask=[N,C,JA]);
      'y': [N,C] bool (optional).
    params: 'text_kernel','text_bias'[, 'text_kernel_bw','text_bias_bw'],
      'image_kernel','image_bias'[, ..._bw], 'att_W','att_b', 'qatt_W','qatt_b',
      'out_W','out_b' [, time-warp: 'WH_W','WH_b','WC_W','WC_b','window_t'].
    cfg: dict(simiMatrix, add_tanh, use_question_att, use_eu_output,
      use_time_warp, warp_type).
    """
    def cell(name):
        return (params[name + "_kernel"], params[name + "_bias"],
                params.get(name + "_kernel_bw"), params.get(name + "_bias_bw"))

    hq, lq = encode_stream(inputs["q"]["x"], inputs["q"]["mask"], *cell("text"))       # :694-697
    _, lch = encode_stream(inputs["choices"]["x"], inputs["choices"]["mask"], *cell("text"))  # :802-812
    hs, ms = [], []
    for st in inputs["ctx"]:
        h, _ = encode_stream(st["x"], st["mask"], *cell(st.get("cell", "text")))
        m = np.asarray(st["mask"], bool)
        if h.ndim == 5:                                   # photo titles :886, :899
            N, M = h.shape[:2]
            h = h.reshape(N, M, -1, h.shape[-1])
            m = m.reshape(N, M, -1)
        hs.append(h)
        ms.append(m)
    hall, hall_mask = context_tensor(hs, ms)              # :863-912
    out = {"hq": hq, "lq": lq, "lchoices": lch, "hall_pre_warp": hall, "hall_mask": hall_mask}
    if cfg.get("use_time_warp", False):                   # :953-1009
        hall, Cl, C = time_warp_literal(hall, lq, params["WH_W"], params["WH_b"], params["WC_W"],
                                        params["WC_b"], cfg.get("warp_type", 1), params.get("window_t", 3.0))
        out["C"] = Cl
        out["C_masked"] = C                                # after time_indication_func (:995): what attention_3d gets
    out["hall"] = hall
    qmask = np.asarray(inputs["q"]["mask"], bool)
    twa = bool(cfg.get("use_time_warp_att", False))
    g1, att = attention_3d(hall, hq, params.get("att_W"), params.get("att_b"), hall_mask, qmask,
                           simiMatrix=cfg["simiMatrix"], add_tanh=cfg.get("add_tanh", False),
                           time_warp_att=twa, C=out["C_masked"] if twa else None)              # :1020, C of :995
    out["g1_all"], out["att_logits"] = g1, att
    if cfg.get("use_question_att", False):                # :1044
        N = hq.shape[0]
        gq, qatt = attention(hq, g1[:, None, :], params.get("qatt_W"), params.get("qatt_b"), qmask,
                             np.ones((N, 1), bool), simiMatrix=cfg["simiMatrix"],
                             add_tanh=cfg.get("add_tanh", False))
        out["q_att_logits"] = qatt
    else:
        gq = lq                                           # :1049
    out["gq"] = gq
    logits, yp = scorer(gq, g1, lch, params["out_W"], params["out_b"],
                        cfg.get("use_eu_output", False), cfg.get("add_tanh", False))  # :1053-1083
    out["logits"], out["yp"] = logits, yp
    if inputs.get("y") is not None:
        out["loss"] = softmax_cross_entropy_mean(logits, inputs["y"])                # :1085-1096
        if cfg.get("wd"):                                                            # add_wd, :347-354 (see fvta_fused.WD_COVER)
            from .fvta_fused import WD_COVER
            for k, mult in WD_COVER.items():
                if params.get(k) is not None and (k not in ("qatt_W", "qatt_b") or cfg.get("use_question_att", False)):
                    out["loss"] = out["loss"] + mult * cfg["wd"] * 0.5 * np.sum(np.asarray(params[k], np.float64) ** 2)
    return out


# ----------------------------------------------------------------------------
# model.py:658-1037 -- the soft-attention baselines, literal
# ----------------------------------------------------------------------------
def model_v1_forward(params, inputs, cfg):
    """Same contract as oracle.fvta_fused.model_v1_forward (float64 NumPy, tile/concat/linear as written)."""
    def cell(name):
        return (params[name + "_kernel"], params[name + "_bias"],
                params.get(name + "_kernel_bw"), params.get(name + "_bias_bw"))

    simi = cfg["simiMatrix"]
    bi = bool(cfg.get("use_bidirection", False))
    concat = bool(cfg.get("concat", False))
    qmask = np.asarray(inputs["q"]["mask"], bool)
    cmask = np.asarray(inputs["choices"]["mask"], bool)
    hq, lq = encode_stream(inputs["q"]["x"], qmask, *cell("text"))                       # :660-663
    hchoices, lchoices = encode_stream(inputs["choices"]["x"], cmask, *cell("text"))     # :767-778
    N, w = hq.shape[0], hq.shape[-1]
    hs, g1s = [], []
    for k, st in enumerate(inputs["ctx"]):
        h, last = encode_stream(st["x"], st["mask"], *cell(st.get("cell", "text")))
        m = np.asarray(st["mask"], bool)
        hs.append(h.reshape(N, -1, w))                                                   # :923-928
        if cfg.get("use_ml_att", False):
            W, b = params.get("ml%d_W" % k), params.get("ml%d_b" % k)
            if st.get("cell", "text") == "text" and np.asarray(st["x"]).ndim == 4:       # at/ad/when/where :838-842
                g, _ = attention(h, hq, W, b, m, qmask, simiMatrix=simi, feat_order="v1", bidirect=bi)
            else:                                                                        # pts / pis :849-850
                g, _ = attention(h, hq, W, b, m, None, simiMatrix=1, feat_order="v1")
        elif cfg.get("use_tgif_ml_att", False):                                          # :853-866
            g, _ = attention_tgif(h, lq, *[params["tg%d_%s" % (k, n)] for n in
                                           ("q_W", "q_b", "h_W", "h_b", "p_W", "p_b", "f_W", "f_b")], hinfo_mask=m)
        else:
            g0 = last.mean(axis=2) if last.ndim == 4 else last                           # :875-880
            g = g0.mean(axis=1)                                                          # :882-887
        g1s.append(g)
    out = {"hq": hq, "lq": lq}
    if concat:
        g1_a = np.concatenate(g1s, 1)                                                    # :889
        out["g1"] = np.stack(g1s, 1)
    else:
        g1 = np.stack(g1s, 1)                                                            # :892 (fails on mixed widths, as tf.stack)
        if bi:
            g1 = linear(g1, params["sq_g1_W"], params["sq_g1_b"])                        # :896-897
        out["g1"] = g1
        if cfg.get("use_mm_att", False):
            g1_a, out["mm_att_logits"] = attention(g1, hq, params.get("mm_W"), params.get("mm_b"), None, qmask,
                                                   simiMatrix=simi, feat_order="v1", bidirect=bi)   # :904
            if bi:
                g1_a = linear(g1_a, params["sq_mm_W"], params["sq_mm_b"])                # :906
        else:
            g1_a = g1.mean(axis=1)                                                       # :909
    if cfg.get("use_direct_links", False):
        full = np.concatenate(hs, 1)                                                     # :929-936
        full_a, out["att_logits"] = attention(full, hq, params.get("full_W"), params.get("full_b"), None, None,
                                              simiMatrix=simi, feat_order="v1")          # :947
        g1_all = full_a if cfg.get("direct_links_only", False) else full_a + g1_a        # :950-953
    else:
        g1_all = g1_a
    if cfg.get("use_choices_att", False):
        gchoices = attention_keeprank1(hchoices, hq, params.get("catt_W"), params.get("catt_b"), cmask, qmask,
                                       simiMatrix=simi, bidirect=bi)                     # :967
        if bi:
            gchoices = linear(gchoices, params["sq_catt_W"], params["sq_catt_b"])        # :969
    else:
        gchoices = lchoices
    if cfg.get("use_question_att", False):
        gq, out["q_att_logits"] = attention(hq, g1, params.get("qatt_W"), params.get("qatt_b"), qmask, None,
                                            simiMatrix=simi, feat_order="v1", bidirect=bi)  # :978 (g1 undefined under concat)
        if bi:
            gq = linear(gq, params["sq_qatt_W"], params["sq_qatt_b"])                    # :980
    else:
        gq = lq
    if concat:                                                                           # :987-991
        gchoices = linear(gchoices, params["cc_ch_W"], params["cc_ch_b"])
        gq = linear(gq, params["cc_q_W"], params["cc_q_b"])
    out["g1_all"], out["gq"], out["gchoices"] = g1_all, gq, gchoices
    logits, yp = scorer(gq, g1_all, gchoices, params["out_W"], params["out_b"], cfg.get("use_eu_output", False), False)
    out["logits"], out["yp"] = logits, yp
    if inputs.get("y") is not None:
        out["loss"] = softmax_cross_entropy_mean(logits, inputs["y"])
    return out
