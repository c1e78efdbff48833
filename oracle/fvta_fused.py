"""CPU ORACLE (fused, differentiable) -- TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH.

PARITY UNPINNED (see oracle/fvta_literal.py header): the reference cannot run
here and ships no vectors.  This is the second, independent restatement of the
same reference math (PyTorch-CPU, autograd for gradients).  It differs from
the literal oracle in *how* it computes, not *what*: the similarity logits use
the bilinear decomposition of SURVEY.md section 3.5 (a GEMM plus rank-1 terms
instead of the tile/concat/linear materialisation of model_v2.py:230-249).
Agreement of the two restatements is checked in tests/test_oracle_cross.py.

Only `tests/`, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may
import this module.  It is also the "port" timed as the CPU baseline.
"""
import torch

VERY_NEGATIVE_NUMBER = -1e30


# ---------------------------------------------------------------- helpers ---
def exp_mask(val, mask):
    """utils.py:210-213."""
    return val + (1 - mask.to(val.dtype)) * VERY_NEGATIVE_NUMBER


def softmax(logits):
    """model_v2.py:23-28."""
    return torch.softmax(logits, dim=-1)


def softsel(target, logits):
    """model_v2.py:39-48."""
    return (softmax(logits).unsqueeze(-1) * target).sum(dim=-2)


def linear(x, W, b, add_tanh=False):
    """model_v2.py:75-100."""
    out = x @ W + b
    return torch.tanh(out) if add_tanh else out


def split_att_W(W, w, simiMatrix, feat_order="v2"):
    """Split att_logits/W [F,1] (model_v2.py:242-248) into the per-channel
    vectors of the bilinear form x[t,j] = sum_c U[c] h q + (Rh.h + R2.h^2) +
    (Cq.q + C2.q^2) + b.  Returns (U, Rh, R2) with Cq/C2 = the q-side twins."""
    W = W.reshape(-1)
    z = torch.zeros(w, dtype=W.dtype)
    if simiMatrix == 1:      # [h, q, h*q]
        Wh, Wq, Whq = W[:w], W[w:2 * w], W[2 * w:]
        return Whq, Wh, z, Wq, z
    if simiMatrix == 2:      # v2: [h*q, (h-q)^2]; model.py:149: [(h-q)^2, h*q]
        W1, W2 = (W[:w], W[w:]) if feat_order == "v2" else (W[w:], W[:w])
        return W1 - 2 * W2, z, W2, z, W2
    if simiMatrix == 3:      # [h, q, (h-q)^2, h*q]
        Wh, Wq, W2, Whq = W[:w], W[w:2 * w], W[2 * w:3 * w], W[3 * w:]
        return Whq - 2 * W2, Wh, W2, Wq, W2
    raise ValueError("similarity matrix not implemented")


def simi_logits(h, q, W, b, simiMatrix, add_tanh, feat_order="v2"):
    """h[..., T, w], q[..., JQ, w] (broadcastable lead dims) -> x[..., T, JQ]."""
    w = h.shape[-1]
    if simiMatrix == 4:      # model_v2.py:250-254, eps from tf.nn.l2_normalize
        hn = h * torch.rsqrt(torch.clamp((h * h).sum(-1, keepdim=True), min=1e-12))
        qn = q * torch.rsqrt(torch.clamp((q * q).sum(-1, keepdim=True), min=1e-12))
        return hn @ qn.transpose(-1, -2)
    U, Rh, R2, Cq, C2 = split_att_W(W, w, simiMatrix, feat_order)
    x = (h * U) @ q.transpose(-1, -2)
    x = x + (h @ Rh + (h * h) @ R2).unsqueeze(-1) + (q @ Cq + (q * q) @ C2).unsqueeze(-2) + b.reshape(())
    return torch.tanh(x) if add_tanh else x


def max_over_j(a, max_grad="split"):
    """tf.reduce_max(a_logits, JQ axis) (model_v2.py:268 / 177).  Values are the same in both modes; the GRADIENT of an
    exact tie differs: "split" is TensorFlow's (and torch.amax's): the tied cells share it equally; "first" is what the HIP
    kernels do (deliberate deviation): the FIRST arg-max position takes all of it.  Tie = within 1e-12
    relative of the maximum (identical rows give bit-identical logits on the GPU; a blocked CPU GEMM may differ in the last
    fp64 bit between two copies of a row)."""
    if max_grad == "split":
        return a.amax(dim=-1)
    assert max_grad == "first"
    m = a.detach().amax(dim=-1, keepdim=True)
    first = (a.detach() >= m - 1e-12 * m.abs()).to(torch.int8).argmax(dim=-1, keepdim=True)   # first maximal index
    return torch.gather(a, -1, first).squeeze(-1)


def attention(hinfo, hq, W=None, b=None, hinfo_mask=None, hq_mask=None, simiMatrix=1,
              add_tanh=False, bidirect=False, feat_order="v2", max_grad="split"):
    """model_v2.py:125-201 / model.py:117-186."""
    N, w = hinfo.shape[0], hinfo.shape[-1]
    h = hinfo.reshape(N, -1, w)
    a = simi_logits(h, hq, W, b, simiMatrix, add_tanh, feat_order)       # [N,V,JQ]
    if hinfo_mask is not None and hq_mask is not None:
        mask = hinfo_mask.reshape(N, -1)[:, :, None] & hq_mask[:, None, :]
        a = exp_mask(a, mask)
    h_a = softsel(h, max_over_j(a, max_grad))
    if bidirect:
        q_a = (softmax(a).unsqueeze(-1) * hq[:, None, :, :]).sum(-2).mean(1)
        h_a = torch.cat([h_a, q_a], 1)
    return h_a, a


def attention_3d(hinfo, hq, W=None, b=None, hinfo_mask=None, hq_mask=None, simiMatrix=1,
                 add_tanh=False, time_warp_att=False, C=None, max_grad="split"):
    """model_v2.py:210-298.  max_grad: see max_over_j (the max over T at :288 splits ties in both modes, as the kernels do)."""
    N, K, w = hinfo.shape[0], hinfo.shape[1], hinfo.shape[-1]
    h = hinfo.reshape(N, K, -1, w)
    a = simi_logits(h, hq[:, None], W, b, simiMatrix, add_tanh)          # [N,K,T,JQ]
    if hinfo_mask is not None and hq_mask is not None:
        mask = hinfo_mask.reshape(N, K, -1)[..., None] & hq_mask[:, None, None, :]
        a = exp_mask(a, mask)
    amax = max_over_j(a, max_grad)
    s = a.amax(dim=(3, 2)) if max_grad == "split" else amax.amax(dim=2)
    if time_warp_att:
        amax = (amax.unsqueeze(-1) * C[:, None]).sum(-1)
    u = softsel(h, amax)
    return softsel(u, s), a


# --------------------------------------------------------------- encoders ---
def lstm_direction(x, seq_len, kernel, bias, reverse):
    """[TF-internal] dynamic_rnn over BasicLSTMCell with sequence_length
    (SURVEY 3.6).  `reverse` runs on reverse_sequence(x) and un-reverses the
    outputs.  x[B,J,in] -> out[B,J,d], h_final[B,d]."""
    B, J, din = x.shape
    d = kernel.shape[1] // 4
    Wx, Wh = kernel[:din], kernel[din:]
    ar = torch.arange(J)
    if reverse:
        idx = torch.where(ar[None, :] < seq_len[:, None], seq_len[:, None] - 1 - ar[None, :], ar[None, :])
        x = torch.gather(x, 1, idx[:, :, None].expand(B, J, din))
    zx = x @ Wx + bias
    c = x.new_zeros(B, d)
    h = x.new_zeros(B, d)
    outs = []
    for t in range(J):
        z = zx[:, t] + h @ Wh
        i, j, f, o = z.split(d, dim=1)
        nc = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        nh = torch.tanh(nc) * torch.sigmoid(o)
        live = (t < seq_len)[:, None]
        c = torch.where(live, nc, c)
        h = torch.where(live, nh, h)
        outs.append(torch.where(live, nh, torch.zeros_like(nh)))
    out = torch.stack(outs, 1)
    if reverse:
        out = torch.gather(out, 1, idx[:, :, None].expand(B, J, d))
    return out, h


def dropout_keep_masks(n, keep_prob, seed):
    """The keep decisions of the build's LSTM input dropout for an input of n elements: [2, n] bool, row 0 the forward
    direction's, row 1 the backward direction's (DropoutWrapper(cell, input_keep_prob), model_v2.py:657-661, is called in
    both loops of bidirectional_dynamic_rnn: two independent masks per input element).  TensorFlow's random stream cannot
    be restated; this is the counter-based hash the library uses instead (splitmix64 of seed + golden * (index + 1), top 32
    bits below keep_prob * 2^32) -- same distribution, reproducible."""
    return dropout_keep_flat(2 * n, keep_prob, seed).reshape(2, n)


def dropout_keep_flat(n, keep_prob, seed):
    """keep decisions of elements 0..n-1 of one dropped tensor (the hash of dropout_keep_masks); conv1d's dropout of the
    gathered char embeddings [ntok, W, cdim] (model_v2.py:58-62) uses it with n = ntok * W * cdim."""
    import numpy as np
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = np.uint64(seed & (2 ** 64 - 1)) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    thr = np.uint64(int(float(np.float32(keep_prob)) * 4294967296.0))
    return torch.from_numpy((z >> np.uint64(32)) < thr)


def encode_stream(x, mask, kernel_fw, bias_fw, kernel_bw=None, bias_bw=None, input_keep=None, keep_prob=1.0):
    """model_v2.py:694-826 (one bidirectional_dynamic_rnn call + concats).  input_keep [2, *x.shape] bool with keep_prob:
    the DropoutWrapper of model_v2.py:657-661 -- tf.nn.dropout(inputs, keep_prob) = inputs / keep_prob * keep inside each
    direction's loop, an independent mask per direction."""
    if kernel_bw is None:
        kernel_bw, bias_bw = kernel_fw, bias_fw
    lead = x.shape[:-2]
    J, din = x.shape[-2], x.shape[-1]
    xf = x.reshape(-1, J, din)
    ln = mask.reshape(-1, J).to(torch.int64).sum(1)
    if input_keep is not None:
        # (the library scales by fp32(1) / fp32(keep_prob))
        scale = float(torch.tensor(1.0, dtype=torch.float32) / torch.tensor(keep_prob, dtype=torch.float32))
        kf = input_keep.reshape(2, -1, J, din).to(xf.dtype)
        of, hf = lstm_direction(xf * scale * kf[0], ln, kernel_fw, bias_fw, False)
        ob, hb = lstm_direction(xf * scale * kf[1], ln, kernel_bw, bias_bw, True)
    else:
        of, hf = lstm_direction(xf, ln, kernel_fw, bias_fw, False)
        ob, hb = lstm_direction(xf, ln, kernel_bw, bias_bw, True)
    hcat = torch.cat([of, ob], 2)
    last = torch.cat([hf, hb], 1)
    return hcat.reshape(*lead, J, hcat.shape[-1]), last.reshape(*lead, last.shape[-1])


def context_tensor(streams, masks):
    """model_v2.py:863-914."""
    JMAX = max(s.shape[2] for s in streams)
    hs = [torch.nn.functional.pad(s, (0, 0, 0, JMAX - s.shape[2])) for s in streams]
    ms = [torch.nn.functional.pad(m, (0, JMAX - m.shape[2])) for m in masks]
    return torch.stack(hs, 1), torch.stack(ms, 1)


def time_warp_closed(hall, lq, WH_W, WH_b, WC_W, WC_b, warp_type=1, window_t=3.0):
    """Closed form of model_v2.py:953-1009: because hall_t1 and
    hall_t2 are the same expression (:979-980) the feature is [h^2, 0] and
    h'[n,k,t] = h[n,k,t] * c[n,t] * count(t)."""
    import math
    N, K, w = hall.shape[0], hall.shape[1], hall.shape[-1]
    h = hall.reshape(N, K, -1, w)
    T = h.shape[2]
    WHv = (h * h) @ WH_W[:w] + WH_b + lq[:, None, None, :]
    c = torch.tanh((WHv @ WC_W + WC_b)[..., 0].sum(1))                 # [N,T]
    ar = torch.arange(T)
    if warp_type == 1:
        cnt = torch.full((T,), float(T))
    elif warp_type == 2:
        cnt = torch.ones(T)
    elif warp_type == 3:
        cnt = (ar + 1).to(torch.float32)
    elif warp_type == 4:
        cnt = (T - ar).to(torch.float32)
    elif warp_type == 5:
        win = int(math.ceil(window_t))
        cnt = (torch.minimum(ar + win, torch.tensor(T - 1)) - torch.clamp(ar - win, min=0) + 1).to(torch.float32)
    else:
        raise Exception("time warping type not implemented")
    scale = (c * cnt.to(c.dtype)[None, :])
    return (h * scale[:, None, :, None]).reshape(hall.shape), c


def time_indication_band(T, warp_type=1, window_t=3.0, dtype=torch.float32):
    """the 0/1 factor time_indication_func multiplies C with (model_v2.py:301-341): [T,T]"""
    import math
    ones = torch.ones(T, T, dtype=dtype)
    if warp_type == 1:
        return ones
    if warp_type == 2:
        return torch.eye(T, dtype=dtype)
    if warp_type == 3:
        return torch.tril(ones)
    if warp_type == 4:
        return torch.triu(ones)
    if warp_type == 5:
        win = int(math.ceil(window_t))
        return torch.triu(torch.tril(ones, win), -win)
    raise Exception("time warping type not implemented")


# ------------------------------------------------- embedding front-end ---
def conv1d(x, filt, bias):
    """model_v2.py:52-70 (keep_prob = 1) as an unfold + matmul: x [B, JX, W, cdim], filt [1, height, cdim, cwdim]."""
    height = filt.shape[1]
    win = x.unfold(2, height, 1)                                  # [B, JX, P, cdim, height]
    win = win.permute(0, 1, 2, 4, 3).reshape(*x.shape[:2], win.shape[2], -1)
    xc = torch.relu(win @ filt[0].reshape(-1, filt.shape[3]) + bias)
    return xc.max(dim=2).values


def embed_tokens(word_ids, char_ids, word_emb_mat, existing_emb_mat, char_emb, filt, bias, char_keep=None, keep_prob=1.0):
    """model_v2.py:524-620 for one text input; see oracle/fvta_literal.py:embed_tokens.  char_keep (bool, shape of
    char_ids + [cdim]) with keep_prob: conv1d's tf.nn.dropout of the gathered char embeddings while training (:58-62)."""
    table = torch.cat([word_emb_mat, existing_emb_mat], 0)
    A = table[word_ids.long()]
    if char_emb is None:
        return A
    Ac = char_emb[char_ids.long()]
    if char_keep is not None:
        scale = float(torch.tensor(1.0, dtype=torch.float32) / torch.tensor(keep_prob, dtype=torch.float32))
        Ac = Ac * scale * char_keep.to(Ac.dtype)
    lead = tuple(char_ids.shape[:-2])
    J, W = char_ids.shape[-2:]
    xc = conv1d(Ac.reshape(-1, J, W, char_emb.shape[1]), filt, bias).reshape(lead + (J, filt.shape[3]))
    return torch.cat([xc, A], -1)


def image_features(pis, image_emb_mat, W=None, b=None, add_tanh=False):
    """model_v2.py:634-645."""
    x = image_emb_mat[pis.long()]
    return x if W is None else linear(x, W, b, add_tanh)


def scorer(gq, g1, gch, W, b, use_eu_output=False, add_tanh=False):
    """model_v2.py:1061-1079."""
    g1t = g1[:, None, :].expand_as(gch)
    gqt = gq[:, None, :].expand_as(gch)
    feats = [gqt, g1t, gch, g1t * gch, gqt * gch]
    if use_eu_output:
        feats += [(g1t - gch) ** 2, (gqt - gch) ** 2]
    logits = linear(torch.cat(feats, 2), W, b, add_tanh=(add_tanh and use_eu_output))[..., 0]
    return logits, torch.softmax(logits, -1)


class _TfSoftmaxXent(torch.autograd.Function):
    """[TF-internal] tf.nn.softmax_cross_entropy_with_logits (model_v2.py:1088).  The TF-1 kernel
    (SoftmaxCrossEntropyWithLogits) emits the per-row loss AND a `backprop` tensor = softmax(logits) - labels; its
    registered gradient is grad_loss[:, None] * backprop -- labels are not assumed to sum to 1.  For a row whose labels
    are all False (the padded rows of a short last batch, model_v2.py:1270) the loss is 0 but d logits is softmax, where
    autograd of -sum(y log softmax) gives 0."""

    @staticmethod
    def forward(ctx, logits, y):
        yf = y.to(logits.dtype)
        ctx.save_for_backward(torch.softmax(logits, -1) - yf)
        return -(yf * torch.log_softmax(logits, -1)).sum(-1)

    @staticmethod
    def backward(ctx, g):
        (backprop,) = ctx.saved_tensors
        return g.unsqueeze(-1) * backprop, None


def softmax_cross_entropy_mean(logits, y, tf_grad=True):
    """model_v2.py:1088-1090: mean over ALL N rows.  tf_grad: gradient of TF-1's kernel (see _TfSoftmaxXent);
    False: plain autograd of the same value."""
    if tf_grad:
        return _TfSoftmaxXent.apply(logits, y).mean()
    return -(y.to(logits.dtype) * torch.log_softmax(logits, -1)).sum(1).mean()


def attention_gru_cell(inputs, state, Wg, bg, Wc, Wi, bi):
    """attention_gru_cell.py:50-70."""
    d = state.shape[1]
    x, g = inputs[:, :d], inputs[:, d:d + 1]
    r = torch.sigmoid(torch.cat([x, state], 1) @ Wg + bg) * (state @ Wc)
    return (1 - g) * state + g * torch.tanh(r + x @ Wi + bi)


def dmn_memory(gq, facts, facts_length, params, num_hops, episodes=None):
    """model_dmnplus.py:503-516 with `_generate_episode` :113-136 / `_get_attention` :89-111: the hop loop of the DMN+
    episodic memory.  params: TF variable names (memory/attention/fc{1,2}/{weights,biases},
    memory/attention_gru/rnn/attention_gru_cell/{gates,candidate,input}/..., memory/hop_<i>/dense/{kernel,bias})."""
    N, F, d = facts.shape
    cell = "memory/attention_gru/rnn/attention_gru_cell/"
    live = (torch.arange(F)[None, :] < torch.as_tensor(facts_length)[:, None]).to(facts.dtype)
    prev = gq
    for i in range(num_hops):
        q3, m3 = gq[:, None, :], prev[:, None, :]
        feats = torch.cat([facts * q3, facts * m3, (facts - q3).abs(), (facts - m3).abs()], 2)             # :93-98
        a1 = torch.tanh(feats @ params["memory/attention/fc1/weights"] + params["memory/attention/fc1/biases"])
        logit = (a1 @ params["memory/attention/fc2/weights"] + params["memory/attention/fc2/biases"])[..., 0]
        att = torch.softmax(logit, 1)                                                                       # :120, all F facts
        state = torch.zeros(N, d, dtype=facts.dtype)
        for t in range(F):                                                                                  # dynamic_rnn, :127-131
            new = attention_gru_cell(torch.cat([facts[:, t], att[:, t:t + 1]], 1), state, params[cell + "gates/weights"],
                                     params[cell + "gates/biases"], params[cell + "candidate/weights"],
                                     params[cell + "input/weights"], params[cell + "input/biases"])
            state = torch.where(live[:, t:t + 1] > 0, new, state)                                            # sequence_length: copy through
        if episodes is not None:
            episodes.append(state)
        prev = torch.relu(torch.cat([prev, state, gq], 1) @ params["memory/hop_%d/dense/kernel" % i]
                          + params["memory/hop_%d/dense/bias" % i])                                          # :510-514
    return prev


def embed_inputs(params, tok, cfg):
    """model_v2.py:524-645 over a whole token-id batch: `tok` has the layout of the encoder-input dict but text
    streams carry ids [..., J] (+ chars [..., J, W]) and the photo stream pis [N, M, JI]; returns the encoder-input
    dict fvta_forward takes.  params: word_emb, existing_emb_mat, (char_emb, conv_filter, conv_bias), (img_W, img_b)."""
    use_char = "char_emb" in params and params.get("char_emb") is not None

    def text(st):   # (a stream's optional "char_keep" with cfg["keep_prob"]: conv1d's dropout masks of a training step)
        return embed_tokens(st["ids"], st.get("chars"), params["word_emb"], params["existing_emb_mat"],
                            params["char_emb"] if use_char else None, params.get("conv_filter"), params.get("conv_bias"),
                            char_keep=st.get("char_keep") if use_char else None, keep_prob=float(cfg.get("keep_prob", 1.0)))

    ctx = []
    for st in tok["ctx"]:
        if st.get("cell", "text") == "image":
            x = image_features(st["pis"], tok["image_emb_mat"].to(params["word_emb"].dtype), params.get("img_W"),
                               params.get("img_b"), bool(cfg.get("add_tanh", False)))
        else:
            x = text(st)
        ctx.append(dict(x=x, mask=st["mask"], cell=st.get("cell", "text"), keep=st.get("keep")))
    return dict(ctx=ctx, q=dict(x=text(tok["q"]), mask=tok["q"]["mask"], keep=tok["q"].get("keep")),
                choices=dict(x=text(tok["choices"]), mask=tok["choices"]["mask"], keep=tok["choices"].get("keep")),
                y=tok.get("y"))


# ------------------------------------------------------------- whole path ---
def fvta_forward(params, inputs, cfg):
    """Same contract as oracle.fvta_literal.fvta_forward, on torch tensors."""
    def cell(name):
        return (params[name + "_kernel"], params[name + "_bias"],
                params.get(name + "_kernel_bw"), params.get(name + "_bias_bw"))

    # (a stream's optional "keep" [2, *x.shape] with cfg["keep_prob"]: the DropoutWrapper masks of a training step)
    kp = float(cfg.get("keep_prob", 1.0))
    hq, lq = encode_stream(inputs["q"]["x"], inputs["q"]["mask"], *cell("text"), input_keep=inputs["q"].get("keep"), keep_prob=kp)
    _, lch = encode_stream(inputs["choices"]["x"], inputs["choices"]["mask"], *cell("text"),
                           input_keep=inputs["choices"].get("keep"), keep_prob=kp)
    hs, ms = [], []
    for st in inputs["ctx"]:
        h, _ = encode_stream(st["x"], st["mask"], *cell(st.get("cell", "text")), input_keep=st.get("keep"), keep_prob=kp)
        m = st["mask"]
        if h.dim() == 5:
            N, M = h.shape[:2]
            h = h.reshape(N, M, -1, h.shape[-1])
            m = m.reshape(N, M, -1)
        hs.append(h)
        ms.append(m)
    hall, hall_mask = context_tensor(hs, ms)
    out = {"hq": hq, "lq": lq, "lchoices": lch, "hall_pre_warp": hall, "hall_mask": hall_mask}
    if cfg.get("use_time_warp", False):
        hall, c = time_warp_closed(hall, lq, params["WH_W"], params["WH_b"], params["WC_W"], params["WC_b"],
                                   cfg.get("warp_type", 1), float(params.get("window_t", 3.0)))
        out["c_warp"] = c
    out["hall"] = hall
    qmask = inputs["q"]["mask"]
    twa = bool(cfg.get("use_time_warp_att", False))            # model_v2.py:1020: C of the time-warp block (:995)
    C = None
    if twa:
        T = out["c_warp"].shape[1]
        C = out["c_warp"][:, :, None] * time_indication_band(T, cfg.get("warp_type", 1), float(params.get("window_t", 3.0)),
                                                             out["c_warp"].dtype)[None]
    g1, att = attention_3d(hall, hq, params.get("att_W"), params.get("att_b"), hall_mask, qmask,
                           simiMatrix=cfg["simiMatrix"], add_tanh=cfg.get("add_tanh", False), time_warp_att=twa, C=C)
    out["g1_all"], out["att_logits"] = g1, att
    if cfg.get("use_question_att", False):
        N = hq.shape[0]
        gq, qatt = attention(hq, g1[:, None, :], params.get("qatt_W"), params.get("qatt_b"), qmask,
                             torch.ones(N, 1, dtype=torch.bool), simiMatrix=cfg["simiMatrix"],
                             add_tanh=cfg.get("add_tanh", False))
        out["q_att_logits"] = qatt
    else:
        gq = lq
    out["gq"] = gq
    logits, yp = scorer(gq, g1, lch, params["out_W"], params["out_b"],
                        cfg.get("use_eu_output", False), cfg.get("add_tanh", False))
    out["logits"], out["yp"] = logits, yp
    if inputs.get("y") is not None:
        out["loss"] = softmax_cross_entropy_mean(logits, inputs["y"], bool(cfg.get("tf_xent_grad", True))) \
            + weight_decay_terms(params, cfg)
    return out


# add_wd call sites of model_v2.py and the variables (oracle short keys) each call covers (model_v2.py:347-354: one
# wd * l2_loss(var) per trainable of the CURRENT variable scope, added to the "losses" collection per call)
WD_COVER = {
    "text_kernel": 1, "text_bias": 1, "text_kernel_bw": 1, "text_bias_bw": 1,          # reader scope, :835-836
    "image_kernel": 1, "image_bias": 1, "image_kernel_bw": 1, "image_bias_bw": 1,
    "att_W": 1, "att_b": 1,                                                              # attention/all, :295-296
    "qatt_W": 1, "qatt_b": 1,                                                            # question_att, :198-199
    "img_W": 1, "img_b": 1,                                                              # image_trans_linear, :96-97
    "conv_filter": 7, "conv_bias": 7,                                                    # conv1d x 7 calls, :564-571
}


def weight_decay_terms(params, cfg):
    """sum of the l2 entries of the "losses" collection (model_v2.py:1094-1095); 0 when --wd is unset"""
    wd = cfg.get("wd", None)
    if not wd:
        return 0.0
    total = 0.0
    for k, mult in WD_COVER.items():
        if params.get(k) is not None and (k != "qatt_W" and k != "qatt_b" or cfg.get("use_question_att", False)):
            total = total + mult * wd * 0.5 * (params[k] ** 2).sum()      # tf.nn.l2_loss = sum(t^2) / 2
    return total


# ---------------------------------------------------- model.py (soft-attention baselines) ---
def attention_keeprank1(hinfo, hq, W, b, hinfo_mask=None, hq_mask=None, simiMatrix=1, bidirect=False):
    """model.py:248-318: one softsel per (n, m), max over the question inside -> [N,M,w]; bidirect (:297-308) appends
    the question attended by every row, averaged over the rows -> [N,M,2w]."""
    N, M, w = hinfo.shape[0], hinfo.shape[1], hinfo.shape[-1]
    h = hinfo.reshape(N, M, -1, w)
    a = simi_logits(h, hq[:, None], W, b, simiMatrix, False, "v1")       # [N,M,V,JQ]
    if hinfo_mask is not None and hq_mask is not None:
        a = exp_mask(a, hinfo_mask.reshape(N, M, -1)[..., None] & hq_mask[:, None, None, :])
    h_a = softsel(h, a.amax(dim=3))
    if bidirect:
        q_a = (softmax(a).unsqueeze(-1) * hq[:, None, None, :, :]).sum(-2).mean(2)
        h_a = torch.cat([h_a, q_a], 2)
    return h_a


def attention_tgif(hinfo, lq, p, hinfo_mask=None):
    """model.py:210-245 (the TGIF-QA attention): p = dict(q_W, q_b, h_W, h_b, p_W, p_b, f_W, f_b) for the scopes mlp_q,
    mlp_h, preatt, final.  exp_mask is applied to the PROBABILITIES (:234), as written."""
    N, w = hinfo.shape[0], hinfo.shape[-1]
    h = hinfo.reshape(N, -1, w)
    q_in = linear(lq, p["q_W"], p["q_b"])
    h_in = linear(h, p["h_W"], p["h_b"])
    score = linear(q_in[:, None, :] + h_in, p["p_W"], p["p_b"])[..., 0]
    att = softmax(score)
    if hinfo_mask is not None:
        att = exp_mask(att, hinfo_mask.reshape(N, -1))
    attended = (h * att.unsqueeze(-1)).sum(1)
    return torch.tanh(linear(attended, p["f_W"], p["f_b"])) + lq, att


def v1_stream_is_masked(st):
    """model.py:838-850: the album-level text streams (at, ad, when, where: x [N,M,J,in]) are attended under
    (stream mask & q_mask) with config.simiMatrix; photo titles and photos (pts :849, pis :850) are called WITHOUT
    hq_mask and without simiMatrix -- no mask is applied (:137 needs both) and the similarity is the default 1."""
    return st.get("cell", "text") == "text" and st["x"].dim() == 4


def model_v1_forward(params, inputs, cfg):
    """model.py:658-1037 from the encoder inputs to the loss: the soft-attention baselines (use_ml_att / use_tgif_ml_att /
    use_mm_att / use_direct_links / use_choices_att / use_question_att / use_bidirection / concat).  Same `inputs` as
    fvta_forward; params as there plus ml{k}_W/b or tg{k}_{q,h,p,f}_{W,b} (per context stream), mm_W/b, full_W/b,
    catt_W/b, the bidrection_squash linears sq_{g1,mm,catt,qatt}_{W,b} and the concat linears cc_{ch,q}_{W,b}.
    Flag combinations the reference's own graph construction rejects are rejected here too (ValueError)."""
    def cell(name):
        return (params[name + "_kernel"], params[name + "_bias"],
                params.get(name + "_kernel_bw"), params.get(name + "_bias_bw"))

    simi = cfg["simiMatrix"]
    bi = bool(cfg.get("use_bidirection", False))
    concat = bool(cfg.get("concat", False))
    if bi and cfg.get("use_ml_att", False):
        raise ValueError("use_bidirection + use_ml_att: tf.stack of [N,4d] (at..where) with [N,2d] (pts, pis: model.py:849-850 "
                         "pass no bidirect) fails")
    if concat and (cfg.get("use_question_att", False) or cfg.get("use_direct_links", False)):
        raise ValueError("concat: g1 is undefined for question_att (model.py:978) and full_a [N,2d] cannot be added to / "
                         "scored against the [N,12d] concat (:952, :1013)")
    qmask = inputs["q"]["mask"]
    hq, lq = encode_stream(inputs["q"]["x"], qmask, *cell("text"))                       # :660-663
    hch, lch = encode_stream(inputs["choices"]["x"], inputs["choices"]["mask"], *cell("text"))   # :767-778
    N, w = hq.shape[0], hq.shape[-1]
    hs, g1s = [], []
    for k, st in enumerate(inputs["ctx"]):
        h, last = encode_stream(st["x"], st["mask"], *cell(st.get("cell", "text")))      # :691-799
        hs.append(h.reshape(N, -1, w))
        if cfg.get("use_ml_att", False):                                                 # :834-850
            if v1_stream_is_masked(st):
                g, _ = attention(h, hq, params.get("ml%d_W" % k), params.get("ml%d_b" % k), st["mask"], qmask,
                                 simiMatrix=simi, feat_order="v1")
            else:
                g, _ = attention(h, hq, params.get("ml%d_W" % k), params.get("ml%d_b" % k), simiMatrix=1)
        elif cfg.get("use_tgif_ml_att", False):                                          # :851-866, mlp_dim = d
            g, _ = attention_tgif(h, lq, {n: params["tg%d_%s" % (k, n)] for n in
                                          ("q_W", "q_b", "h_W", "h_b", "p_W", "p_b", "f_W", "f_b")}, st["mask"])
        else:                                                                            # :868-885: means of the last states
            g = (last.mean(2) if last.dim() == 4 else last).mean(1)
        g1s.append(g)
    out = {"hq": hq, "lq": lq}
    if concat:
        g1_a = torch.cat(g1s, 1)                                                         # :889  [N,K*w]
        out["g1"] = torch.stack(g1s, 1)
    else:
        g1 = torch.stack(g1s, 1)                                                         # :892  [N,K,w]
        if bi:
            g1 = linear(g1, params["sq_g1_W"], params["sq_g1_b"])                        # :896-897
        out["g1"] = g1
        if cfg.get("use_mm_att", False):                                                 # :901-906 (hinfo_mask None: unmasked)
            g1_a, out["mm_att_logits"] = attention(g1, hq, params.get("mm_W"), params.get("mm_b"), simiMatrix=simi,
                                                   feat_order="v1", bidirect=bi)
            if bi:
                g1_a = linear(g1_a, params["sq_mm_W"], params["sq_mm_b"])
        else:
            g1_a = g1.mean(1)                                                            # :909
    if cfg.get("use_direct_links", False):                                               # :916-952
        full = torch.cat(hs, 1)
        full_a, out["att_logits"] = attention(full, hq, params.get("full_W"), params.get("full_b"), simiMatrix=simi,
                                              feat_order="v1")
        g1_all = full_a if cfg.get("direct_links_only", False) else full_a + g1_a
    else:
        g1_all = g1_a
    if cfg.get("use_choices_att", False):                                                # :966-969
        gch = attention_keeprank1(hch, hq, params.get("catt_W"), params.get("catt_b"), inputs["choices"]["mask"], qmask,
                                  simiMatrix=simi, bidirect=bi)
        if bi:
            gch = linear(gch, params["sq_catt_W"], params["sq_catt_b"])
    else:
        gch = lch                                                                        # :971
    if cfg.get("use_question_att", False):                                               # :977-980 (hq_mask None: unmasked)
        gq, out["q_att_logits"] = attention(hq, g1, params.get("qatt_W"), params.get("qatt_b"), simiMatrix=simi,
                                            feat_order="v1", bidirect=bi)
        if bi:
            gq = linear(gq, params["sq_qatt_W"], params["sq_qatt_b"])
    else:
        gq = lq                                                                          # :982
    if concat:                                                                           # :987-991
        gch = linear(gch, params["cc_ch_W"], params["cc_ch_b"])
        gq = linear(gq, params["cc_q_W"], params["cc_q_b"])
    out["g1_all"], out["gq"], out["gchoices"] = g1_all, gq, gch
    logits, yp = scorer(gq, g1_all, gch, params["out_W"], params["out_b"], cfg.get("use_eu_output", False), False)  # :996-1013
    out["logits"], out["yp"] = logits, yp
    if inputs.get("y") is not None:                                                      # :1023-1033
        out["loss"] = softmax_cross_entropy_mean(logits, inputs["y"], bool(cfg.get("tf_xent_grad", True))) \
            + v1_weight_decay_terms(params, cfg)
    return out


def v1_weight_decay_terms(params, cfg):
    """model.py's add_wd call sites (:320-327, one l2 term per trainable of the calling scope): reader (:802), every
    attention()/attention_keeprank1() call's own scope (:184, :315), image_trans_linear (:95 via :611), conv1d x 7."""
    wd = cfg.get("wd", None)
    if not wd:
        return 0.0
    cover = {k: v for k, v in WD_COVER.items() if k not in ("att_W", "att_b", "qatt_W", "qatt_b")}
    on = {"ml": cfg.get("use_ml_att", False), "tg": cfg.get("use_tgif_ml_att", False) and not cfg.get("use_ml_att", False),
          "mm_": cfg.get("use_mm_att", False) and not cfg.get("concat", False), "full_": cfg.get("use_direct_links", False),
          "catt_": cfg.get("use_choices_att", False), "qatt_": cfg.get("use_question_att", False)}
    for k in params:                                                   # (the squash / concat linears pass no wd)
        if any(k.startswith(pre) and flag for pre, flag in on.items()):
            cover[k] = 1
    total = 0.0
    for k, mult in cover.items():
        if params.get(k) is not None:
            total = total + mult * wd * 0.5 * (params[k] ** 2).sum()
    return total
