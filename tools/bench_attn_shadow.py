"""Focal attention over bf16 shadow rows at the metric shape: fp32 rows (fvta_attn_fwd / _bwd) against the shadow kernels with
(a) the half-rows contiguous in row order, (b) the half-rows where the bi-LSTM keeps them -- hs[dir][token][sequence][d], a row's
neighbours 13 MB apart.  usage: python tools/bench_attn_shadow.py [N K M JTOK JQ w]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import ops
N, K, M, JT, JQ, w = (int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (64, 6, 40, 30, 30, 1024)))
T = M * JT
g = torch.Generator(device="cuda").manual_seed(0)
h = (torch.rand(N, K, T, w, device="cuda", generator=g) * 2 - 1).bfloat16()
q = torch.randn(N, JQ, w, device="cuda", generator=g) * 0.5
W = torch.randn(2 * w, device="cuda", generator=g) * 0.1
b = torch.zeros(1, device="cuda")
hm = torch.ones(N, K, T, dtype=torch.uint8, device="cuda")
# the row set of bench.py's dense metric shape (the one `algorithmic_bytes_per_call` is computed on): the LAST stream is the photo
# stream, M rows of its T; FVTA_ATTN_ALL_ROWS=1: every row valid
if not os.environ.get("FVTA_ATTN_ALL_ROWS") and K > 1:
    hm[:, K - 1, M:] = 0
print("valid rows: %d of %d" % (int(hm.sum().item()), N * K * T))
qm = torch.ones(N, JQ, dtype=torch.uint8, device="cuda")
op = ops.FocalAttention(N, K, T, JQ, w, 2, False)
R, d = N * K * T, w // 2
rows = torch.arange(R, device="cuda", dtype=torch.int64)
# (a) contiguous, row order
ca = [h.view(R, w)[:, i * d:(i + 1) * d].contiguous() for i in range(2)]
tab_a = torch.stack([c.data_ptr() + rows * d * 2 for c in ca]).contiguous()
# (b) hs[dir][tok][seq][d]: row (n, k, m, tok) -> sequence (n K + k) M + m, B = N K M sequences
B = N * K * M
seq, tok = rows // JT, rows % JT
slot = tok * B + seq
cb = []
for i in range(2):
    buf = torch.empty(JT * B, d, dtype=torch.bfloat16, device="cuda")
    buf[slot] = h.view(R, w)[:, i * d:(i + 1) * d]
    cb.append(buf)
tab_b = torch.stack([c.data_ptr() + slot * d * 2 for c in cb]).contiguous()
h32 = h.float()
gha = torch.randn(N, w, device="cuda", generator=g)
d_h = torch.zeros(N, K, T, w, device="cuda")
d_q = torch.zeros(N, JQ, w, device="cuda")
dW, db = torch.zeros_like(W), torch.zeros(1, device="cuda")


def timeit(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


res = {}
ref = op.forward(h32, q, hm, qm, W, b)[0].clone()
res["fp32 rows fwd"] = timeit(lambda: op.forward(h32, q, hm, qm, W, b))
res["fp32 rows bwd"] = timeit(lambda: op.backward(h32, q, hm, qm, W, b, gha, d_h, d_q, dW, db, accumulate=2))
for name, tab in (("shadow contiguous", tab_a), ("shadow hs layout", tab_b)):
    out = op.forward_shadow(tab, q, hm, qm, W, b)
    print(name, "max |h_a - fp32-row h_a| = %.3e" % (out - ref).abs().max().item())
    res[name + " fwd"] = timeit(lambda: op.forward_shadow(tab, q, hm, qm, W, b))
    res[name + " bwd"] = timeit(lambda: op.backward_shadow(tab, q, hm, qm, W, b, gha, d_h, d_q, dW, db, accumulate=2))
for k, v in res.items():
    print("%-24s %.3f ms" % (k, v))
