"""Times the text-cell bi-LSTM forward / backward calls alone at the metric shape (diagnostics)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import ops
prec = 1 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else 0
B, J, din, d = 12864, 30, 200, 512
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, J, din, device="cuda", generator=g)
lens = torch.full((B,), J)
k = (torch.rand(din + d, 4 * d, device="cuda", generator=g) * 2 - 1) * 0.05
b = torch.zeros(4 * d, device="cuda")
ar = torch.arange(B, dtype=torch.int64)
op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                share_fw_bw=True, precision=prec, training=True)
op.make_plan(lens)
out = torch.empty(B, J, 2 * d, device="cuda")
dout = torch.randn(B, J, 2 * d, device="cuda", generator=g)
dx = torch.zeros_like(x); dk = torch.zeros_like(k); db = torch.zeros_like(b)
def timeit(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("fwd ms", round(timeit(lambda: op.forward(x, out, k, b)), 3), "dbg", os.environ.get("FVTA_DEBUG_SKIP"))
if len(sys.argv) > 2:
    print("bwd ms", round(timeit(lambda: op.backward(x, out, dout, k, None, dx, dk, db)), 3))
