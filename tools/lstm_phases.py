"""Phase timeline of one workgroup of lstm_step_fwd_bf16 at step t=5 (FVTA_DEBUG_SKIP = 32768 | wg << 16)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.getcwd())
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd"
if bwd:
    os.environ["FVTA_LSTM_STAMP_BWD"] = str(wg)
else:
    os.environ["FVTA_DEBUG_SKIP"] = str(32768 | (wg << 16))
from fvta_memexqa_amd import ops, _lib
B, J, din, d = 12864, 30, 200, 512
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, J, din, device="cuda", generator=g)
lens = torch.full((B,), J)
k = (torch.rand(din + d, 4 * d, device="cuda", generator=g) * 2 - 1) * 0.05
b = torch.zeros(4 * d, device="cuda")
ar = torch.arange(B, dtype=torch.int64)
op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                share_fw_bw=True, precision=1, training=True)
op.make_plan(lens)
out = torch.empty(B, J, 2 * d, device="cuda")
for _ in range(2):
    op.forward(x, out, k, b)
if bwd:
    dout = torch.randn(B, J, 2 * d, device="cuda", generator=g)
    dx = torch.zeros_like(x); dk = torch.zeros_like(k); db = torch.zeros_like(b)
    op.backward(x, out, dout, k, None, dx, dk, db)
torch.cuda.synchronize()
lib = _lib.load()
def rd(i):
    ms = ctypes.c_double(); n = ctypes.c_int64()
    lib.fvta_profile_collect(100000 + i, ctypes.byref(ms), ctypes.byref(n))
    return n.value
t0, t1, t2, nt = rd(0), rd(1), rd(2), rd(3)
print("wg", wg, "k-tiles", nt, " k-loop cycles", t1 - t0, " epilogue cycles", t2 - t1)
rows = []
for t in range(nt):
    s = [rd(8 + 4 * t + i) for i in range(4)]
    nxt = rd(8 + 4 * (t + 1)) if t + 1 < nt else t1
    rows.append((s[1] - s[0], s[2] - s[1], s[3] - s[2], nxt - s[3]))
print("tile   wait_vmcnt  barrier+issue  mfma_issue  tail")
for t, r in enumerate(rows[:24]):
    print("%4d %10d %12d %11d %6d" % ((t,) + r))
import statistics
print("mean per k-tile:", [round(statistics.mean(r[i] for r in rows)) for i in range(4)], "sum", round(sum(sum(r) for r in rows) / len(rows)))
