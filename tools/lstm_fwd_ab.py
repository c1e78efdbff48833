"""A/B of the weights-in-registers forward step kernel against the tiled one (run twice: FVTA_LSTM_WREG=0 / unset):
text-cell bi-LSTM forward at the metric shape, output checksums + time; `FVTA_LSTM_WREG=0` selects the tiled kernel."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import ops
B, J, din, d = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (12864, 30, 200, 512)))
dense = not (len(sys.argv) > 5 and sys.argv[5] == "ragged")
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, J, din, device="cuda", generator=g)
lens = torch.full((B,), J) if dense else torch.randint(0, J + 1, (B,), generator=torch.Generator().manual_seed(1))
k = (torch.rand(din + d, 4 * d, device="cuda", generator=g) * 2 - 1) * 0.05
b = torch.randn(4 * d, device="cuda", generator=g) * 0.1
ar = torch.arange(B, dtype=torch.int64)
op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                share_fw_bw=True, precision=int(os.environ.get("FVTA_AB_PREC", "1")), training=True,
                out_skip=B * J * 2 * d if os.environ.get("FVTA_AB_SKIP") else 0)   # FVTA_AB_SKIP=1: no fp32 rows stored (shadow rows)
op.make_plan(lens)
out = torch.zeros(B, J, 2 * d, device="cuda")
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ms = timeit(lambda: op.forward(x, out, k, b))
tag = os.environ.get("FVTA_LSTM_WREG", "1")
print("WREG=%s fwd %.3f ms  sum %.6f  abs %.6f  finite %s" % (tag, ms, out.double().sum().item(), out.double().abs().sum().item(),
      bool(torch.isfinite(out).all())))
if os.environ.get("FVTA_AB_SAVE"): torch.save(out.cpu(), "/tmp/lstm_fwd_ab_out_%s.pt" % tag)
other = "/tmp/lstm_fwd_ab_out_%s.pt" % ("0" if tag != "0" else "1")
if os.path.exists(other):
    o2 = torch.load(other)
    diff = (out.cpu() - o2).abs()
    print("max |wreg - tiled| = %.3e  (mean %.3e, max |out| %.3f)" % (diff.max().item(), diff.mean().item(), o2.abs().max().item()))
