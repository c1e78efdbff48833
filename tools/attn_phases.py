"""Phase timeline of the 16-row attention forward kernel (FVTA_ATTN_DBG=16 | wave<<8): shader-clock stamps of one wave."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
wave = int(sys.argv[1]) if len(sys.argv) > 1 else 0
os.environ["FVTA_ATTN_DBG"] = str(16 | (wave << 8) | (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
from fvta_memexqa_amd import ops
N, K, T, JQ, w = 64, 40, 150, 30, 1024
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(N, K, T, w, device="cuda", generator=g) * 0.5
q = torch.randn(N, JQ, w, device="cuda", generator=g) * 0.5
W = torch.randn(2 * w, device="cuda", generator=g) * 0.1
b = torch.zeros(1, device="cuda")
hm = torch.ones(N, K, T, dtype=torch.uint8, device="cuda")
qm = torch.ones(N, JQ, dtype=torch.uint8, device="cuda")
op = ops.FocalAttention(N, K, T, JQ, w, 2, True)
for _ in range(3):
    op.forward(h, q, hm, qm, W, b)
torch.cuda.synchronize()
st = op.work[(32 << 20):(32 << 20) + 64 * 16 * 8].view(torch.int64).view(64, 16).cpu()
names = ["issue_next", "first/B", "mfma+rt", "bar1", "post", "bar2", "softmax+u"]
print("tile  " + "  ".join("%12s" % n for n in names) + "   total(next start - start)")
for gi in range(2, 40):
    d = [(st[gi, k + 1] - st[gi, k]).item() for k in range(7)]
    print("%4d  " % gi + "  ".join("%12d" % x for x in d) + "   %d" % (st[gi + 1, 0] - st[gi, 0]).item())
print("prologue cycles (entry -> tile loop):", (st[0, 14] - st[0, 15]).item(), " tile0 start - entry:", (st[0, 0] - st[0, 15]).item(),
      " tiles 0..49:", (st[49, 0] - st[0, 0]).item())
