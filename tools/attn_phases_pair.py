"""Phase timeline of attn_fwd_pair16 (diagnostics build, FVTA_ATTN_DBG = 16 | wave << 8): shader-clock stamps of one wave
of workgroup 0 over its first 32 rounds.  Columns: first blocks landed + first MFMA group | rest of the score loop |
row terms + wait(partner consumed) + publish | wait(partner published) | max / arg-max / softmax | weighted sum + refill."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
wave = int(sys.argv[1]) if len(sys.argv) > 1 else 0
os.environ["FVTA_ATTN_DBG"] = str(16 | (wave << 8))
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
st = op.work[(32 << 20):(32 << 20) + 32 * 16 * 8].view(torch.int64).view(32, 16).cpu()
names = ["first", "score", "rt+done+pub", "wait pub", "softmax", "wsum"]
print("wave %d" % wave)
print("round " + "  ".join("%12s" % n for n in names) + "   total")
tot = [0] * 7
for gi in range(1, 24):
    d = [(st[gi, k + 1] - st[gi, k]).item() for k in range(6)]
    d.append((st[gi + 1, 0] - st[gi, 0]).item())
    tot = [a + b for a, b in zip(tot, d)]
    print("%4d  " % gi + "  ".join("%12d" % x for x in d[:6]) + "   %d" % d[6])
print("mean  " + "  ".join("%12d" % (x // 23) for x in tot[:6]) + "   %d" % (tot[6] // 23))
print("entry -> round 0:", (st[0, 0] - st[0, 15]).item())
