"""Determinism stress of the focal-attention forward alone (same inputs, many launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvta_memexqa_amd import ops

N, K, T, JQ, w = 8, 40, 150, 30, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ragged = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = torch.Generator().manual_seed(3)
dev = "cuda:0"
h = (torch.randn(N, K, T, w, generator=g) * 0.5).to(dev)
q = (torch.randn(N, JQ, w, generator=g) * 0.5).to(dev)
hm = (torch.rand(N, K, T, generator=g) < (0.6 if ragged else 2.0)).to(torch.uint8).to(dev)
qm = torch.ones(N, JQ, dtype=torch.uint8).to(dev)
h = h * hm.unsqueeze(-1).float()
W = (torch.randn(2 * w, 1, generator=g) * 0.05).to(dev)
b = torch.zeros(1).to(dev)
att = ops.FocalAttention(N, K, T, JQ, w, 2, True)
ref = None
bad = 0
for it in range(iters):
    ha, lg = att.forward(h, q, hm, qm, W, b, True)
    torch.cuda.synchronize()
    if ref is None:
        ref = (ha.clone(), lg.clone())
        continue
    d = (ha - ref[0]).abs().max().item()
    dl = (lg - ref[1]).abs()
    if d != 0 or dl.max().item() != 0:
        bad += 1
        idx = (dl > 0).nonzero()
        rows = idx[:, :3].unique(dim=0)
        r0 = rows[0].tolist()
        dd = (lg - ref[1])[r0[0], r0[1], r0[2]]
        print("rows", rows.shape[0], "row0", r0, "rank", int(hm[r0[0], r0[1], :r0[2]].sum()), "cnt", int(hm[r0[0], r0[1]].sum()),
              "dj", [round(x, 4) for x in dd.tolist()][:12])
        ks = {}
        for r in rows.tolist():
            ks.setdefault((r[0], r[1]), []).append(int(hm[r[0], r[1], :r[2]].sum()))
        print("  per (n,k) ranks:", [(k, v[:6], len(v)) for k, v in list(ks.items())[:6]])
        print("iter", it, "ha diff", d, "logit diff", dl.max().item(), "n", idx.shape[0],
              "first", idx[:3].tolist(), "js", sorted(set(idx[:, 3].tolist()))[:40], flush=True)
print("done bad=", bad)
