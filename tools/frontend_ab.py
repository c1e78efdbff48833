"""The embedding front-end's kernels alone at the metric shape (385,920 tokens of 16 characters, 2,560 photos):
HIP-event time per call of the char-CNN forward / backward and the photo transform forward / backward.
  python tools/frontend_ab.py [reps [char_emb_size]]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from fvta_memexqa_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator().manual_seed(3)
ntok, W, cd, cw, wd, VW, VF, VC = 12864 * 30, 16, (int(sys.argv[2]) if len(sys.argv) > 2 else 8), 100, 100, 400, 20000, 100   # argv[2]: char_emb_size (100: the published flag set)
cu = lambda t: t.cuda().contiguous()
ids = cu(torch.randint(0, VW + VF, (ntok,), generator=g, dtype=torch.int32))
ch = cu(torch.randint(0, VC, (ntok, W), generator=g, dtype=torch.int32))
stride = cw + wd + 24
tok_off = cu(torch.arange(ntok, dtype=torch.int64) * stride)
x = torch.zeros(ntok * stride, device="cuda")
dx = torch.randn(ntok * stride, device="cuda")
we, fe = cu(torch.randn(VW, wd, generator=g)), cu(torch.randn(VF, wd, generator=g))
ce, fl, bi = cu(torch.randn(VC, cd, generator=g)), cu(torch.randn(5, cd, cw, generator=g) * 0.3), cu(torch.randn(cw, generator=g))
op = ops.TokenEmbed(ntok, W, cd, cw, wd, VW, VW + VF, VC)
dwe, dce = torch.zeros(VW, wd, device="cuda"), torch.zeros(VC, cd, device="cuda")
dfl, dbi = torch.zeros(5, cd, cw, device="cuda"), torch.zeros(cw, device="cuda")

M, idim, tdim, VI = 2560, 2537, 100, 3000
feat = cu(torch.randn(VI, idim, generator=g))
Wi, b_i = cu(torch.randn(idim, tdim, generator=g) * 0.02), cu(torch.randn(tdim, generator=g))
pidx = cu(torch.randint(0, VI, (M,), generator=g, dtype=torch.int32))
roff = cu(torch.arange(M, dtype=torch.int64) * 128)
xi, dxi = torch.zeros(M * 128, device="cuda"), torch.randn(M * 128, device="cuda")
dWi, dbi2 = torch.zeros(idim, tdim, device="cuda"), torch.zeros(tdim, device="cuda")
im = ops.ImageTrans(M, idim, tdim, True)

def timed(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:16s} {e0.elapsed_time(e1) / reps:8.3f} ms", flush=True)

timed("embed_fwd", lambda: op.forward(ids, ch, tok_off, we, fe, ce, fl, bi, x))
timed("embed_bwd", lambda: op.backward(ids, ch, tok_off, ce, fl, dx, dwe, dce, dfl, dbi))
timed("img_fwd", lambda: im.forward(pidx, roff, feat, Wi, b_i, xi))
timed("img_bwd", lambda: im.backward(pidx, roff, feat, xi, dxi, dWi, dbi2))
