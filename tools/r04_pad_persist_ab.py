"""What does fvta_lstm_desc.out_pads_persist save when the lengths CHANGE every step (bench.py repeats one batch, where
nothing is left to zero)?  The text cell's forward at the metric shape, ragged lengths drawn anew per step from a pool of
8 batches, persist off / on: ms per (plan + forward)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import ops
B, J, din, d = 12864, 30, 200, 512
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, J, din, device="cuda", generator=g)
k = (torch.rand(din + d, 4 * d, device="cuda", generator=g) * 2 - 1) * 0.05
b = torch.randn(4 * d, device="cuda", generator=g) * 0.1
ar = torch.arange(B, dtype=torch.int64)
pool = [torch.randint(0, J + 1, (B,), generator=torch.Generator().manual_seed(s)).to("cuda", torch.int32) for s in range(8)]
for persist in (False, True):
    for same in (True, False):
        op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                        share_fw_bw=True, precision=1, training=True, out_pads_persist=persist)
        out = torch.zeros(B, J, 2 * d, device="cuda")
        def step(i):
            op.make_plan(pool[0 if same else i % 8])
            op.forward(x, out, k, b)
        for i in range(4): step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 24
        for i in range(n): step(i)
        torch.cuda.synchronize()
        print("persist %d, %s: %.3f ms per plan + forward" % (persist, "one batch repeated" if same else "new lengths every step",
                                                              (time.perf_counter() - t0) / n * 1e3))
