"""Experiment: the text cell as ONE call over B sequences vs TWO concurrent calls over B/2 on two streams
(does the k-loop phase of one overlap the HBM-bound epilogue phase of the other?)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import ops
B, J, din, d = 13120, 30, 200, 512
g = torch.Generator(device="cuda").manual_seed(0)
k = (torch.rand(din + d, 4 * d, device="cuda", generator=g) * 2 - 1) * 0.05
b = torch.zeros(4 * d, device="cuda")
def make(Bn):
    x = torch.randn(Bn, J, din, device="cuda", generator=g)
    ar = torch.arange(Bn, dtype=torch.int64)
    op = ops.BiLstm(Bn, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((Bn,), J, dtype=torch.int32), 2 * d,
                    share_fw_bw=True, precision=1, training=True)
    op.make_plan(torch.full((Bn,), J))
    out = torch.empty(Bn, J, 2 * d, device="cuda")
    dout = torch.randn(Bn, J, 2 * d, device="cuda", generator=g)
    return dict(op=op, x=x, out=out, dout=dout, dx=torch.zeros_like(x), dk=torch.zeros_like(k), db=torch.zeros_like(b))
def timeit(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
full = make(B)
print("one call   fwd %.3f ms" % timeit(lambda: full["op"].forward(full["x"], full["out"], k, b)))
print("one call   bwd %.3f ms" % timeit(lambda: full["op"].backward(full["x"], full["out"], full["dout"], k, None, full["dx"], full["dk"], full["db"])))
for parts in (2, 3, 4):
    hs = [make(B // parts) for _ in range(parts)]
    ss = [torch.cuda.Stream() for _ in range(parts)]
    def run(which):
        main = torch.cuda.current_stream()
        for h, s in zip(hs, ss):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                if which == "fwd":
                    h["op"].forward(h["x"], h["out"], k, b)
                else:
                    h["op"].backward(h["x"], h["out"], h["dout"], k, None, h["dx"], h["dk"], h["db"])
        for s in ss:
            main.wait_stream(s)
    print("%d streams  fwd %.3f ms" % (parts, timeit(lambda: run("fwd"))))
    print("%d streams  bwd %.3f ms" % (parts, timeit(lambda: run("bwd"))))
