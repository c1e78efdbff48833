"""Does compute-bound work on a CU-MASKED stream run beside the text cell's backward recurrence for free?
lstm_bwd_fused_bf16's grid at the metric shape is 208 workgroups of one CU each (52 row tiles x 2 column tiles x 2 directions):
48 of the 256 CUs have nothing to do for 4.1 ms of the 12 ms step.  A stream created with hipExtStreamCreateWithCUMask
(mask bit i -> XCD i % 8, shader engine (i / 8) % 4: tools/probes/cumask_probe.hip) confines a second kernel to those CUs.
Here: the backward of the text cell alone / a bf16 K-major GEMM (the weight gradient's shape) alone on NCU masked CUs / both.
usage: python tools/cumask_overlap.py [ncu=48]"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import ops
ncu = int(sys.argv[1]) if len(sys.argv) > 1 else 48
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
mask = (ctypes.c_uint32 * 8)()
for i in range(ncu):
    mask[i // 32] |= 1 << (i % 32)
sp = ctypes.c_void_p()
rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), 8, mask)
assert rc == 0, rc
side = torch.cuda.ExternalStream(sp.value)
B, J, din, d = 13120, 30, 200, 512
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, J, din, device="cuda", generator=g)
lens = torch.full((B,), J)
k = (torch.rand(din + d, 4 * d, device="cuda", generator=g) * 2 - 1) * 0.05
b = torch.randn(4 * d, device="cuda", generator=g) * 0.1
ar = torch.arange(B, dtype=torch.int64)
op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                share_fw_bw=True, precision=1, training=True, dx_overwrite=True)
op.make_plan(lens)
out = torch.zeros(B, J, 2 * d, device="cuda")
op.forward(x, out, k, b)
d_out = torch.randn(B, J, 2 * d, device="cuda", generator=g)
dx, dk, db = torch.zeros_like(x), torch.zeros_like(k), torch.zeros_like(b)
KK, M, N = 65536, 1024, 2048
A = torch.randn(KK, M, device="cuda", generator=g)
Bm = torch.randn(KK, N, device="cuda", generator=g)
nge = 3


def bwd():
    op.backward(x, out, d_out, k, None, dx, dk, db)


def gemms():
    for _ in range(nge):
        ops.test_gemm(A, Bm, 2, precision=1)


def timed(main_f, side_f, n=4):
    tm, ts = [], []
    for _ in range(n + 1):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if side_f:
            with torch.cuda.stream(side):
                e[2].record()
                side_f()
                e[3].record()
        if main_f:
            e[0].record()
            main_f()
            e[1].record()
        torch.cuda.synchronize()
        tm.append(e[0].elapsed_time(e[1]) if main_f else 0.0)
        ts.append(e[2].elapsed_time(e[3]) if side_f else 0.0)
    return min(tm[1:]), min(ts[1:])


gemms(); bwd(); torch.cuda.synchronize()
t0 = time.perf_counter(); gemms(); torch.cuda.synchronize(); full = (time.perf_counter() - t0) * 1e3
print("backward alone              : %.3f ms" % timed(bwd, None)[0])
print("%d GEMMs alone, whole chip   : %.3f ms" % (nge, full))
print("%d GEMMs alone, %d masked CUs: %.3f ms" % (nge, ncu, timed(None, gemms)[1]))
m, s = timed(bwd, gemms)
print("both: backward %.3f ms, GEMMs on %d masked CUs %.3f ms" % (m, ncu, s))
