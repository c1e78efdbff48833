"""Which parts of the train step run at the socket's power limit?  Loops ONE op of the text cell at the metric shape for a
couple of seconds each -- forward recurrence, backward (recurrence + dx + dW), weight-gradient-heavy and attention loops -- while
a thread samples `rocm-smi --showclocks --showpower`, and prints the median shader clock / socket power per phase.
usage: python tools/power_by_kernel.py [seconds per phase = 2.5]"""
import os, re, subprocess, sys, threading, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import ops

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
samples, phase, stop = [], ["idle"], [False]


def sampler():
    pat = re.compile(r"sclk clock level: \S+ \((\d+)Mhz\).*?Power \(W\): ([\d.]+)", re.S)
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            continue
        m = pat.search(out)
        if m:
            samples.append((phase[0], int(m.group(1)), float(m.group(2))))


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
dout = torch.randn(B, J, 2 * d, device="cuda", generator=g)
dx, dk, db = torch.zeros_like(x), torch.zeros_like(k), torch.zeros_like(b)
op.forward(x, out, k, b)
KK, M, N = 131072, 1024, 2048
A = torch.randn(KK, M, device="cuda", generator=g)
Bm = torch.randn(KK, N, device="cuda", generator=g)
big = torch.empty(1 << 28, device="cuda")  # 1 GiB of floats: an HBM stream


def loop(name, f):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); one = time.perf_counter() - t0
    n = max(3, int(secs / max(one, 1e-4)))
    time.sleep(1.0)          # (let the governor fall back to idle between phases)
    phase[0] = name
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    phase[0] = "idle"
    return one * 1e3, el / n * 1e3


th = threading.Thread(target=sampler, daemon=True); th.start()
res = {}
res["lstm forward (30 steps)"] = loop("fwd", lambda: op.forward(x, out, k, b))
res["lstm backward (30 steps + dx + dW)"] = loop("bwd", lambda: op.backward(x, out, dout, k, None, dx, dk, db))
res["bf16 k-major GEMM 131072 x 1024 x 2048 (the dW shape)"] = loop("gemm", lambda: ops.test_gemm(A, Bm, 2, precision=1))
res["HBM stream: 1 GiB fill"] = loop("fill", lambda: big.fill_(1.0))
stop[0] = True; th.join(timeout=6)
import statistics as st
for name, tag in (("idle", "idle"), ("lstm forward (30 steps)", "fwd"), ("lstm backward (30 steps + dx + dW)", "bwd"),
                  ("bf16 k-major GEMM 131072 x 1024 x 2048 (the dW shape)", "gemm"), ("HBM stream: 1 GiB fill", "fill")):
    ss = [s for s in samples if s[0] == tag]
    ss = ss[1:-1] if len(ss) > 4 else ss   # (the samples at a phase's edges straddle it)
    if not ss: print("%-55s no samples" % name); continue
    ms = res.get(name, (0, 0))
    print("%-55s %3d samples  sclk median %4d MHz (min %4d)  power median %6.0f W (max %6.0f)   %.3f ms alone / %.3f ms in the loop" % (
        name, len(ss), st.median(s[1] for s in ss), min(s[1] for s in ss), st.median(s[2] for s in ss), max(s[2] for s in ss), ms[0], ms[1]))
