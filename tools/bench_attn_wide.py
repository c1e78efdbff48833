"""Times the focal attention forward alone at BASELINE.json configs[4]'s shape (32 x 7 x 7200 x 2048, JQ = 60; diagnostics).
  python tools/bench_attn_wide.py [exact]"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import ops, _lib
lib = _lib.load()
N, K, T, JQ, w = 32, 7, 7200, 60, 2048
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(N, K, T, w, device="cuda", generator=g) * 0.5
q = torch.randn(N, JQ, w, device="cuda", generator=g) * 0.5
W = torch.randn(2 * w, device="cuda", generator=g) * 0.05
b = torch.zeros(1, device="cuda")
hm = torch.ones(N, K, T, dtype=torch.uint8, device="cuda"); hm[:, 6, 120:] = 0     # the photo stream: 120 rows
qm = torch.ones(N, JQ, dtype=torch.uint8, device="cuda")
op = ops.FocalAttention(N, K, T, JQ, w, 2, True)
if len(sys.argv) > 1 and sys.argv[1] == "exact":
    lib.fvta_attn_kernel_select(1, 0)
for _ in range(2): op.forward(h, q, hm, qm, W, b)
torch.cuda.synchronize()
lib.fvta_profile_enable(1)
n = 5
for _ in range(n): ha, _ = op.forward(h, q, hm, qm, W, b)
torch.cuda.synchronize(); lib.fvta_profile_enable(0)
ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
lib.fvta_profile_collect(4, ctypes.byref(ms), ctypes.byref(cnt))
rows = int(hm.sum().item())
gb = (rows * w + N * JQ * w + N * w) * 4 / 1e9
t = ms.value / max(cnt.value, 1)
print("%s: attention forward main kernel %.3f ms, %.2f GB algorithmic -> %.2f TB/s = %.3f of 8 TB/s; checksum %.6f"
      % ("exact" if len(sys.argv) > 1 else "fast", t, gb, gb / t, gb / t / 8.0, float(ha.double().abs().sum())))
