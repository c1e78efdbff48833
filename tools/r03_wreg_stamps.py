"""Where a workgroup of lstm_fwd_wreg_bf16 spends its cycles (library built with -DFVTA_DIAG -DFVTA_WREG_STAMP):
FVTA_DEBUG_SKIP=<workgroup> selects the stamped workgroup (wave 0, step t = 5)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.getcwd())
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
os.environ["FVTA_DEBUG_SKIP"] = str(wg)
from fvta_memexqa_amd import ops, _lib
B, J, din, d = 12864, 30, 200, 512
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, J, din, device="cuda", generator=g)
lens = torch.full((B,), J)
k = (torch.rand(din + d, 4 * d, device="cuda", generator=g) * 2 - 1) * 0.05
b = torch.zeros(4 * d, device="cuda")
ar = torch.arange(B, dtype=torch.int64)
op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                share_fw_bw=True, precision=1, training=True)
op.make_plan(lens)
out = torch.empty(B, J, 2 * d, device="cuda")
for _ in range(3):
    op.forward(x, out, k, b)
torch.cuda.synchronize()
lib = _lib.load()
def rd(i):
    ms = ctypes.c_double(); n = ctypes.c_int64()
    lib.fvta_profile_collect(200000 + i, ctypes.byref(ms), ctypes.byref(n))
    return n.value
t0, t1, t2, nt, w, bar, slab, n, iss, stg, rows = [rd(i) for i in range(11)]
print("wg %d: tiles %d, total %d cycles (prologue %d), per tile %d; hand-overs %d: vmcnt wait %d (%.0f each), barrier %d (%.0f each); slab write %d (%.0f per tile)"
      % (wg, nt, t2 - t0, t1 - t0, (t2 - t1) / max(nt, 1), n, w, w / max(n, 1), bar, bar / max(n, 1), slab, slab / max(nt, 1)))
print("   per tile: DMA issue %.0f, gate stages %.0f, row loads %.0f" % (iss / max(nt, 1), stg / max(nt, 1), rows / max(nt, 1)))
