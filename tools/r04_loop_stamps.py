"""Where do a k-tile's cycles go in the tiled kernels' k-loops?  Needs a -DFVTA_LOOP_STAMP build of lstm_bf16.hip
(tools/r04_build_variants.sh with SRC=lstm_bf16 stamp:-DFVTA_LOOP_STAMP, loaded through FVTA_LIB_PATH): every wave of one
workgroup sums the cycles it spends waiting for the k-tile's DMA (vmcnt), at the barrier, issuing the next refill, and in the
fragment reads + MFMAs.  Prints them for dx (the last tiled launch of a backward call is dW: its stamps print too).
  FVTA_LIB_PATH=.../libfvta_hip_stamp.so python tools/r04_loop_stamps.py [dx|dw]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import _lib, ops
which = sys.argv[1] if len(sys.argv) > 1 else "dx"
B, J, din, d = 12864, 30, 200, 512
lib = _lib.load()
raw = ctypes.CDLL(os.environ["FVTA_LIB_PATH"])
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, J, din, device="cuda", generator=g)
k = (torch.rand(din + d, 4 * d, device="cuda", generator=g) * 2 - 1) * 0.05
b = torch.randn(4 * d, device="cuda", generator=g) * 0.1
ar = torch.arange(B, dtype=torch.int64)
op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                share_fw_bw=True, precision=1, training=True)
op.make_plan(torch.full((B,), J))
out = torch.zeros(B, J, 2 * d, device="cuda")
dout = torch.randn(B, J, 2 * d, device="cuda", generator=g)
dx = torch.zeros_like(x); dk = torch.zeros_like(k); db = torch.zeros_like(b)
op.forward(x, out, k, b)
for _ in range(3):
    op.backward(x, out, dout, k, None, dx if which != "nodx" else None, dk, db)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert raw.fvta_debug_loop_stamps(buf) == 0
print("last tiled launch (the weight gradient unless the library was built to skip it): cycles per wave, summed over its k-tiles")
for w in range(8):
    v = [buf[4 * w + i] for i in range(4)]
    print("wave %d: wait %8d  barrier %8d  issue %8d  compute %8d  total %8d" % (w, v[0], v[1], v[2], v[3], sum(v)))
