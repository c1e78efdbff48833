"""Times the text-cell bi-LSTM backward STEP launches alone (the library's HIP-event bracket FVTA_PROF_LSTM_STEP_BWD) at
the metric shape, plus dx and dW, for the library named by FVTA_LIB_PATH / the kernel set of FVTA_LSTM_WREG.
  python tools/lstm_bwd_ab.py [B J din d [ragged]]"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import _lib, ops
B, J, din, d = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (12864, 30, 200, 512)))
dense = not (len(sys.argv) > 5 and sys.argv[5] == "ragged")
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, J, din, device="cuda", generator=g)
lens = torch.full((B,), J) if dense else torch.randint(0, J + 1, (B,), generator=torch.Generator().manual_seed(1))
k = (torch.rand(din + d, 4 * d, device="cuda", generator=g) * 2 - 1) * 0.05
b = torch.randn(4 * d, device="cuda", generator=g) * 0.1
ar = torch.arange(B, dtype=torch.int64)
op = ops.BiLstm(B, J, din, d, ar * J * din, ar * J * 2 * d, torch.full((B,), J, dtype=torch.int32), 2 * d,
                share_fw_bw=True, precision=int(os.environ.get("FVTA_AB_PREC", "1")), training=True,
                dx_overwrite=bool(os.environ.get("FVTA_AB_DXOW")))   # FVTA_AB_DXOW=1: backward() writes dx (the model's setting)
op.make_plan(lens)
out = torch.zeros(B, J, 2 * d, device="cuda")
dout = torch.randn(B, J, 2 * d, device="cuda", generator=g)
dx = torch.zeros_like(x); dk = torch.zeros_like(k); db = torch.zeros_like(b)
op.forward(x, out, k, b)
side = ops.pick_side_stream(torch.device("cuda", 0))[0] if os.environ.get("FVTA_AB_SIDE") else None   # a second stream: the split step
run = lambda: op.backward(x, out, dout, k, None, dx, dk, db, side_stream=side)
for _ in range(2): run()
torch.cuda.synchronize()
lib.fvta_profile_enable(1)
n = 5
t0 = time.perf_counter()
for _ in range(n): run()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
lib.fvta_profile_enable(0)
def collect(pid):
    ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
    lib.fvta_profile_collect(pid, ctypes.byref(ms), ctypes.byref(cnt))
    return ms.value / n, cnt.value // n
s_ms, s_n = collect(2)
print("lib=%s WREG=%s: step %.3f ms (%d launches, %.1f us each)  dx %.3f  dW %.3f  wall %.3f  finite %s" % (
    os.path.basename(os.environ.get("FVTA_LIB_PATH", "product")), os.environ.get("FVTA_LSTM_WREG", "-"), s_ms, s_n,
    1e3 * s_ms / max(1, s_n), collect(6)[0], collect(3)[0], wall, bool(torch.isfinite(dk).all())))
# (dk accumulates over the 7 calls above; the checksum identifies the weight-gradient kernel's result bit by bit)
print("dk sum %.10e  abs %.10e  db sum %.10e" % (dk.double().sum().item(), dk.double().abs().sum().item(), db.double().sum().item()))
print("dx sum %.10e  abs %.10e" % (dx.double().sum().item(), dx.double().abs().sum().item()))
if os.environ.get("FVTA_AB_SAVE"): torch.save((dk.cpu(), dx.cpu()), "/tmp/lstm_bwd_ab_dk_%s.pt" % os.environ["FVTA_AB_SAVE"])
if os.environ.get("FVTA_AB_CMP"):
    dk0, dx0 = torch.load("/tmp/lstm_bwd_ab_dk_%s.pt" % os.environ["FVTA_AB_CMP"])
    print("against %s: dk bitwise equal %s, dx bitwise equal %s, max |dx diff| %.3e (max |dx| %.3f)" % (
        os.environ["FVTA_AB_CMP"], torch.equal(dk.cpu(), dk0), torch.equal(dx.cpu(), dx0), (dx.cpu() - dx0).abs().max().item(), dx0.abs().max().item()))
