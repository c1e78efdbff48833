"""Achievable HBM rate of mixed read / write stream sets on this device (fvta_probe_hbm_mix): the roof the LSTM step
epilogues can be held against.  Streams of 64 MiB (past L2, inside the Infinity Cache only for the smallest sets) and
256 MiB each; median of 7 after a warm-up."""
import ctypes, statistics, sys, os
import torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
for mb in (64, 256):
    per = mb << 20
    for nr, nw in ((1, 0), (5, 0), (0, 2), (1, 1), (5, 2), (3, 4), (2, 1)):
        buf = torch.empty((nr + nw) * per, dtype=torch.uint8, device=dev).zero_()
        ts = []
        for i in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.fvta_probe_hbm_mix(buf.data_ptr(), per, nr, nw, s), "probe")
            e1.record(); e1.synchronize()
            if i: ts.append(e0.elapsed_time(e1))
        t = statistics.median(ts)
        print("streams of %3d MiB: %d read + %d written: %.0f GB/s (%.3f ms)" % (mb, nr, nw, (nr + nw) * per / t / 1e6, t))
        del buf
