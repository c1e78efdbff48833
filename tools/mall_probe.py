"""Read rate of fvta_probe_hbm_read over buffers of different sizes, repeated passes: sizes that fit the 256 MB memory-side
(Infinity) cache but not the L2s (8 x 4 MB) show what a re-read that misses L2 costs -- the budget of any two-pass kernel."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvta_memexqa_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
sink = torch.zeros(16, device=dev)
strm = torch.cuda.current_stream().cuda_stream
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1536):
    buf = torch.empty(mb << 18, dtype=torch.float32, device=dev).zero_()
    ts = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.fvta_probe_hbm_read(buf.data_ptr(), buf.numel() * 4, sink.data_ptr(), strm)
        e1.record()
        e1.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1))
    print("%5d MB  %.1f GB/s (median of 10 back-to-back passes, %.1f us)" % (mb, buf.numel() * 4 / (statistics.median(ts) * 1e-3) / 1e9,
                                                                    statistics.median(ts) * 1e3), flush=True)
    del buf
