"""One-rank RCCL overhead (FVTA_DIST_FORCE=1): does the photo cell's side stream still run beside the main stream once the
process group's own stream exists?  Prints the measured concurrency ratio (1.0: concurrent, 2.0: serialised) before and
after the first collectives, and the step time with the collectives (a) as shipped, (b) with RCCL's stream created
BEFORE the model picks its side stream.
  FVTA_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 python tools/rccl_probe.py [warm]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import _lib, dist, ops
from fvta_memexqa_amd.model_v2 import Model
from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs
from fvta_memexqa_amd.trainer import Trainer
ws, rank, local = dist.init()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
lib = _lib.load()
if len(sys.argv) > 1 and sys.argv[1] == "warm" and dist.is_dist():
    t = torch.zeros(1 << 20, device=dev)
    torch.distributed.all_reduce(t)
    torch.distributed.all_reduce(t[:1000], async_op=True).wait()
    torch.cuda.synchronize()
spec = SynthSpec(**dict(CONFIGS["metric"], dense=True))
cfg = dict(spec.cfg(), batch_size=spec.N, precision="bf16", optimizer="adam", init_lr=0.001)
model = Model(cfg, text_in=spec.text_in, img_in=spec.img_in, device=dev)
tr = Trainer(model, cfg); tr.need_dx = True
L = model.load_inputs(make_inputs(spec), training=True)
main = torch.cuda.current_stream(dev)
print("picked side stream: ratio %.3f" % model.side_stream_ratio, "  re-measured %.3f" % ops.concurrency_ratio(main, model._side))
def timeit(n=20):
    for _ in range(5): tr.step_device(L)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step_device(L)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ms = timeit()
print("dist=%s step %.3f ms; side/main ratio after the steps %.3f" % (dist.is_dist(), ms, ops.concurrency_ratio(main, model._side)))
if dist.is_dist():
    # the same steps without the collectives (the process group and its stream still exist)
    import fvta_memexqa_amd.dist as D
    keep = (D.allreduce_async, D.allreduce_grads)
    D.allreduce_async = lambda x: None
    D.allreduce_grads = lambda g, e=0, w=None: 1.0
    print("   collectives stubbed out: %.3f ms" % timeit())
    D.allreduce_async, D.allreduce_grads = keep
    # only the late (blocking) one
    D.allreduce_async = lambda x: None
    print("   late all-reduce only (whole bucket): %.3f ms" % timeit())
    D.allreduce_async = keep[0]
    print("   as shipped again: %.3f ms" % timeit())
    # the round-3 form: the early bucket (scorer / attention / photo cell) reduced asynchronously from the side stream while
    # the text cell's recurrence runs, the rest after the backward
    model.early_allreduce = True
    print("   early bucket beside the recurrence (early_allreduce=True): %.3f ms" % timeit())
    model.early_allreduce = False
    print("   as shipped once more: %.3f ms" % timeit())
dist.shutdown()
