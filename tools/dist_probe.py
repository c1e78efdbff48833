"""Where does the data-parallel step spend its extra time?  Run with one rank and FVTA_DIST_FORCE=1:
  FVTA_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
      --master-port 29513 tools/dist_probe.py
Prints host enqueue time and device time of a step with and without the gradient all-reduce, and of the
all-reduce alone."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from fvta_memexqa_amd import dist
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs
    from fvta_memexqa_amd.trainer import Trainer
    import torch.distributed as td

    ws, rank, local = dist.init()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    spec = SynthSpec(**dict(CONFIGS["metric"], dense=True))
    cfg = dict(spec.cfg(), batch_size=spec.N, precision="bf16", optimizer="adadelta", init_lr=0.5)
    model = Model(cfg, text_in=spec.text_in, img_in=spec.img_in, device=dev)
    tr = Trainer(model, cfg)
    tr.need_dx = True
    L = model.load_inputs(make_inputs(spec, rank=rank), training=True)
    g = model.params.grad
    print("flat gradient: %.1f MB, dist initialised: %s" % (g.numel() * 4 / 1e6, td.is_initialized()), flush=True)

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3

    def step_no_ar():
        model.zero_grad()
        model.forward(L)
        model.backward(L, loss_scale=1.0, need_dx=True)
        tr.opt.apply(model.params, 1.0)

    print("step without all-reduce: host enqueue %.2f ms, total %.2f ms" % timed(step_no_ar), flush=True)
    if td.is_initialized():
        print("all-reduce alone:        host enqueue %.2f ms, total %.2f ms" % timed(lambda: td.all_reduce(g)), flush=True)
        print("step with all-reduce:    host enqueue %.2f ms, total %.2f ms" % timed(lambda: tr.step_device(L)), flush=True)
        w = td.all_reduce(g, async_op=True)
        w.wait()
    print("step without all-reduce: host enqueue %.2f ms, total %.2f ms" % timed(step_no_ar), flush=True)
    dist.shutdown()


if __name__ == "__main__":
    if os.environ.get("PROBE_STREAM", "0") == "1":      # run everything on a non-default stream
        with torch.cuda.stream(torch.cuda.Stream()):
            main()
    else:
        main()
