"""One-rank probes of the data-parallel step (measurement aid; runs on the GPU box, no torchrun needed):

  python tools/dist_probe.py {nopg|pg_first|hip_first|lib_first} [nsteps]

nopg       no process group: the single-GPU step.
pg_first   RCCL communicator created before HIP is touched by this process;
hip_first  a HIP allocation first, then the communicator;
lib_first  libfvta_hip.so loaded + a HIP allocation first, then the communicator.
For the three communicator modes the script also times the flat-gradient all-reduce alone and the step with it
(`Trainer.step_device` under FVTA_DIST_FORCE=1).  History: creating the communicator before the model moved the model's
side stream onto the main stream's hardware queue (15.7 -> 21.9 ms per step); `ops.pick_side_stream` now measures
which stream really runs beside the main one, and all four orders give the same step time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29515")):
    os.environ.setdefault(k, v)


def main():
    import torch.distributed as td
    from fvta_memexqa_amd import _lib
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs
    from fvta_memexqa_amd.trainer import Trainer

    mode = sys.argv[1] if len(sys.argv) > 1 else "nopg"
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    if mode in ("hip_first", "lib_first"):
        torch.cuda.set_device(0)
        if mode == "lib_first":
            _lib.load()
        torch.zeros(1 << 20, device=dev).sum().item()
    if mode != "nopg":
        os.environ["FVTA_DIST_FORCE"] = "1"
        torch.cuda.set_device(0)
        td.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    spec = SynthSpec(**dict(CONFIGS["metric"], dense=True))
    cfg = dict(spec.cfg(), batch_size=spec.N, precision="bf16", optimizer="adadelta", init_lr=0.5)
    model = Model(cfg, text_in=spec.text_in, img_in=spec.img_in, device=dev)
    tr = Trainer(model, cfg)
    tr.need_dx = True
    L = model.load_inputs(make_inputs(spec, rank=0), training=True)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return (t1 - t0) / nsteps * 1e3, (time.perf_counter() - t0) / nsteps * 1e3

    def step_no_ar():
        model.zero_grad()
        model.forward(L)
        model.backward(L, loss_scale=1.0, need_dx=True)
        tr.opt.apply(model.params, 1.0)

    print("[%s] step without all-reduce: host enqueue %.2f ms, total %.2f ms (side stream concurrency ratio %.2f)"
          % ((mode,) + timed(step_no_ar) + (model.side_stream_ratio,)), flush=True)
    if mode != "nopg":
        g = model.params.grad
        print("[%s] all-reduce of %.1f MB alone: host enqueue %.2f ms, total %.2f ms"
              % ((mode, g.numel() * 4 / 1e6) + timed(lambda: td.all_reduce(g))), flush=True)
        print("[%s] step with all-reduce:    host enqueue %.2f ms, total %.2f ms" % ((mode,) + timed(lambda: tr.step_device(L))),
              flush=True)
        td.destroy_process_group()


if __name__ == "__main__":
    main()
