"""Order dependence of the RCCL-communicator slowdown (measurement aid).  No torchrun needed:
  python tools/dist_probe3.py {pg_first|hip_first|lib_first|nopg} [nsteps]"""
import os
import sys
import time

os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29515")

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch.distributed as td
    from fvta_memexqa_amd import _lib
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs
    from fvta_memexqa_amd.trainer import Trainer

    mode = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    if mode == "hip_first":
        torch.cuda.set_device(0)
        torch.zeros(1 << 20, device=dev).sum().item()
    if mode == "lib_first":
        torch.cuda.set_device(0)
        _lib.load()
        torch.zeros(1 << 20, device=dev).sum().item()
    if mode != "nopg":
        torch.cuda.set_device(0)
        td.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    spec = SynthSpec(**dict(CONFIGS["metric"], dense=True))
    cfg = dict(spec.cfg(), batch_size=spec.N, precision="bf16", optimizer="adadelta", init_lr=0.5)
    model = Model(cfg, text_in=spec.text_in, img_in=spec.img_in, device=dev)
    tr = Trainer(model, cfg)
    tr.need_dx = True
    L = model.load_inputs(make_inputs(spec, rank=0), training=True)

    def step():
        model.zero_grad()
        model.forward(L)
        model.backward(L, loss_scale=1.0, need_dx=True)
        tr.opt.apply(model.params, 1.0)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    print("[%s] %.2f ms/step (side stream concurrency ratio %.2f)" % (mode, (time.perf_counter() - t0) / nsteps * 1e3, model.side_stream_ratio), flush=True)
    if mode != "nopg":
        td.destroy_process_group()


if __name__ == "__main__":
    main()
