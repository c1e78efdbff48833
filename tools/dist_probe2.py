"""Does creating the RCCL communicator slow later kernels of the same process?  (measurement aid)
  FVTA_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
      --master-port 29514 tools/dist_probe2.py [gloo|nccl|nccl_lazy]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch.distributed as td
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs
    from fvta_memexqa_amd.trainer import Trainer

    mode = sys.argv[1] if len(sys.argv) > 1 else "nccl"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    spec = SynthSpec(**dict(CONFIGS["metric"], dense=True))
    cfg = dict(spec.cfg(), batch_size=spec.N, precision="bf16", optimizer="adadelta", init_lr=0.5)

    def build():
        model = Model(cfg, text_in=spec.text_in, img_in=spec.img_in, device=dev)
        tr = Trainer(model, cfg)
        tr.need_dx = True
        L = model.load_inputs(make_inputs(spec, rank=0), training=True)

        def step():
            model.zero_grad()
            model.forward(L)
            model.backward(L, loss_scale=1.0, need_dx=True)
            tr.opt.apply(model.params, 1.0)
        return step, model

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    step, model = build()
    print("[%s] before init_process_group:            %.2f ms/step" % (mode, timed(step)), flush=True)
    kw = {}
    if mode == "nccl":
        kw["device_id"] = dev
    td.init_process_group(backend="gloo" if mode == "gloo" else "nccl", rank=0, world_size=1, **kw)
    print("[%s] after init, same model:               %.2f ms/step" % (mode, timed(step)), flush=True)
    if mode != "gloo":
        td.all_reduce(model.params.grad)
        torch.cuda.synchronize()
        print("[%s] after first all-reduce, same model:   %.2f ms/step" % (mode, timed(step)), flush=True)
    step2, model2 = build()
    print("[%s] model built after init:               %.2f ms/step" % (mode, timed(step2)), flush=True)
    for k in sorted(os.environ):
        if k.startswith(("HSA_", "HIP_", "NCCL_", "RCCL_", "GPU_", "ROC", "AMD_", "TORCH_NCCL")):
            print("   env %s=%s" % (k, os.environ[k]))
    td.destroy_process_group()
    print("[%s] after destroy_process_group:          %.2f ms/step" % (mode, timed(step)), flush=True)


if __name__ == "__main__":
    main()
