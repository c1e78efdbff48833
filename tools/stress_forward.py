"""Determinism stress: run the same forward many times, report which stage differs between runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvta_memexqa_amd.model_v2 import Model
from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs, make_params

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
spec = SynthSpec(dense=False, **dict(CONFIGS["metric"], N=8))
params, inputs = make_params(spec), make_inputs(spec)
model = Model(dict(spec.cfg(), batch_size=spec.N, precision=prec), text_in=spec.text_in, img_in=spec.img_in)
model.set_oracle_params(params)
L = model.load_inputs(inputs)
ref = None
bad = 0
for it in range(iters):
    yp = model.forward(L)
    torch.cuda.synchronize()
    cur = dict(arena=L.arena.clone(), g1=L.g1.clone(), yp=yp.clone())
    if ref is None:
        ref = cur
        continue
    for k in cur:
        d = (cur[k].float() - ref[k].float()).abs().max().item()
        if d != 0:
            bad += 1
            idx = ((cur[k].float() - ref[k].float()).abs() > 0).nonzero()
            print("iter", it, k, "maxdiff", d, "ndiff", idx.shape[0], "first", idx[0].tolist(), "last", idx[-1].tolist(), flush=True)
print("done bad=", bad)
