"""Experiment: does replaying the train step as one HIP graph (torch.cuda.CUDAGraph) shorten it?"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import _lib
from fvta_memexqa_amd.model_v2 import Model
from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs
from fvta_memexqa_amd.trainer import Trainer
kw = dict(CONFIGS["metric"], dense=True)
if len(sys.argv) > 1 and sys.argv[1] == "readme":     # README.MD:219-226 training sizes: batch 6, 4 albums x 8 photos, 8-word texts, hidden 50
    kw = dict(N=6, A=4, P=8, S=5, L=8, d=50, dense=False)
spec = SynthSpec(**kw)
cfg = dict(spec.cfg(), batch_size=spec.N, precision="bf16" if kw["d"] >= 512 else "f32", optimizer="adam", init_lr=0.001)
model = Model(cfg, text_in=spec.text_in, img_in=spec.img_in)
tr = Trainer(model, cfg); tr.need_dx = True
L = model.load_inputs(make_inputs(spec), training=True)
def fb():
    model.zero_grad(); model.forward(L); model.backward(L, need_dx=True)
def timeit(f, n=20):
    """(ms per call, host enqueue ms per call)"""
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / n * 1e3, 3), round(host / n * 1e3, 3)
print("eager fwd+bwd (ms, host enqueue ms)", timeit(fb))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): fb()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        fb()
    print("graph fwd+bwd (ms, host enqueue ms)", timeit(g.replay))
except Exception as e:
    print("capture failed:", repr(e)[:400])
