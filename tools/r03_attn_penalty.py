"""Why does the context attention take 0.36-0.38 ms inside the bf16 train step and 0.33 alone?  The step with an idle gap
(one spinning wave, fvta_probe_spin) or a cache-flushing read (fvta_probe_hbm_read over a buffer larger than the
Infinity Cache) put between the encoders and the attention: the library's attn_fwd_main bracket per variant."""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import _lib
from fvta_memexqa_amd.model_v2 import Model
from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs
from fvta_memexqa_amd.trainer import Trainer
lib = _lib.load()
spec = SynthSpec(dense=True, **CONFIGS["metric"])
cfg = dict(spec.cfg(), batch_size=spec.N, precision="bf16", optimizer="adam", init_lr=0.001)
model = Model(cfg, text_in=spec.text_in, img_in=spec.img_in)
trainer = Trainer(model, cfg); trainer.need_dx = True
L = model.load_inputs(make_inputs(spec), training=True)
flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda").zero_()   # 1 GiB
sink = torch.zeros(16, dtype=torch.float32, device="cuda")
orig = model._attend
mode = {"v": "plain"}
def attend(L_, want):
    s = torch.cuda.current_stream().cuda_stream
    if mode["v"] == "spin": lib.fvta_probe_spin(1000, s)
    if mode["v"] == "spin20ms": lib.fvta_probe_spin(20000, s)
    if mode["v"] == "flush": lib.fvta_probe_hbm_read(flush.data_ptr(), flush.numel() * 4, sink.data_ptr(), s)
    if mode["v"] == "flush+spin":
        lib.fvta_probe_hbm_read(flush.data_ptr(), flush.numel() * 4, sink.data_ptr(), s); lib.fvta_probe_spin(300, s)
    return orig(L_, want)
model._attend = attend
def collect(pid):
    ms, n = ctypes.c_double(0), ctypes.c_int64(0)
    lib.fvta_profile_collect(pid, ctypes.byref(ms), ctypes.byref(n)); return ms.value, n.value
for m in ("plain", "spin", "flush", "flush+spin", "spin20ms", "plain"):
    mode["v"] = m
    for _ in range(3): trainer.step_device(L)
    torch.cuda.synchronize()
    lib.fvta_profile_enable(1)
    for _ in range(10): trainer.step_device(L)
    torch.cuda.synchronize()
    lib.fvta_profile_enable(0)
    ms, n = collect(4)
    print("%-11s attn_fwd_main %.4f ms per step" % (m, ms / 10))
