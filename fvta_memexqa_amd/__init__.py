"""fvta_memexqa_amd -- MI355X-native (gfx950) FVTA hot path.

Host side (this package) mirrors the reference's model_v2.py (FVTA) / model.py
(soft-attention baselines) / trainer.py / tester.py surface; all arithmetic runs in hand-written HIP kernels behind the
C-ABI declared in include/fvta_hip.h (built into csrc/libfvta_hip.so).
There is NO CPU fallback: ops raise if the library or a GPU is missing.
"""
__version__ = "0.1.0"
