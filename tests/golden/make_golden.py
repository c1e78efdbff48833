"""Generates tests/golden/*.npz.

The reference (Python-2 / TensorFlow-1) cannot be imported or run here and ships no vectors
("parity unpinned", oracle/fvta_literal.py header), so these fixtures are produced by the repo's
own CPU oracle in fp64 -- the literal NumPy restatement for the forward values and the fused torch
restatement (autograd) for the gradients, after checking that the two agree.  They freeze today's
oracle outputs so that neither the oracle nor the HIP path can drift silently.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params, to_dtype, to_numpy  # noqa: E402
from oracle import fvta_fused as F  # noqa: E402
from oracle import fvta_literal as L  # noqa: E402

CASES = {
    "fvta_simi2_tanh_qatt": dict(N=3, A=2, P=3, S=2, L=5, d=32, SA=1, dense=False, simiMatrix=2, add_tanh=True,
                                 use_question_att=True, text_in=12, img_in=8),
    "fvta_simi1_lq_unshared": dict(N=2, A=1, P=4, S=1, L=6, d=32, SA=2, dense=False, simiMatrix=1, add_tanh=False,
                                   use_question_att=False, share_fw_bw=False, text_in=8, img_in=8),
}


def build(name, kw):
    spec = SynthSpec(**kw)
    params, inputs, cfg = make_params(spec), make_inputs(spec), spec.cfg()
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    of = F.fvta_forward(p64, to_dtype(inputs, torch.float64), cfg)
    of["loss"].backward()
    ol = L.fvta_forward(to_numpy({k: v.double() for k, v in params.items()}), to_numpy(to_dtype(inputs, torch.float64)), cfg)
    for k in ("hall", "g1_all", "logits", "yp", "loss"):
        np.testing.assert_allclose(of[k].detach().numpy(), ol[k], rtol=1e-9, atol=1e-11, err_msg=k)
    out = {"spec_" + k: np.asarray(v) for k, v in kw.items()}
    for k in ("hall", "hq", "lchoices", "g1_all", "gq", "att_logits", "logits", "yp", "loss"):
        out["out_" + k] = np.asarray(ol[k], np.float64)
    for k, v in p64.items():
        if v.grad is not None:
            out["grad_" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if k.startswith("out_")})


if __name__ == "__main__":
    for n, kw in CASES.items():
        build(n, kw)
