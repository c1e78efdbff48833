"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py):
CPU: both oracle restatements reproduce them.  GPU: the HIP path reproduces them (1e-4, exact argmax)."""
import glob
import os

import numpy as np
import pytest
import torch

from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params, to_dtype, to_numpy

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "fvta_*.npz")))


def _spec(z):
    kw = {k[5:]: z[k].item() for k in z.files if k.startswith("spec_")}
    return SynthSpec(**kw)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracles_reproduce_golden(path):
    from oracle import fvta_fused as F
    from oracle import fvta_literal as L
    z = np.load(path)
    spec = _spec(z)
    params, inputs = make_params(spec), make_inputs(spec)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    of = F.fvta_forward(p64, to_dtype(inputs, torch.float64), spec.cfg())
    of["loss"].backward()
    ol = L.fvta_forward(to_numpy({k: v.double() for k, v in params.items()}), to_numpy(to_dtype(inputs, torch.float64)), spec.cfg())
    for k in ("hall", "g1_all", "att_logits", "logits", "yp", "loss"):
        np.testing.assert_allclose(ol[k], z["out_" + k], rtol=1e-10, atol=1e-12, err_msg=k)
        np.testing.assert_allclose(of[k].detach().numpy(), z["out_" + k], rtol=1e-8, atol=1e-10, err_msg=k)
    for k, v in p64.items():
        if v.grad is not None:
            np.testing.assert_allclose(v.grad.numpy(), z["grad_" + k], rtol=1e-8, atol=1e-11, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_path_reproduces_golden(path):
    from fvta_memexqa_amd.model_v2 import Model
    z = np.load(path)
    spec = _spec(z)
    params, inputs = make_params(spec), make_inputs(spec)
    model = Model(dict(spec.cfg(), batch_size=spec.N), text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(params)
    L = model.load_inputs(inputs, training=True)
    model.zero_grad()
    yp = model.forward(L, want_logits=True).cpu().numpy()
    model.backward(L, need_dx=False)
    np.testing.assert_allclose(yp, z["out_yp"], rtol=1e-4, atol=1e-6)
    assert (yp.argmax(1) == z["out_yp"].argmax(1)).all()
    np.testing.assert_allclose(model.loss.cpu().numpy()[0], z["out_loss"], rtol=1e-4)
    np.testing.assert_allclose(model.att_logits.cpu().numpy(), z["out_att_logits"], rtol=1e-4, atol=2e-5)
    grads = model.get_oracle_grads()
    for k in z.files:
        if k.startswith("grad_"):
            ref = z[k]
            np.testing.assert_allclose(grads[k[5:]].reshape(ref.shape), ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k)
