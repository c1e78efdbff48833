"""Data parallelism end to end on the device: two ranks (gloo -- a single GPU cannot host two RCCL ranks), each running
the REAL model on its shard of the batch through Trainer.step_device (early gradient bucket reduced asynchronously from
the photo cell's stream, the rest after the text cell, 1/world in the optimiser).  After two steps the parameters equal a
single process trained on the whole batch: the reference's global-batch mean (model_v2.py:1090)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spec(n):
    from fvta_memexqa_amd.synth import SynthSpec
    return SynthSpec(N=n, A=2, P=3, S=2, L=5, d=32, SA=1, dense=False, text_in=12, img_in=8)


def _shard(inputs, lo, hi):
    cut = lambda st: {k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in st.items()}
    return dict(ctx=[cut(s) for s in inputs["ctx"]], q=cut(inputs["q"]), choices=cut(inputs["choices"]), y=inputs["y"][lo:hi])


def _worker(rank, ws, port, precision, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from fvta_memexqa_amd import dist
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import make_inputs, make_params
    from fvta_memexqa_amd.trainer import Trainer
    dist.init(backend="gloo")
    spec = _spec(8)
    lo, hi = dist.shard_range(spec.N, ws, rank)
    # (bf16 run: the opt-in early bucket -- reduced asynchronously from the side stream -- so that both forms stay covered)
    cfg = dict(spec.cfg(), batch_size=hi - lo, init_lr=0.5, precision=precision, early_allreduce=(precision == "bf16"))
    model = Model(cfg, text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(make_params(spec))
    tr = Trainer(model, cfg)
    tr.need_dx = True
    L = model.load_inputs(_shard(make_inputs(spec), lo, hi), training=True)
    losses = []
    for _ in range(2):
        loss = tr.step_device(L)
        losses.append(float(dist.mean_over_ranks(loss.clone()).item()))
    torch.cuda.synchronize()
    q.put((rank, model.params.flat.cpu().numpy(), losses, model.params.early_numel))
    dist.barrier()
    dist.shutdown()


def _reference_worker(precision, q):
    """one process, the whole batch"""
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import make_inputs, make_params
    from fvta_memexqa_amd.trainer import Trainer
    spec = _spec(8)
    cfg = dict(spec.cfg(), batch_size=spec.N, init_lr=0.5, precision=precision)
    model = Model(cfg, text_in=spec.text_in, img_in=spec.img_in)
    model.set_oracle_params(make_params(spec))
    tr = Trainer(model, cfg)
    tr.need_dx = True
    L = model.load_inputs(make_inputs(spec), training=True)
    losses = [float(tr.step_device(L).item()) for _ in range(2)]
    q.put((-1, model.params.flat.cpu().numpy(), losses, model.params.early_numel))


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_two_ranks_equal_one_process_on_the_whole_batch(precision):
    # every GPU user of this test is a CHILD process; the pytest process must not have initialised the GPU before it
    # starts them (conftest.py runs this file first)
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: child processes may not be exec'd from it here; "
                    "run tests/test_gpu_dist.py first or alone")
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, ws, port, precision, q)) for r in range(ws)]
    ps.append(ctx.Process(target=_reference_worker, args=(precision, q)))
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    (_, ref, ref_losses, _), r0, r1 = res
    assert 0 < r0[3] < ref.size                                       # the flat gradient has an early part and a late one
    tol = dict(rtol=2e-4, atol=2e-6) if precision == "f32" else dict(rtol=5e-2, atol=2e-3)
    assert np.array_equal(r0[1], r1[1]), "ranks must hold identical parameters"
    np.testing.assert_allclose(r0[1], ref, **tol)
    np.testing.assert_allclose(r0[2], ref_losses, rtol=1e-4 if precision == "f32" else 2e-2)
