"""The N>1 path on CPU: world_size-2 gloo processes exercise exactly the host logic bench.py /
Trainer use under data parallelism -- contiguous QA-pair shards, ONE sum all-reduce of the flat
gradient bucket, the 1/world factor handed to the optimiser, max-over-ranks timing."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from fvta_memexqa_amd import dist
    w, r, _ = dist.init(backend="gloo")
    assert (w, r) == (ws, rank) and dist.is_dist()
    lo, hi = dist.shard_range(512 // 4, ws, rank)                  # 128 QA pairs over 2 ranks
    # a "flat gradient bucket": rank-local mean gradient of its shard
    g_full = torch.arange(128 * 10, dtype=torch.float32).reshape(128, 10)
    flat = g_full[lo:hi].mean(0).clone()
    scale = dist.allreduce_grads(flat)
    t = dist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    dist.barrier()
    q.put((rank, lo, hi, (flat * scale).tolist(), scale, t))   # plain lists: no shared-memory handles


def test_two_rank_gloo_data_parallel_equals_global_batch_mean():
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    g_full = torch.arange(128 * 10, dtype=torch.float32).reshape(128, 10)
    assert [(r[1], r[2]) for r in res] == [(0, 64), (64, 128)]
    for r in res:
        assert r[4] == 0.5 and r[5] == 2.0
        torch.testing.assert_close(torch.tensor(r[3]), g_full.mean(0))           # == the reference's global-batch mean


class _StubParams:
    """what Trainer.step_device touches of model_v2.ParamStore: flat / grad buffers and the early-bucket boundary"""

    def __init__(self, n, early):
        self.flat = torch.zeros(n)
        self.grad = torch.zeros(n)
        self.early_numel = early


class _StubModel:
    """forward/backward of a least-squares toy whose gradient is the MEAN over the rank's shard (as the model's loss is,
    model_v2.py:1090); backward starts the early bucket's all-reduce the way Model.backward does"""

    def __init__(self, x, y, early):
        self.x, self.y = x, y
        self.params = _StubParams(x.shape[1], early)
        self.global_step, self.loss, self.early_work = 0, None, None

    def zero_grad(self):
        self.params.grad.zero_()

    def forward(self, layout):
        self.res = self.x @ self.params.flat - self.y
        self.loss = (self.res ** 2).mean().reshape(1) * 0.5

    def backward(self, layout, loss_scale=1.0, need_dx=False):
        from fvta_memexqa_amd import dist
        self.params.grad += loss_scale * (self.x.t() @ self.res) / self.x.shape[0]
        self.early_work = dist.allreduce_async(self.params.grad[:self.params.early_numel])


class _Sgd:
    def __init__(self, lr):
        self.lr = lr

    def apply(self, params, grad_scale):
        params.flat -= self.lr * grad_scale * params.grad


def _trainer_worker(rank, ws, port, q, early):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from fvta_memexqa_amd import dist
    from fvta_memexqa_amd.trainer import Trainer
    dist.init(backend="gloo")
    g = torch.Generator().manual_seed(3)
    X, Y = torch.randn(16, 6, generator=g), torch.randn(16, generator=g)
    lo, hi = dist.shard_range(16, ws, rank)
    model = _StubModel(X[lo:hi], Y[lo:hi], early)
    tr = Trainer(model, dict(init_lr=0.1))
    tr.opt = _Sgd(0.1)                              # the product optimisers are HIP kernels; the reduce path is the real one
    losses = []
    for _ in range(3):
        loss = tr.step_device(None)
        losses.append(float(dist.mean_over_ranks(loss.clone())))
    q.put((rank, model.params.flat.tolist(), losses, model.global_step))


@pytest.mark.parametrize("early", [0, 4, 6])
def test_trainer_step_device_reduce_then_scale_matches_global_batch_sgd(early):
    """Trainer.step_device under two gloo ranks: early bucket reduced asynchronously (started in backward), the rest
    after it, 1/world in the optimiser -- parameters and reported loss equal single-process SGD on the global batch."""
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_trainer_worker, args=(r, ws, port, q, early)) for r in range(ws)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(3)
    X, Y = torch.randn(16, 6, generator=g), torch.randn(16, generator=g)
    w = torch.zeros(6)
    ref_losses = []
    for _ in range(3):
        r = X @ w - Y
        ref_losses.append(float((r ** 2).mean() * 0.5))
        w = w - 0.1 * (X.t() @ r) / 16
    for r in res:
        torch.testing.assert_close(torch.tensor(r[1]), w, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(torch.tensor(r[2]), torch.tensor(ref_losses), rtol=1e-5, atol=1e-6)
        assert r[3] == 3


def test_rank_aware_batches_cover_an_epoch_once():
    """Dataset.get_batches(rank=, world=, seed=): all ranks walk one shuffled order; an epoch's batches are disjoint
    across ranks and cover every example once; the single-process call is unchanged."""
    from fvta_memexqa_amd.utils import Dataset
    n = 23
    data = {"q": list(range(n)), "aid": [[] for _ in range(n)]}
    shared = {"albums": {}, "pid2feat": {"p": __import__("numpy").zeros(3, "float32")}}
    ds = Dataset(data, "train", shared=shared)
    seen = []
    for rank in range(2):
        for idxs, _ in ds.get_batches(4, 100, shuffle=True, cap=True, rank=rank, world=2, seed=11):
            assert len(idxs) <= 4
            seen += list(idxs)
    assert sorted(seen) == list(range(n))
    with pytest.raises(ValueError):
        next(ds.get_batches(4, 1, shuffle=True, rank=0, world=2))
    # the short last global group (23 examples; 3 ranks x 4: the tail holds 11; 4 ranks x 4: it holds 7): every rank gets a
    # non-empty batch, the tail is covered once
    for world, bs in ((3, 4), (4, 4)):
        tails, seen = [], []
        for rank in range(world):
            batches = [idxs for idxs, _ in ds.get_batches(bs, 100, shuffle=False, cap=True, rank=rank, world=world)]
            assert all(len(b) > 0 for b in batches), (world, rank, batches)
            tails.append(batches[-1])
            seen += [i for b in batches for i in b]
        assert sorted(seen) == list(range(n))
        assert max(len(t) for t in tails) - min(len(t) for t in tails) <= 1
    # fewer examples left than ranks: no way to give every rank a batch -- refused rather than hung
    small = Dataset({"q": list(range(9)), "aid": [[] for _ in range(9)]}, "train", shared=shared)
    with pytest.raises(ValueError):
        list(small.get_batches(4, 100, shuffle=False, cap=True, rank=0, world=2))   # 9 = 8 + 1: one example for two ranks
    a = [i for i, _ in ds.get_batches(5, 5, shuffle=False)]
    assert a[0] == (0, 1, 2, 3, 4) and a[4] == (20, 21, 22)


def test_single_process_is_a_no_op():
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    from fvta_memexqa_amd import dist
    assert dist.world()[0] == 1
    g = torch.ones(4)
    assert dist.allreduce_grads(g) == 1.0 and torch.equal(g, torch.ones(4))
    with pytest.raises(ValueError):
        dist.shard_range(10, 4, 0)
