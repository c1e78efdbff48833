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


def test_single_process_is_a_no_op():
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    from fvta_memexqa_amd import dist
    assert dist.world()[0] == 1
    g = torch.ones(4)
    assert dist.allreduce_grads(g) == 1.0 and torch.equal(g, torch.ones(4))
    with pytest.raises(ValueError):
        dist.shard_range(10, 4, 0)
