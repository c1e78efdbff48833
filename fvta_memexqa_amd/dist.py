"""Data parallelism over QA pairs: one process per GPU, albums/questions sharded
along the batch axis, ONE collective per training step -- a sum all-reduce of the flat fp32
gradient buffer over RCCL/xGMI (backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).
The reference has no distributed code at all; forward/inference needs no collective."""
import os

import torch
import torch.distributed as dist


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_* (torch.distributed.run)."""
    ws, rank, local = world()
    if ws == 1 and os.environ.get("FVTA_DIST_FORCE", "0") != "1":   # FVTA_DIST_FORCE=1: exercise the collective path with one rank
        return ws, rank, local
    if backend is None:   # FVTA_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses that) -- a test affordance
        backend = os.environ.get("FVTA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws, **kw)
    return ws, rank, local


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("FVTA_DIST_FORCE", "0") == "1")


def shard_range(global_batch, ws, rank):
    """Contiguous equal shards of the global batch (config 4: 512 -> 8 x 64)."""
    if global_batch % ws:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, ws))
    per = global_batch // ws
    return rank * per, (rank + 1) * per


def allreduce_async(flat_slice):
    """Start the sum all-reduce of a slice of the flat gradient buffer whose producers are already enqueued on the
    CURRENT stream (torch orders the collective's stream behind it); returns the work handle, or None when there is one
    rank.  Used for the "early" bucket -- scorer / attention / photo-cell gradients, final while the text cell's
    backward recurrence still runs -- so that its wire time hides behind that recurrence."""
    if not is_dist():
        return None
    return dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_grads(flat_grad, early_numel=0, early_work=None):
    """Sum the flat gradient bucket across ranks; returns the 1/world factor the optimiser must apply so that the
    update equals the reference's global-batch mean (model_v2.py:1090).  If the first `early_numel` elements are
    already being reduced (`early_work` from allreduce_async) only the rest is reduced here, then the early work is
    waited for (the current stream waits, not the host)."""
    if not is_dist():
        return 1.0
    if early_work is not None and 0 < early_numel < flat_grad.numel():
        dist.all_reduce(flat_grad[early_numel:], op=dist.ReduceOp.SUM)
        early_work.wait()
    else:
        if early_work is not None:
            early_work.wait()
            if early_numel >= flat_grad.numel():
                return 1.0 / dist.get_world_size()
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


def mean_over_ranks(t):
    """mean of a (device or CPU) tensor over ranks, in place: the loss a data-parallel step reports is the global-batch
    mean (equal shards), as the single-process reference's is"""
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= dist.get_world_size()
    return t


def barrier():
    if is_dist():
        dist.barrier()


def max_over_ranks(value, device):
    if not is_dist():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device):
    """every rank's `value` (a float), in rank order -- bench.py's per-rank step times"""
    if not is_dist():
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def describe():
    """what the process group itself reports (not the environment): backend and world size"""
    if not is_dist():
        return dict(backend=None, world_size=1, rank=0)
    return dict(backend=str(dist.get_backend()), world_size=int(dist.get_world_size()), rank=int(dist.get_rank()))


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
