"""Scene sharding across ranks (one process per GPU).

The forward hot path has no cross-scene dependency, so multi-GPU execution is a partition of the
scene stream: rank ``r`` of ``W`` owns scenes ``r*B .. r*B+B-1`` of every global batch and never
exchanges activations.  The only collective is the timing reduction of the bench contract
(max over ranks); ``torch.distributed`` backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests.
"""
import os

import torch


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1-process default)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def scene_seeds(rank, world, batch_per_rank, first_seed=1000, step=0):
    """Seeds of the synthetic scenes rank ``rank`` owns in global batch ``step`` (disjoint across ranks)."""
    base = first_seed + (step * world + rank) * batch_per_rank
    return list(range(base, base + batch_per_rank))


def init(backend, device=None):
    import torch.distributed as dist
    if not dist.is_initialized():
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, **kw)
    return dist


def max_over_ranks(seconds, device="cpu"):
    """Wall time of the slowest rank (the bench contract's MAX over ranks)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(count, device="cpu"):
    """Sum of a per-rank scalar (e.g. scenes processed) over all ranks."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(count)
    t = torch.tensor([count], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
