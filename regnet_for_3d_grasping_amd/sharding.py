"""Scene sharding across ranks (one process per GPU).

The forward hot path has no cross-scene dependency, so multi-GPU execution is a partition of the
scene stream: rank ``r`` of ``W`` owns scenes ``r*B .. r*B+B-1`` of every global batch and never
exchanges activations.  The only collective is the timing reduction of the bench contract
(max over ranks); ``torch.distributed`` backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests.
"""
import os

import torch


# A process group of ONE rank normally counts as "not distributed" (no collective is issued, no bucket is built).  With this
# switch on it counts: `scripts/scale_driver.sh` at N = 1 (bench.py --single-rank-group) then runs the SAME code a multi-GPU
# job runs -- the RCCL communicator, `describe_collective`, the gradient bucket's side-stream all-reduce -- on a one-GPU box.
SINGLE_RANK_GROUP = False


def group_active():
    """True when collectives should be issued: an initialised process group of more than one rank (or of one rank under
    SINGLE_RANK_GROUP)."""
    import torch.distributed as dist
    return bool(dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or SINGLE_RANK_GROUP))


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1-process default)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def scene_seeds(rank, world, batch_per_rank, first_seed=1000, step=0):
    """Seeds of the synthetic scenes rank ``rank`` owns in global batch ``step`` (disjoint across ranks)."""
    base = first_seed + (step * world + rank) * batch_per_rank
    return list(range(base, base + batch_per_rank))


def init(backend, device=None, reserve=("forward", "train")):
    """Create the process group ("nccl" = RCCL).  ``reserve``: which of this package's HIP stream sets are created and bound
    to hardware queues BEFORE RCCL creates its own ("forward": pipeline.reserve_streams, "train": train_step.reserve_streams,
    in this order) -- stream -> queue binding is first come first served and a pipeline bound behind RCCL's streams runs
    9 % slower (pipeline.reserve_streams).  GPU devices only."""
    import torch.distributed as dist
    if not dist.is_initialized():
        if device is not None and torch.device(device).type == "cuda":
            for what in reserve or ():
                if what == "forward":
                    from . import pipeline
                    pipeline.reserve_streams(device)
                elif what == "train":
                    from . import train_step
                    train_step.reserve_streams(device)
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


def describe_collective(device, probe_elements=7066927 + 128, probe_iters=5):
    """What the process group REALLY is, for the bench line of a multi-rank run (``config.collective``): backend and world
    size as the group reports them, the RCCL version, a startup all-reduce that proves ``world`` distinct ranks on distinct
    devices, and the duration / bus bandwidth of an all-reduce the size of the training iteration's gradient bucket
    (``probe_elements`` fp32: both networks' 7 066 927 parameters + the presence flags, utils.py:129-133 replaced by ONE
    flat all-reduce).  Collective: every rank must call it.  Returns None without an initialised group of > 1 ranks."""
    import time

    import torch.distributed as dist
    if not group_active():
        return None
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = str(dist.get_backend())
    on_gpu = torch.device(device).type == "cuda"
    # one-hot rank vector: the sum over ranks is all ones iff the group holds `world` DISTINCT ranks
    onehot = torch.zeros(world, dtype=torch.int64, device=device)
    onehot[rank] = 1
    dist.all_reduce(onehot, op=dist.ReduceOp.SUM)
    distinct_ranks = int((onehot == 1).sum().item())
    # (device index, PCI bus id) of every rank's GPU
    ident = torch.zeros(2, dtype=torch.int64, device=device)
    if on_gpu:
        idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        props = torch.cuda.get_device_properties(idx)
        ident[0], ident[1] = idx, (getattr(props, "pci_domain_id", 0) << 16) | (getattr(props, "pci_bus_id", 0) << 8) | getattr(props, "pci_device_id", 0)
    gathered = [torch.zeros_like(ident) for _ in range(world)]
    dist.all_gather(gathered, ident)
    devices = sorted({(int(g[0]), int(g[1])) for g in gathered})
    version = None
    if backend == "nccl":
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except (RuntimeError, AttributeError, TypeError):
            version = None
    buf = torch.ones(probe_elements, dtype=torch.float32, device=device)

    def fence():
        if on_gpu:
            torch.cuda.synchronize()
    dist.all_reduce(buf)            # warm-up: communicator set-up, first-use allocations
    fence()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(probe_iters):
        dist.all_reduce(buf)
    fence()
    ms = max_over_ranks((time.perf_counter() - t0) / probe_iters * 1e3, device)
    nbytes = probe_elements * 4
    return {"backend": backend, "world_size": world, "rccl_version": version,
            "distinct_ranks_by_allreduce": distinct_ranks, "distinct_devices": len(devices),
            "devices": ["cuda:%d pci %06x" % d for d in devices],
            "probe_bytes": nbytes, "probe_allreduce_ms": round(ms, 4),
            # ring convention: every byte crosses 2 (W - 1) / W links' worth of the slowest link
            "bus_GBps": round(2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 3) if ms > 0 else None}


# ---- host side of one-process-per-GPU: which cores a rank's Python, OpenMP and torch intra-op threads run on ------------
def _parse_cpulist(text):
    """'0-15,128-143' -> [0, ..., 15, 128, ..., 143] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def rank_core_slice(local_rank, local_world, cpus):
    """The share of ``cpus`` (a list of logical CPU ids, e.g. a NUMA node's) that local rank ``local_rank`` of
    ``local_world`` ranks sharing them gets: contiguous, disjoint, every rank at least one core."""
    cpus = list(cpus)
    per = max(1, len(cpus) // max(1, local_world))
    lo = (local_rank * per) % max(1, len(cpus))
    return cpus[lo:lo + per] or cpus[:1]


def gpu_numa_node(device_index):
    """NUMA node of GPU ``device_index`` from sysfs (PCI bus id of the torch device), or None when unknown."""
    try:
        bus = torch.cuda.get_device_properties(device_index).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(device_index), "pci_domain_id", 0)
        dev = getattr(torch.cuda.get_device_properties(device_index), "pci_device_id", 0)
        with open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (dom, bus, dev)) as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except (OSError, ValueError, RuntimeError, AssertionError, AttributeError):
        return None


def pin_rank(local_rank, local_world, device_index=None):
    """Give this rank's process its own cores: the region stage's host thread (numpy draws, label matching), the launch
    thread and the OpenMP / torch intra-op pools of N ranks on one host otherwise migrate over all sockets and fight for
    the same cores.  The slice comes from the GPU's NUMA node when sysfs knows it (ranks whose GPUs share a node split
    that node's cores), else from the process's current affinity mask.  Returns the list of cores (also when the
    platform has no ``sched_setaffinity``: then nothing is pinned and only the thread pools are sized)."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cpus, share, slot = allowed, local_world, local_rank
    node = gpu_numa_node(device_index) if device_index is not None else None
    if node is not None:
        try:
            with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
                on_node = [c for c in _parse_cpulist(f.read()) if c in set(allowed)]
            peers = [r for r in range(local_world) if gpu_numa_node(r) == node]   # local ranks = device indices
            if on_node and local_rank in peers:
                cpus, share, slot = on_node, len(peers), peers.index(local_rank)
        except OSError:
            pass
    mine = rank_core_slice(slot, share, cpus)
    if hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            pass
    threads = max(1, len(mine) // 2)          # half for the intra-op pool, the rest for the launch + region threads
    os.environ["OMP_NUM_THREADS"] = str(threads)
    torch.set_num_threads(threads)
    return mine
