"""Data-parallel ScoreNet training step (the reference's ``--mode pretrain_score`` step,
train.py:124-150: forward with labels -> ``loss.sum()`` -> backward -> Adam, lr 1e-3, StepLR(5, 0.5)
per epoch, utils.py:117-121) with one process per GPU instead of ``nn.DataParallel``
(utils.py:129-133).

Each rank runs the whole step on its own scenes -- train-mode forward/backward go through the
operator-granular kernels (``group_points`` / ``feature_interpolate`` backward are the fp32
scatter-adds of csrc/gather.hip, convolutions and BatchNorm through torch autograd) -- and the
only exchange is ONE all-reduce of the flat fp32 gradient buffer (5 542 531 elements = 22.2 MB for
ScoreNet) over RCCL/xGMI ("nccl" backend; "gloo" in the CPU tests).  DataParallel SUMS the
per-replica mean losses (train.py:138), so the default reduction is a sum, not a mean; BatchNorm
statistics stay per rank exactly as DataParallel keeps them per replica.
"""
import torch


def allreduce_gradients(parameters, reduce="sum"):
    """One collective for all gradients: flatten -> all_reduce(SUM) -> scatter back.  (Stand-alone helper for an
    arbitrary parameter list; the trainers use the persistent ``GradientBucket`` below, which builds nothing per step.)

    The buffer layout depends on the PARAMETER LIST only, never on this rank's data: every parameter with
    ``requires_grad`` owns a slot, and a parameter whose ``.grad`` is None on this rank (the reference's never-used
    ``linear_cls``; the whole region network when this rank's region stage failed; the refine heads when this
    rank had fewer than two grasps in the gripper) contributes zeros.  All ranks therefore always issue the same
    all_reduce of the same length -- a rank that skipped a loss can neither shorten the collective nor miss it
    (``nn.DataParallel``, utils.py:129-133, has no such failure mode either: its replicas reduce every parameter).
    After the reduction a parameter receives a ``.grad`` iff ANY rank had one (decided by a per-parameter
    participation count carried at the tail of the same buffer), so replicas apply identical optimizer steps and
    parameters no rank touched keep ``grad is None`` (Adam leaves their state alone, like a single process).
    ``reduce='mean'`` divides by the world size.  Returns the number of gradient elements reduced."""
    import torch.distributed as dist
    if not _distributed():
        return 0
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return 0
    n_grad = sum(p.numel() for p in params)
    dev = params[0].device
    pieces = [p.grad.reshape(-1).to(torch.float32) if p.grad is not None
              else torch.zeros(p.numel(), dtype=torch.float32, device=dev) for p in params]
    pieces.append(torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=torch.float32).to(dev))
    flat = torch.cat(pieces)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if reduce == "mean":
        flat[:n_grad] /= dist.get_world_size()
    present = flat[n_grad:].tolist()     # one host read per step; identical on every rank
    offset = 0
    for i, p in enumerate(params):
        n = p.numel()
        if present[i] > 0:
            g = flat[offset:offset + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
        offset += n
    return n_grad


# RefineTrainer: back-propagate the ScoreNet loss through the segmentation head RIGHT AFTER the ScoreNet forward, before the
# region stage.  The region stage (centre selection, grouping draws, label matching, class-balanced losses) is ~8-10 ms
# of host-paced work during which the GPU has nothing to do, and the head's backward (four 204 800-row layers: ~5 ms of
# GEMMs and BatchNorm passes at 8 scenes) depends on the score loss only; its gradient with respect to the 256-channel
# point feature is kept and joins the region losses' gradient at that tensor, so the trunk is still traversed once.
# Same gradients as one ``total.backward()`` up to the order of one fp32 addition (tests/test_gpu_train.py).
EARLY_HEAD_BACKWARD = True
# Workgroup slots (of 2 per CU) the head's persistent contractions leave empty while they run beside the region stage: a
# persistent workgroup holds its CU's register file for the whole launch, so without free slots the region stage's kernels
# advance one launch per contraction (profiles/r04m: gather_max 0.38 instead of 0.06 ms, the heads' layers 0.1-0.36 instead of
# 0.02-0.09).  64 slots = 64 CUs running one workgroup instead of two for 1.5 ms per iteration.
HEAD_BACKWARD_FREE_SLOTS = 64


def _distributed():
    from . import sharding
    return sharding.group_active()


_stream_pools = {}


# Priority class of the region stage's stream.  The stage is the iteration's critical path between the forward and the trunk's backward
# (5.7 ms, host-paced; the head's backward beside it takes 4.1 and has the slack), and -1 (the high-priority class) shortens an
# iteration of a process that runs NOTHING ELSE by 0.3-0.5 ms -- but in a process that has also created an inference pipeline's
# streams (bench.py's default line: the `train` object after the forward measurement) it costs 15 ms: 45.6 -> 60.3 ms per iteration,
# every phase of the trunk stream 1.3-1.45x slower (stream -> hardware-queue binding again, see pipeline.reserve_streams).  Stays 0.
REGION_STREAM_PRIORITY = 0


def reserve_streams(device):
    """The side streams of a training iteration on ``device`` -- {"geometry": the next batch's sampling / grouping (low
    priority... the same ``priority=-1`` class the pipeline uses), "region": the region stage beside the head's backward,
    "capture": hipGraph captures, "comm": the gradient all-reduce} -- created once per device, bound to hardware queues in the
    order an iteration first uses them, and shared by every trainer built afterwards.  ``sharding.init`` calls this BEFORE it
    creates the process group: bound behind RCCL's own streams the region stage shares a hardware queue with the trunk (its
    kernels then wait for the head's backward: +3 ms per iteration, see pipeline.reserve_streams)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _stream_pools.get(idx)
    if pool is None:
        pool = {"geometry": torch.cuda.Stream(dev, priority=-1), "region": torch.cuda.Stream(dev, priority=REGION_STREAM_PRIORITY),
                "capture": torch.cuda.Stream(dev), "comm": torch.cuda.Stream(dev)}
        for name in ("geometry", "region", "capture", "comm"):
            _ = pool[name].cuda_stream
        _stream_pools[idx] = pool
    return pool


class GradientBucket:
    """ONE persistent flat fp32 buffer for the gradients of one or several networks, and ONE gradient all-reduce per step.

    Layout (a function of the parameter list only, fixed at construction, identical on every rank): ``flat`` =
    ``[grad(p_0) | grad(p_1) | ... ]`` over every trainable parameter of ``modules`` in order, ``flags`` = one presence word
    per parameter.  ``prepare()`` (instead of ``zero_grad``) zeroes the buffer with one fill and points every ``p.grad`` at its
    slice, so autograd ACCUMULATES STRAIGHT INTO the buffer -- nothing is gathered or concatenated per step.  Which
    parameters received a gradient is recorded by post-accumulate hooks on the host (no device read).
    ``reduce_gradients()`` issues two collectives on a side stream: first the presence flags (n_params words, < 1 KB) --
    they depend on nothing the device is still computing, so that all-reduce and the copy of its result to pinned memory run
    AHEAD of the backward still in flight -- then the gradients, behind the compute stream (bracketed by events:
    ``last_ms``).  The host needs the reduced flags before the optimizers (a parameter NO rank touched -- the reference's
    never-used ``linear_cls``, the region network of a step in which every rank took the fallback of train.py:430-435 --
    keeps ``grad is None``, so optimizers skip it exactly as in a single process) and gets them without waiting for the
    backward or the gradient all-reduce: the launching thread keeps running ahead of the device (round 5 read the flags out
    of the gradient all-reduce's own buffer: one collective, but a host stall of a whole backward per iteration -- 5 ms of
    a 46 ms iteration once the iteration was replayed from hipGraphs).  A rank that skipped a loss still issues the same
    two collectives of the same lengths (its slices hold zeros)."""

    def __init__(self, modules, reduce="sum"):
        self.params = [p for m in modules for p in m.parameters() if p.requires_grad]
        self.reduce = reduce
        self.n_grad = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(self.n_grad, dtype=torch.float32, device=dev)
        self.flags = torch.zeros(len(self.params), dtype=torch.float32, device=dev)
        self.views, offset = [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("GradientBucket holds float32 parameters")
            self.views.append(self.flat[offset:offset + p.numel()].view_as(p))
            offset += p.numel()
        self.touched = [False] * len(self.params)
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self.touched.__setitem__(i, True))
        self.collectives = 0            # gradient all-reduces issued so far
        self.last_ms = None             # event-timed duration of the last gradient all-reduce (GPU tensors)
        self._comm_stream = None
        self._events = None
        self._host_flags = None
        self._local_flags = None
        self._timing_pending = False

    def mark_touched(self, params):
        """Gradients written into ``p.grad`` by hand (``torch.autograd.grad`` results) do not fire the accumulate hooks."""
        ids = {id(p) for p in params}
        for i, p in enumerate(self.params):
            if id(p) in ids:
                self.touched[i] = True

    def prepare(self, lazy=()):
        """Replaces ``optimizer.zero_grad()``: one fill, every ``.grad`` a view of the bucket.  Parameters in ``lazy`` get
        ``grad = None`` instead: autograd then ASSIGNS their gradient (no in-place add kernel per parameter inside the
        backward) and ``reduce_gradients`` moves what it finds into their slices with one multi-tensor copy -- for networks
        whose backward is a host-paced chain of small launches on the iteration's critical path (the region network)."""
        self.flat.zero_()
        skip = {id(p) for p in lazy}
        for p, v in zip(self.params, self.views):
            p.grad = None if id(p) in skip else v
        self.touched = [False] * len(self.params)

    def _collect_lazy(self):
        """Gradients autograd assigned outside the bucket (``prepare(lazy=...)``) -> their slices; ``p.grad`` = the slice."""
        dst, src = [], []
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g.reshape(v.shape) if g.shape != v.shape else g)
                p.grad = v
                self.touched[i] = True
        if dst:
            torch._foreach_copy_(dst, src)

    def _world(self):
        import torch.distributed as dist
        if _distributed():
            return dist, dist.get_world_size()
        return None, 1

    def reduce_gradients(self):
        """The step's collectives; afterwards ``p.grad`` is the reduced gradient, or None where no rank had one.
        Returns the number of gradient elements reduced (0 without a process group)."""
        dist, world = self._world()
        self._collect_lazy()
        local = self.touched
        if dist is None:
            for p, t in zip(self.params, local):
                if not t:
                    p.grad = None
            return 0
        if self.flat.is_cuda:
            dev = self.flat.device
            if self._comm_stream is None:
                self._comm_stream = reserve_streams(dev)["comm"]
                self._events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self._host_flags = torch.empty(len(self.params), dtype=torch.float32).pin_memory()
                self._local_flags = torch.empty(len(self.params), dtype=torch.float32).pin_memory()
            cur = torch.cuda.current_stream(dev)
            self._local_flags.copy_(torch.tensor([1.0 if t else 0.0 for t in local], dtype=torch.float32))
            with torch.cuda.stream(self._comm_stream):
                # the flags: nothing here waits for the compute stream
                self.flags.copy_(self._local_flags, non_blocking=True)
                dist.all_reduce(self.flags, op=dist.ReduceOp.SUM)
                self._host_flags.copy_(self.flags, non_blocking=True)
                flags_done = torch.cuda.Event()
                flags_done.record()
            self._comm_stream.wait_stream(cur)                # the gradients: behind the backward
            with torch.cuda.stream(self._comm_stream):
                self._events[0].record()
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                self._events[1].record()
                if self.reduce == "mean":
                    self.flat /= world
            cur.wait_stream(self._comm_stream)                # the optimizer's kernels queue behind the collective
            flags_done.synchronize()                          # host: the flags only (microseconds; the backward is still running)
            present = self._host_flags.tolist()
            self._timing_pending = True
        else:
            self.flags.copy_(torch.tensor([1.0 if t else 0.0 for t in local], dtype=torch.float32))
            dist.all_reduce(self.flags, op=dist.ReduceOp.SUM)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if self.reduce == "mean":
                self.flat /= world
            present = self.flags.tolist()
        self.collectives += 1
        for p, v, n in zip(self.params, self.views, present):
            p.grad = v if n > 0 else None      # (a lazy parameter only ANOTHER rank touched receives its slice here)
        return self.n_grad

    def last_allreduce_ms(self):
        """Event-timed duration of the most recent gradient all-reduce whose events have completed (GPU tensors), else None.
        Does not block: a collective still in flight leaves the previous value."""
        if self._events is not None and self._timing_pending and self._events[1].query():
            self.last_ms = self._events[0].elapsed_time(self._events[1])
            self._timing_pending = False
        return self.last_ms


def broadcast_module_state(*modules, src=0):
    """Make every rank start from rank ``src``'s parameters and buffers (BatchNorm running statistics included):
    one flat broadcast per dtype.  ``nn.DataParallel`` re-replicates device 0's weights every step
    (utils.py:129-133); with one process per GPU the replicas only stay identical if they START identical and
    see identical (all-reduced) gradients -- without this an unseeded ``construct_scorenet(load_flag=False)``
    would silently train N different models.  No-op without an initialised process group."""
    import torch.distributed as dist
    if not _distributed():
        return 0
    tensors = []
    for m in modules:
        tensors.extend(p.data for p in m.parameters())
        tensors.extend(b.data for b in m.buffers())
    total = 0
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype in sorted(by_dtype, key=str):      # same order on every rank
        group = by_dtype[dtype]
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src=src)
        offset = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[offset:offset + n].view_as(t))
            offset += n
        total += flat.numel()
    return total


def broadcast_buffers(*modules, src=0):
    """Every rank takes rank ``src``'s BUFFERS (BatchNorm running statistics, ``num_batches_tracked``): parameters stay
    identical across ranks by construction (same start, all-reduced gradients), running statistics do not -- each rank
    tracks its own shard, as each ``nn.DataParallel`` replica does, and the reference's checkpoint holds device 0's
    (utils.py:129-133, train.py:467-468).  Call before evaluating or saving on a rank other than ``src`` --
    ``save_checkpoint`` does.  No-op without a process group.  Returns the number of elements broadcast."""
    import torch.distributed as dist
    if not _distributed():
        return 0
    total = 0
    by_dtype = {}
    for m in modules:
        for b in m.buffers():
            by_dtype.setdefault(b.dtype, []).append(b.data)
    for dtype in sorted(by_dtype, key=str):
        group = by_dtype[dtype]
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src=src)
        offset = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[offset:offset + n].view_as(t))
            offset += n
        total += flat.numel()
    return total


def save_checkpoint(score_net, region_net, score_path, region_path, src=0):
    """The reference's end-of-epoch save (train.py:467-468: two whole-object pickles) under one process per GPU: all ranks
    take rank ``src``'s BatchNorm buffers (a collective: EVERY rank must call this), then rank ``src`` alone writes the
    files (``checkpoint.save_model``: the reference's class paths)."""
    import torch.distributed as dist
    from . import checkpoint
    nets = [n for n in (score_net, region_net) if n is not None]
    broadcast_buffers(*nets, src=src)
    rank = dist.get_rank() if _distributed() else 0
    if rank == src:
        if score_net is not None:
            checkpoint.save_model(score_net, score_path)
        if region_net is not None:
            checkpoint.save_model(region_net, region_path)
    if _distributed():
        dist.barrier()


def _adam(module, lr):
    """The reference's optimizer (utils.py:118: Adam over the network's parameters, default betas / eps) -- on the GPU as torch's FUSED
    implementation (one multi-tensor kernel per step instead of ~10 foreach launches per state tensor group: two steps take
    0.3 ms instead of 1.06 at the reference's 212 parameter tensors); same update rule."""
    params = list(module.parameters())
    fused = bool(params) and all(p.is_cuda for p in params)
    return torch.optim.Adam([{"params": params, "initial_lr": lr}], lr=lr, **({"fused": True} if fused else {}))


class ScoreTrainer:
    """ScoreNet + Adam + StepLR with the reference's hyper-parameters; ``step(pc, pc_score)`` is one
    training iteration on this rank's scenes and returns the (local) loss."""

    def __init__(self, score_net, lr=0.001, reduce="sum"):
        self.net = score_net
        self.reduce = reduce
        self.optimizer = _adam(score_net, lr)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=5, gamma=0.5)
        self.geometry = GeometryPrefetcher(score_net)
        self.bucket = None           # created by the first step that finds a process group (_ensure_bucket)
        self._iterations = 0
        self._ensure_bucket()

    def prefetch(self, pc):
        """Start the geometry of a FUTURE batch on a side stream; pass the result to ``step(..., plan=...)``."""
        return self.geometry.prefetch(pc)

    def _ensure_bucket(self):
        """The gradient all-reduce is decided per STEP, not at construction: a process group initialised after the
        trainer was built must not leave the ranks training apart without an error."""
        if self.bucket is None and _distributed():
            if getattr(self, "_iterations", 0) > 0:    # (optimizer moments / step counts of the local steps would stay per rank)
                raise RuntimeError("a process group appeared after %d local training steps: build the ScoreTrainer after "
                                   "torch.distributed is initialised" % self._iterations)
            broadcast_module_state(self.net)
            self.bucket = GradientBucket([self.net], self.reduce)

    def step(self, pc, pc_score, pc_label=None, plan=None):
        self.net.train()
        self._ensure_bucket()
        if self.bucket is None:
            self.optimizer.zero_grad()
        else:
            self.bucket.prepare()
        with torch.enable_grad():
            _, _, loss = self.net(pc, pc_score, pc_label, plan=GeometryPrefetcher.acquire(plan, pc.device))
            loss_total = loss.sum()
            loss_total.backward()
        if self.bucket is not None:
            self.bucket.reduce_gradients()
        self.optimizer.step()
        self._iterations += 1
        return loss_total.detach()

    def end_epoch(self):
        self.scheduler.step()


class GeometryPrefetcher:
    """Computes the geometry plan of a batch (every FPS / ball-query / 3-NN index of ScoreNet: functions of xyz only,
    no gradients) on a side HIP stream, so that the NEXT batch's ~4 ms level-1 sampling chain -- one CU per scene --
    runs underneath the current iteration's forward/backward instead of at the head of its own."""

    def __init__(self, score_net):
        self.score_net = score_net
        self.stream = None

    def prefetch(self, pc):
        """pc (B,N,6) on the GPU -> handle for ``step(..., plan=handle)``.  Enqueue-only, does not block the host."""
        from . import fused
        if self.stream is None:
            self.stream = reserve_streams(pc.device)["geometry"]
        cur = torch.cuda.current_stream(pc.device)
        self.stream.wait_stream(cur)          # pc may still be in flight on the caller's stream
        with torch.cuda.stream(self.stream):
            plan = self.score_net.plan(pc)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        pc.record_stream(self.stream)
        return {"plan": plan, "ready": ready, "tensors": fused.plan_tensors(plan)}

    @staticmethod
    def acquire(handle, device):
        """Make the current stream wait for a prefetched plan; returns the plan (None passes through)."""
        if handle is None:
            return None
        cur = torch.cuda.current_stream(device)
        cur.wait_event(handle["ready"])
        for t in handle["tensors"]:
            t.record_stream(cur)
        return handle["plan"]



# RefineTrainer: replay the fixed-shape part of an iteration -- ScoreNet's forward, the segmentation head's backward, the trunk's
# backward: ~450 of an iteration's ~590 launches, 11-12 ms of the launching thread's time at 8 x 25 600 -- as three hipGraphs
# (``_TrunkGraphs``) once this many eager iterations of the same shape have run (every lazily built cache must exist before
# a capture).  The region stage in between is the reference's host code (data-dependent sizes, numpy's stream) and stays
# eager.  Module switch, like the others: nothing in the product reads the environment.
TRAIN_GRAPHS = True
GRAPH_WARMUP_ITERATIONS = 2


def _clone_plan(plan):
    """A geometry plan with every tensor cloned (static addresses for a captured forward); other entries are kept."""
    return {kind: [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in level.items()} for level in plan[kind]]
            for kind in ("sa", "fp")}


class _TrunkGraphs:
    """hipGraph replays of the fixed-shape part of ONE training iteration shape (B, N) of ``RefineTrainer``:

        g_forward   ScoreNet forward with labels (train-mode BatchNorm incl. running statistics, dropout from the graph-safe
                    generator state) + the score loss + the feature map's contiguous rows for the region stage's pools
        g_head      the segmentation head's backward (``torch.autograd.grad`` down to the 256-channel point feature)
        g_trunk     the trunk's backward from the point feature's gradient -- a STATIC buffer the head's backward writes and
                    the region stage's pools add to (``region_ops.set_feature_grad_sink``)

    captured once from the same autograd graph the eager iteration builds (same kernels, same order, same streams' worth of
    work) over static copies of the inputs -- the batch, its labels, its geometry plan (20 index / distance tensors, copied in
    by two ``_foreach_copy_`` launches) -- and one private memory pool (~30 GB at 8 x 25 600: the activations an eager
    iteration allocates and frees).  Parameter gradients land in static tensors (``p.grad`` of the ScoreNet parameters, or the
    ``GradientBucket``'s views when a process group exists: captured as in-place accumulations behind ``prepare()``'s fill).
    Valid while the parameters keep their addresses (optimizer steps and ``load_state_dict`` update in place) and the
    module switches their values; ``matches`` is checked every step and a mismatch drops back to the eager path."""

    def __init__(self, trainer, pc, pc_score, plan):
        import gc
        import weakref

        from . import conv1x1_train, fused
        net = trainer.score_net
        seg = net.extrat_featurePN2
        dev = pc.device
        self.key = self._key(pc, pc_score)
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.signature = self._signature()
        self.head = [p for m in (seg.mlp, seg.conv_score, seg.bn_score) for p in m.parameters() if p.requires_grad]
        self.pc, self.target = pc.clone(), pc_score.clone()
        self.plan = _clone_plan(plan)
        self.plan_tensors = fused.plan_tensors(self.plan)
        self.stream = reserve_streams(dev)["capture"]
        bucket = trainer.bucket
        if bucket is not None:
            bucket.prepare()                 # p.grad = the bucket's views: the captured accumulations are in-place adds
        else:
            for p in self.params:
                p.grad = None                # the captured backward ASSIGNS: these tensors are the static gradients
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        # no destructor of a stale graph may run while a stream captures (pipeline._StageGraphs): collect now, collector off
        gc.collect()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        grabbed = {}
        hook = seg.register_forward_hook(lambda _m, _i, out: grabbed.__setitem__("feat", out[0]))
        # The captured forward sees every parameter through a fresh LEAF that shares its storage (optimizer steps and
        # ``load_state_dict`` update in place, so replays read the current values).  Why not the parameters themselves: a
        # parameter's gradient-accumulator node belongs to the stream of the iteration that created it and lives as long as
        # ANY graph of that iteration (a loss a caller still holds); the autograd engine synchronises a backward with that
        # node's stream even when ``autograd.grad`` only captures the gradient -- for a node of an earlier eager iteration that
        # is a stream outside the capture (observed: hipStreamEndCapture crashes).  Fresh leaves get their nodes inside it.
        named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
        alias = {n: p.detach().requires_grad_(True) for n, p in named}
        alias_of = {id(p): alias[n] for n, p in named}
        head_ids = {id(p) for p in self.head}
        trunk = [p for p in self.params if id(p) not in head_ids]
        try:
            self.g_forward, self.g_head, self.g_trunk = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # capture_error_mode "thread_local": other threads of the process keep making calls a capture forbids -- RCCL's
            # watchdog polls its collectives' events (hipEventQuery: observed to abort the process under the default, global,
            # mode), a data loader may allocate; the autograd engine's device thread LAUNCHES into the capturing stream, which
            # any mode records
            with torch.enable_grad():
                with torch.cuda.graph(self.g_forward, stream=self.stream, capture_error_mode="thread_local"):
                    all_feature, self.score, self.loss = torch.func.functional_call(
                        net, alias, (self.pc, self.target, None), {"plan": self.plan})
                    self.total = self.loss.sum()
                    self.all_feature = all_feature.detach()
                    self.rows = self.all_feature.contiguous().view(-1, all_feature.shape[2])
                feat = grabbed["feat"]
                before = conv1x1_train.reserve_stream_slots(HEAD_BACKWARD_FREE_SLOTS)
                try:
                    with torch.cuda.graph(self.g_head, stream=self.stream, pool=self.g_forward.pool(), capture_error_mode="thread_local"):
                        grads = torch.autograd.grad(self.total, [feat] + [alias_of[id(p)] for p in self.head], retain_graph=True,
                                                    allow_unused=True)
                        self.g_feat = grads[0]
                        if bucket is not None:      # (multi-tensor adds: a handful of nodes instead of one per parameter)
                            pairs = [(p.grad, g) for p, g in zip(self.head, grads[1:]) if g is not None]
                            torch._foreach_add_([a for a, _ in pairs], [b for _, b in pairs])
                finally:
                    conv1x1_train.reserve_stream_slots(before)
                if not (self.g_feat.is_contiguous() and tuple(self.g_feat.shape) == tuple(feat.shape)):
                    raise RuntimeError("the point feature's gradient is not a contiguous (B, C, N) tensor")
                self.head_grads = [(p, g) for p, g in zip(self.head, grads[1:]) if g is not None]
                with torch.cuda.graph(self.g_trunk, stream=self.stream, pool=self.g_forward.pool(), capture_error_mode="thread_local"):
                    tgrads = torch.autograd.grad([feat], [alias_of[id(p)] for p in trunk], [self.g_feat], allow_unused=True)
                    if bucket is not None:
                        pairs = [(p.grad, g) for p, g in zip(trunk, tgrads) if g is not None]
                        torch._foreach_add_([a for a, _ in pairs], [b for _, b in pairs])
                self.trunk_grads = [(p, g) for p, g in zip(trunk, tgrads) if g is not None]
                if bucket is None:
                    for p, g in self.trunk_grads + self.head_grads:
                        p.grad = g
        finally:
            hook.remove()
            if gc_was_enabled:
                gc.enable()
        cur.wait_stream(self.stream)
        # (the captured autograd graph has been consumed; what is kept are plain tensors in the graphs' pool)
        self.score, self.loss, self.total = self.score.detach(), self.loss.detach(), self.total.detach()
        if bucket is not None:
            bucket.mark_touched([p for p, _ in self.head_grads + self.trunk_grads])
            self.touched = [i for i, t in enumerate(bucket.touched) if t]
            self.static_grads = None
        else:
            self.static_grads = self.trunk_grads + self.head_grads
        self._weakref = weakref
        self.replays = 0

    @staticmethod
    def _key(pc, pc_score):
        return (tuple(pc.shape), pc.dtype, tuple(pc_score.shape), pc_score.dtype, pc.device.index)

    def _signature(self):
        from . import bn_train, conv1x1_train
        from .pn2_utils import modules
        switches = tuple((m.__name__, k, v) for m in (conv1x1_train, bn_train, modules) for k, v in sorted(vars(m).items())
                         if k.isupper() and isinstance(v, (bool, int, float)))
        return (tuple(p.data_ptr() for p in self.params), switches, HEAD_BACKWARD_FREE_SLOTS)

    def matches(self, pc, pc_score):
        return self._key(pc, pc_score) == self.key and self._signature() == self.signature

    def load(self, pc, pc_score, plan):
        """This iteration's batch, labels and geometry plan -> the static buffers (on the current stream)."""
        from . import fused
        self.pc.copy_(pc)
        self.target.copy_(pc_score)
        src = fused.plan_tensors(plan)
        if len(src) != len(self.plan_tensors) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(src, self.plan_tensors)):
            raise RuntimeError("the geometry plan of this batch does not have the captured plan's layout")
        torch._foreach_copy_(self.plan_tensors, src)

    def feature_leaf(self):
        """The replayed feature map as a fresh autograd leaf for the region stage, its contiguous rows already made."""
        from . import gripper_region_network
        leaf = self.all_feature.detach().requires_grad_(True)
        gripper_region_network._rows_cache.ref = self._weakref.ref(leaf)
        gripper_region_network._rows_cache.flat = self.rows
        return leaf


class RefineTrainer:
    """The reference's full training iteration (``--mode train``, train.py:347-384): ScoreNet and the
    grasp-region/refine network are trained together, one Adam + StepLR each,
    ``loss_total = score_loss + stage-2 loss (+ refine loss)``.  Like the reference, a failure inside
    the region stage (e.g. no labelled centre in the batch) falls back to the ScoreNet loss alone.
    ONE flat gradient all-reduce per iteration (both networks share a ``GradientBucket``) when ``torch.distributed`` is
    initialised.  BatchNorm running statistics stay per rank, as ``nn.DataParallel`` keeps them per replica (only device
    0's survive there; here ``broadcast_module_state`` before saving a checkpoint gives every rank rank 0's)."""

    def __init__(self, score_net, region_net, params, gripper_params, lr=0.001, reduce="sum", gc_interval=None, graphs=None):
        """``graphs``: replay ScoreNet's forward / the head's backward / the trunk's backward as hipGraphs (``_TrunkGraphs``)
        once GRAPH_WARMUP_ITERATIONS eager iterations of a shape have run; None = the module switch TRAIN_GRAPHS (GPU only).
        ``gc_interval``: every iteration builds and drops an autograd graph of a few thousand Python objects; CPython's
        automatic cyclic collector then runs a full collection every ~10 iterations that takes 60-100 ms with the device idle
        (measured: iterations of 59.5 ms with spikes of 106-157 ms; none with the collector off).  With ``gc_interval=N`` the
        trainer switches the automatic collector OFF (process-wide, like the "manual GC" switches of large training
        frameworks) and collects itself every N iterations; None leaves the interpreter alone."""
        self.score_net, self.region_net = score_net, region_net
        self.gc_interval, self._iterations = gc_interval, 0
        if gc_interval:
            import gc
            gc.collect()
            gc.freeze()      # what exists now (modules, parameters, the interpreter's own objects) is never traversed again:
            gc.disable()     # a periodic collection then only walks the iterations' garbage (~100 ms -> a few ms)
        self.params, self.gripper_params, self.reduce = params, gripper_params, reduce
        self.opt_score = _adam(score_net, lr)
        self.opt_region = _adam(region_net, lr)
        self.sched_score = torch.optim.lr_scheduler.StepLR(self.opt_score, step_size=5, gamma=0.5)
        self.sched_region = torch.optim.lr_scheduler.StepLR(self.opt_region, step_size=5, gamma=0.5)
        self.geometry = GeometryPrefetcher(score_net)
        self._region_stream = None
        self._region_params = [p for p in region_net.parameters() if p.requires_grad]
        self._marks = None
        self.graphs = graphs
        self._graphs = None              # _TrunkGraphs of the shape the last iterations had
        self._eager_shape = (None, 0)    # (shape key, eager iterations of it so far)
        self.graph_replays = 0           # iterations served by replays (bench.py reports it)
        # measurement hook (bench.py --train): a list makes every iteration append {name: HIP event} marks on the stream
        # the trunk runs on -- "start", and for a replayed iteration "forward", "head", "joined" (the region stage and its
        # backward have been waited for), "trunk"; "end" behind the optimizers -- so that the trunk stream's own busy time is
        # read from events, without a profiler
        self.phase_marks = None
        # both networks' gradients in ONE flat buffer: one all-reduce per training iteration (28.3 MB for the reference's
        # 5 542 531 + 1 524 396 parameters)
        self.bucket = None           # created by the first step that finds a process group (_ensure_bucket)
        self._ensure_bucket()

    def _ensure_bucket(self):
        """The gradient all-reduce is decided per STEP, not once at construction: a process group initialised after the
        trainer was built must not leave the ranks training apart without an error."""
        if self.bucket is None and _distributed():
            if self._iterations > 0:
                # rank 0's parameters could be broadcast, but not the Adam moments, step counts and StepLR state the local
                # steps already built: the ranks would apply different optimizer states to the same all-reduced gradient
                raise RuntimeError("a process group appeared after %d local training steps: build the RefineTrainer (or "
                                   "restore a checkpoint into it) after torch.distributed is initialised" % self._iterations)
            broadcast_module_state(self.score_net, self.region_net)
            self.bucket = GradientBucket([self.score_net, self.region_net], self.reduce)

    def prefetch(self, pc):
        """Start the geometry of a FUTURE batch on a side stream; pass the result to ``step(..., plan=...)``."""
        return self.geometry.prefetch(pc)

    def forward_losses(self, pc, pc_score, grasp_records, plan=None, early_head_backward=False):
        """-> (loss_total, parts) with parts = dict(score=..., stage2=... or None, refine=... or None).
        ``early_head_backward`` (used by ``step``): the score loss is back-propagated through the segmentation head
        right after the ScoreNet forward (see EARLY_HEAD_BACKWARD); ``parts['early']`` then carries what ``step`` needs
        to finish the backward pass, and the returned total has the same VALUE but only the region losses' graph."""
        grabbed = {}
        hook = None
        if early_head_backward and torch.is_grad_enabled():
            # the tensor the segmentation head consumes = PointNet2Seg's first output (held by this call only)
            hook = self.score_net.extrat_featurePN2.register_forward_hook(
                lambda _m, _i, out: grabbed.__setitem__("feat", out[0]))
        try:
            all_feature, output_score, loss = self.score_net(pc, pc_score, None,
                                                             plan=GeometryPrefetcher.acquire(plan, pc.device))
        finally:
            if hook is not None:
                hook.remove()
        parts = {"score": loss, "stage2": None, "refine": None}
        total = loss.sum()
        feat = grabbed.get("feat")
        forward_done = None
        if feat is not None and feat.requires_grad and total.requires_grad:
            seg = self.score_net.extrat_featurePN2
            head = [p for m in (seg.mlp, seg.conv_score, seg.bn_score) for p in m.parameters() if p.requires_grad]
            if feat.is_cuda:
                forward_done = torch.cuda.Event()
                forward_done.record()
            # the head's backward shares the GPU with the region stage's small host-paced kernels: its persistent
            # contractions leave HEAD_BACKWARD_FREE_SLOTS workgroup slots empty for them (conv1x1_train.reserve_stream_slots)
            from . import conv1x1_train
            before = conv1x1_train.reserve_stream_slots(HEAD_BACKWARD_FREE_SLOTS) if feat.is_cuda else None
            try:
                grads = torch.autograd.grad(total, [feat] + head, retain_graph=True, allow_unused=True)
            finally:
                if before is not None:
                    conv1x1_train.reserve_stream_slots(before)
            # the region stage sees the feature map as a LEAF: its backward (``step``) ends there, the trunk is traversed once
            all_feature = all_feature.detach().requires_grad_(True)
            parts["early"] = (feat, grads[0], head, grads[1:], total, all_feature)
            total = total.detach()       # its gradient is already out; what is added below is the region stage's share
        total = self._region_stage(pc, output_score, all_feature, grasp_records, total, parts, forward_done)
        return total, parts

    def _region_stage(self, pc, output_score, all_feature, grasp_records, total, parts, forward_done=None):
        """Centre selection, grouping, grasp-region + refine networks and their losses (train.py:354-372) -> ``total`` plus the
        region losses (``parts`` filled in).  ``forward_done``: an event recorded behind ScoreNet's forward when more work (the
        segmentation head's backward) has been enqueued on the current stream since -- the stage then runs on its OWN stream
        behind that event only: its device->host reads would otherwise wait for that work (autograd runs a node's backward on
        the stream of its forward and orders streams itself; ``step`` joins the streams before the optimizer)."""
        import contextlib
        import io

        from .get_regiondataset import get_grasp_allobj
        region_stream = contextlib.nullcontext()
        if forward_done is not None:
            if self._region_stream is None:
                self._region_stream = reserve_streams(all_feature.device)["region"]
            self._region_stream.wait_event(forward_done)
            region_stream = torch.cuda.stream(self._region_stream)
            parts["region_stream"] = self._region_stream
        with region_stream:
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    if all_feature.is_cuda:
                        # the feature map as contiguous rows (what both pools gather from: a 210 MB transpose at B = 8), enqueued
                        # BEFORE the stage's first device->host read instead of between its host-paced launches (a replayed
                        # forward has made them already: ``_TrunkGraphs`` seeds the cache)
                        from . import gripper_region_network
                        gripper_region_network._contiguous_rows(all_feature, detach=all_feature.requires_grad)
                    g = get_grasp_allobj(pc, output_score, self.params, grasp_records, defer_large_groups=True)
                    res = self.region_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, all_feature, self.gripper_params, g[6],
                                          grasp_records)
                loss_tuple, loss_refine_tuple = res[3], res[13]
                total = total + loss_tuple[0].sum()
                parts["stage2"] = loss_tuple[0]
                if len(loss_refine_tuple) > 2:
                    total = total + loss_refine_tuple[0].sum()
                    parts["refine"] = loss_refine_tuple[0]
            except (RuntimeError, IndexError, ValueError, ArithmeticError) as exc:   # the reference uses a bare except (train.py:430)
                parts["region_error"] = repr(exc)
        return total

    def _graphs_for(self, pc, pc_score, plan):
        """The ``_TrunkGraphs`` to replay this iteration with, or None for an eager iteration."""
        on = TRAIN_GRAPHS if self.graphs is None else self.graphs
        if not (on and EARLY_HEAD_BACKWARD and pc.is_cuda and pc_score is not None and pc_score.is_cuda):
            return None
        G = self._graphs
        if G is not None:
            if G.matches(pc, pc_score):
                return G
            self._graphs = None          # another shape / moved parameters / flipped switches: eager again, then a new capture
            self._eager_shape = (None, 0)
        key = _TrunkGraphs._key(pc, pc_score)
        if self._eager_shape[0] != key or self._eager_shape[1] < GRAPH_WARMUP_ITERATIONS:
            return None
        geo = GeometryPrefetcher.acquire(plan, pc.device)
        with torch.no_grad():
            self._graphs = _TrunkGraphs(self, pc, pc_score, geo if geo is not None else self.score_net.plan(pc))
        return self._graphs

    def _step_graphed(self, G, pc, pc_score, grasp_records, plan):
        """``step`` with the fixed-shape part replayed (``_TrunkGraphs``); the region stage, its backward, the all-reduce and
        the optimizers are the eager iteration's."""
        from . import region_ops
        dev = pc.device
        cur = torch.cuda.current_stream(dev)
        geo = GeometryPrefetcher.acquire(plan, dev)
        if geo is None:
            with torch.no_grad():
                geo = self.score_net.plan(pc)
        G.load(pc, pc_score, geo)
        marks = self._marks
        if self.bucket is None:
            self.opt_region.zero_grad()
            for p, g in G.static_grads:      # (an eager iteration in between would have replaced them)
                p.grad = g
        else:
            self.bucket.prepare(lazy=self._region_params)
            for i in G.touched:
                self.bucket.touched[i] = True
        G.g_forward.replay()
        forward_done = torch.cuda.Event(enable_timing=marks is not None)
        forward_done.record()
        G.g_head.replay()
        if marks is not None:
            marks["forward"] = forward_done
            self._mark("head")
        parts = {"score": G.loss.clone(), "stage2": None, "refine": None}
        with torch.enable_grad():
            leaf = G.feature_leaf()
            total = self._region_stage(pc, G.score, leaf, grasp_records, G.total.clone(), parts, forward_done)
            side = parts.pop("region_stream", None)
            if side is not None:
                cur.wait_stream(side)
            if total.requires_grad:
                head_done = torch.cuda.Event()
                head_done.record()
                region_ops.set_feature_grad_sink(G.g_feat, head_done)
                try:
                    total.backward()
                finally:
                    region_ops.set_feature_grad_sink(None)
                if side is not None:
                    cur.wait_stream(side)
                if leaf.grad is not None:
                    G.g_feat.add_(leaf.grad.transpose(1, 2))
        self._mark("joined")
        G.g_trunk.replay()
        self._mark("trunk")
        G.replays += 1
        self.graph_replays += 1
        return total.detach(), parts

    def _mark(self, name):
        if self._marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks[name] = ev

    def step(self, pc, pc_score, grasp_records, plan=None):
        self.score_net.train()
        self.region_net.train()
        self._ensure_bucket()
        self._marks = {} if (self.phase_marks is not None and pc.is_cuda) else None
        self._mark("start")
        G = self._graphs_for(pc, pc_score, plan)
        if G is not None:
            total, parts = self._step_graphed(G, pc, pc_score, grasp_records, plan)
            return self._finish_step(total, parts)
        if pc.is_cuda and pc_score is not None:
            key = _TrunkGraphs._key(pc, pc_score)
            self._eager_shape = (key, self._eager_shape[1] + 1 if self._eager_shape[0] == key else 1)
        if self.bucket is None:      # single process: gradients stay where autograd puts them (no accumulate-into-view adds)
            self.opt_score.zero_grad()
            self.opt_region.zero_grad()
        else:
            self.bucket.prepare(lazy=self._region_params)
        with torch.enable_grad():
            total, parts = self.forward_losses(pc, pc_score, grasp_records, plan, early_head_backward=EARLY_HEAD_BACKWARD)
            early = parts.pop("early", None)
            side = parts.pop("region_stream", None)
            if side is not None:
                torch.cuda.current_stream(pc.device).wait_stream(side)   # (device-side: the region stage's losses are roots below)
            if early is None:
                total.backward()
            else:
                feat, g_feat, head, g_head, _, region_feature = early
                for p, g in zip(head, g_head):
                    if g is not None:
                        if p.grad is None:
                            p.grad = g
                        else:
                            p.grad.add_(g)
                if self.bucket is not None:
                    self.bucket.mark_touched([p for p, g in zip(head, g_head) if g is not None])
                if total.requires_grad:
                    # the region stage contributed losses.  Its backward first -- it ends at the leaf it was given for the feature
                    # map, and on the GPU the two pools add their gradient straight into the head's
                    # (region_ops.set_feature_grad_sink); whatever else reaches the leaf arrives in its .grad.  Then the trunk,
                    # once, from the sum.
                    from . import region_ops
                    if g_feat is not None and g_feat.is_cuda and g_feat.is_contiguous():
                        head_done = torch.cuda.Event()
                        head_done.record()
                        region_ops.set_feature_grad_sink(g_feat, head_done)
                    try:
                        total.backward()
                    finally:
                        region_ops.set_feature_grad_sink(None)
                    if side is not None:
                        torch.cuda.current_stream(pc.device).wait_stream(side)
                    if region_feature.grad is not None:
                        extra = region_feature.grad.transpose(1, 2)
                        g_feat = extra if g_feat is None else g_feat + extra
                torch.autograd.backward([feat], [g_feat])
        return self._finish_step(total.detach(), parts)

    def _finish_step(self, total, parts):
        """The single gradient all-reduce, the two optimizer steps, housekeeping.  What is handed back carries no autograd
        graph: a caller that keeps ``parts`` must not keep the iteration's activations (and its AccumulateGrad nodes) alive."""
        parts = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in parts.items()}
        if self.bucket is not None:
            self.bucket.reduce_gradients()
        self.opt_score.step()
        self.opt_region.step()
        if self._marks is not None:
            self._mark("end")
            self.phase_marks.append(self._marks)
            self._marks = None
        from . import gripper_region_network
        gripper_region_network.forget_rows()     # the iteration's contiguous feature rows (210 MB at B = 8) and their graph
        self._iterations += 1
        if self.gc_interval and self._iterations % self.gc_interval == 0:
            import gc
            gc.collect()
        return total, parts

    def end_epoch(self):
        from . import pn2_ext
        pn2_ext.raise_if_fps_failed()    # cooperative sampling launches (scenes beyond 25 600 points) flag a lost partner
        self.sched_score.step()
        self.sched_region.step()
