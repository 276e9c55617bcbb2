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
    """One collective for all gradients: flatten -> all_reduce(SUM) -> scatter back.

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
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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


def broadcast_module_state(*modules, src=0):
    """Make every rank start from rank ``src``'s parameters and buffers (BatchNorm running statistics included):
    one flat broadcast per dtype.  ``nn.DataParallel`` re-replicates device 0's weights every step
    (utils.py:129-133); with one process per GPU the replicas only stay identical if they START identical and
    see identical (all-reduced) gradients -- without this an unseeded ``construct_scorenet(load_flag=False)``
    would silently train N different models.  No-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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


class ScoreTrainer:
    """ScoreNet + Adam + StepLR with the reference's hyper-parameters; ``step(pc, pc_score)`` is one
    training iteration on this rank's scenes and returns the (local) loss."""

    def __init__(self, score_net, lr=0.001, reduce="sum"):
        self.net = score_net
        self.reduce = reduce
        broadcast_module_state(score_net)
        self.optimizer = torch.optim.Adam([{"params": score_net.parameters(), "initial_lr": lr}], lr=lr)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=5, gamma=0.5)
        self.geometry = GeometryPrefetcher(score_net)

    def prefetch(self, pc):
        """Start the geometry of a FUTURE batch on a side stream; pass the result to ``step(..., plan=...)``."""
        return self.geometry.prefetch(pc)

    def step(self, pc, pc_score, pc_label=None, plan=None):
        self.net.train()
        self.optimizer.zero_grad()
        with torch.enable_grad():
            _, _, loss = self.net(pc, pc_score, pc_label, plan=GeometryPrefetcher.acquire(plan, pc.device))
            loss_total = loss.sum()
            loss_total.backward()
        allreduce_gradients(list(self.net.parameters()), self.reduce)
        self.optimizer.step()
        return loss_total.detach()

    def end_epoch(self):
        self.scheduler.step()


class GeometryPrefetcher:
    """Computes the geometry plan of a batch (every FPS / ball-query / 3-NN index of ScoreNet: functions of xyz only,
    no gradients) on a side HIP stream, so that the NEXT batch's ~9 ms level-1 sampling chain -- one CU per scene --
    runs underneath the current iteration's forward/backward instead of at the head of its own."""

    def __init__(self, score_net):
        self.score_net = score_net
        self.stream = None

    def prefetch(self, pc):
        """pc (B,N,6) on the GPU -> handle for ``step(..., plan=handle)``.  Enqueue-only, does not block the host."""
        from . import fused
        if self.stream is None:
            self.stream = torch.cuda.Stream(pc.device, priority=-1)
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


class RefineTrainer:
    """The reference's full training iteration (``--mode train``, train.py:347-384): ScoreNet and the
    grasp-region/refine network are trained together, one Adam + StepLR each,
    ``loss_total = score_loss + stage-2 loss (+ refine loss)``.  Like the reference, a failure inside
    the region stage (e.g. no labelled centre in the batch) falls back to the ScoreNet loss alone.
    One flat gradient all-reduce per network when ``torch.distributed`` is initialised."""

    def __init__(self, score_net, region_net, params, gripper_params, lr=0.001, reduce="sum"):
        self.score_net, self.region_net = score_net, region_net
        self.params, self.gripper_params, self.reduce = params, gripper_params, reduce
        broadcast_module_state(score_net, region_net)
        self.opt_score = torch.optim.Adam([{"params": score_net.parameters(), "initial_lr": lr}], lr=lr)
        self.opt_region = torch.optim.Adam([{"params": region_net.parameters(), "initial_lr": lr}], lr=lr)
        self.sched_score = torch.optim.lr_scheduler.StepLR(self.opt_score, step_size=5, gamma=0.5)
        self.sched_region = torch.optim.lr_scheduler.StepLR(self.opt_region, step_size=5, gamma=0.5)
        self.geometry = GeometryPrefetcher(score_net)

    def prefetch(self, pc):
        """Start the geometry of a FUTURE batch on a side stream; pass the result to ``step(..., plan=...)``."""
        return self.geometry.prefetch(pc)

    def forward_losses(self, pc, pc_score, grasp_records, plan=None):
        """-> (loss_total, parts) with parts = dict(score=..., stage2=... or None, refine=... or None)."""
        import contextlib
        import io

        from .get_regiondataset import get_grasp_allobj
        all_feature, output_score, loss = self.score_net(pc, pc_score, None,
                                                         plan=GeometryPrefetcher.acquire(plan, pc.device))
        parts = {"score": loss, "stage2": None, "refine": None}
        total = loss.sum()
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                g = get_grasp_allobj(pc, output_score, self.params, grasp_records)
                res = self.region_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, all_feature, self.gripper_params, g[6],
                                      grasp_records)
            loss_tuple, loss_refine_tuple = res[3], res[13]
            total = total + loss_tuple[0].sum()
            parts["stage2"] = loss_tuple[0]
            if len(loss_refine_tuple) > 2:
                total = total + loss_refine_tuple[0].sum()
                parts["refine"] = loss_refine_tuple[0]
        except (RuntimeError, IndexError, ValueError) as exc:   # the reference uses a bare except (train.py:430)
            parts["region_error"] = repr(exc)
        return total, parts

    def step(self, pc, pc_score, grasp_records, plan=None):
        self.score_net.train()
        self.region_net.train()
        self.opt_score.zero_grad()
        self.opt_region.zero_grad()
        with torch.enable_grad():
            total, parts = self.forward_losses(pc, pc_score, grasp_records, plan)
            total.backward()
        allreduce_gradients(list(self.score_net.parameters()), self.reduce)
        allreduce_gradients(list(self.region_net.parameters()), self.reduce)
        self.opt_score.step()
        self.opt_region.step()
        return total.detach(), parts

    def end_epoch(self):
        self.sched_score.step()
        self.sched_region.step()
