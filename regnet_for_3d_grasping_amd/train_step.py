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
    Parameters whose ``.grad`` is None (e.g. the reference's never-used ``linear_cls``) are skipped on
    every rank alike.  ``reduce='mean'`` divides by the world size."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if reduce == "mean":
        flat /= dist.get_world_size()
    offset = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[offset:offset + n].view_as(g))
        offset += n
    return flat.numel()


class ScoreTrainer:
    """ScoreNet + Adam + StepLR with the reference's hyper-parameters; ``step(pc, pc_score)`` is one
    training iteration on this rank's scenes and returns the (local) loss."""

    def __init__(self, score_net, lr=0.001, reduce="sum"):
        self.net = score_net
        self.reduce = reduce
        self.optimizer = torch.optim.Adam([{"params": score_net.parameters(), "initial_lr": lr}], lr=lr)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=5, gamma=0.5)

    def step(self, pc, pc_score, pc_label=None):
        self.net.train()
        self.optimizer.zero_grad()
        with torch.enable_grad():
            _, _, loss = self.net(pc, pc_score, pc_label)
            loss_total = loss.sum()
            loss_total.backward()
        allreduce_gradients(list(self.net.parameters()), self.reduce)
        self.optimizer.step()
        return loss_total.detach()

    def end_epoch(self):
        self.scheduler.step()
