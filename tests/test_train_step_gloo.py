"""Data-parallel training step on CPU: two gloo ranks, oracle backend, a small SA+FP network.
Checks the single flat gradient all-reduce (sum over ranks; fixed buffer layout: a parameter without a gradient on
a rank contributes zeros, a parameter without a gradient on EVERY rank stays grad-less), that replicas stay
bit-identical after the optimizer step, the rank-0 broadcast of the initial state, and the reference's full
training iteration (``RefineTrainer``) on two ranks where one rank falls back to the ScoreNet loss alone
(train.py:430-435) -- the case that used to issue mismatched collectives."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TinySeg(torch.nn.Module):
    """One SA + one FP level + a score head, built from the same modules as ScoreNet."""

    def __init__(self):
        super().__init__()
        from regnet_for_3d_grasping_amd.pn2_utils.modules import PointNetSAModule, PointnetFPModule
        self.sa = PointNetSAModule(3, (16, 32), 64, 0.15, 16, True)
        self.fp = PointnetFPModule(32 + 3, (16,), 3)
        self.head = torch.nn.Conv1d(16, 1, 1)
        self.unused = torch.nn.Linear(4, 4)          # like the reference's linear_cls: never gets a grad

    def forward(self, pc, target):
        pts = pc.permute(0, 2, 1)
        xyz, rgb = pts[:, :3, :], pts[:, 3:6, :]
        new_xyz, feat = self.sa(xyz, rgb)
        up = self.fp(xyz, new_xyz, rgb, feat)
        score = torch.sigmoid(self.head(up)).squeeze(1)
        return torch.nn.functional.mse_loss(score, target)


def _make(seed_offset, n):
    from regnet_for_3d_grasping_amd import synthetic
    pc = torch.from_numpy(np.stack([synthetic.make_scene(7000 + seed_offset, n)], 0))
    target = torch.from_numpy(np.random.default_rng(seed_offset).uniform(0, 1, (1, n)).astype(np.float32))
    return pc, target


def _local_grads(rank, n):
    from oracle.install import oracle_backend
    torch.manual_seed(0)
    net = TinySeg().train()
    pc, target = _make(rank, n)
    with oracle_backend():
        loss = net(pc, target)
        loss.backward()
    return net, loss


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from regnet_for_3d_grasping_amd import sharding, train_step
    dist = sharding.init("gloo")
    net, loss = _local_grads(rank, n)
    local = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    count = train_step.allreduce_gradients(list(net.parameters()), "sum")
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    opt.step()
    torch.save({"local": local, "reduced": {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                "params": {k: p.detach().clone() for k, p in net.named_parameters()}, "count": count,
                "unused_has_grad": net.unused.weight.grad is not None}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks(tmp_path):
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "2"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, n = 2, 1024
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(world))
    assert not r0["unused_has_grad"] and "unused.weight" not in r0["reduced"]
    # every trainable parameter owns a slot (the never-used layer included: 4*4 + 4 zeros), on both ranks alike
    assert r0["count"] == r1["count"] == sum(v.numel() for v in r0["local"].values()) + 20
    for k in r0["local"]:
        want = r0["local"][k] + r1["local"][k]              # DataParallel sums the replica losses
        assert float(want.abs().max()) > 0 or k.endswith("bias")
        torch.testing.assert_close(r0["reduced"][k], want, rtol=0, atol=1e-7)
        assert torch.equal(r0["reduced"][k], r1["reduced"][k])
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k])  # replicas stay in sync after the step


# ---- the real trainer: one rank loses its region stage -------------------------------------------------------
def _refine_worker(rank, world, port, out_dir, fail_rank, no_refine_rank):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(4)
    import datetime
    import torch.distributed as td
    from oracle.install import oracle_backend
    from regnet_for_3d_grasping_amd import synthetic, train_step
    from regnet_for_3d_grasping_amd import get_regiondataset as grd
    from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
    from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
    from tests import golden_util as gu
    td.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    cfg = gu.meta_train()["cfg"]
    N = cfg["N"]
    # DIFFERENT initial weights per rank on purpose: the trainer must broadcast rank 0's
    score_net = ScoreNetwork(training=True)
    score_net.load_state_dict(synthetic.seeded_state_dict(score_net, cfg["score_weights_seed"] + rank))
    region_net = GripperRegionNetwork(training=True, group_num=cfg["params"][2], gripper_num=cfg["gripper_num"],
                                      grasp_score_threshold=cfg["grasp_score_threshold"],
                                      radius=cfg["gripper_params"][2], reg_channel=cfg["reg_channel"])
    region_net.load_state_dict(synthetic.seeded_state_dict(region_net, cfg["region_weights_seed"] + rank))
    trainer = train_step.RefineTrainer(score_net, region_net, cfg["params"], cfg["gripper_params"])
    start = {k: v.clone() for k, v in list(score_net.state_dict().items()) + list(region_net.state_dict().items())}
    pc = synthetic.make_batch(cfg["scene_seed"] + rank, 1, N)
    records = [synthetic.make_grasp_labels(pc[0].numpy(), cfg["label_seed"] + rank)]
    pc_score = torch.from_numpy(np.random.default_rng(cfg["label_seed"] + rank).uniform(0, 1, (1, N)).astype(np.float32))
    torch.manual_seed(cfg["torch_seed"] + rank)
    np.random.seed(cfg["np_seed"] + rank)
    if rank == fail_rank:      # this rank's region stage fails (the reference's bare `except`, train.py:430)
        def boom(*a, **k):
            raise RuntimeError("no labelled centre in this batch")
        grd.get_grasp_allobj = boom
    if rank == no_refine_rank:  # this rank skips the refine loss (fewer than two grasps in the gripper)
        orig = region_net.forward

        def no_refine(*a, **k):
            res = list(orig(*a, **k))
            res[13] = res[13][:2]
            return tuple(res)
        region_net.forward = no_refine
    calls = []
    orig_all_reduce = td.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls.append(int(t.numel()))
        return orig_all_reduce(t, *a, **k)
    td.all_reduce = counting_all_reduce
    with oracle_backend():
        total, parts = trainer.step(pc, pc_score, records)
    td.all_reduce = orig_all_reduce
    named = list(score_net.named_parameters()) + [("region." + k, p) for k, p in region_net.named_parameters()]
    torch.save({"start": start, "total": float(total), "region_error": parts.get("region_error"),
                "stage2": parts["stage2"] is not None, "refine": parts["refine"] is not None,
                "params": {k: p.detach().clone() for k, p in named},
                "has_grad": {k: p.grad is not None for k, p in named}, "all_reduce_sizes": calls,
                "n_params": len(named), "n_grad": sum(p.numel() for _, p in named)},
               os.path.join(out_dir, "refine_rank%d.pt" % rank))
    td.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_refine(tmp_path, fail_rank, no_refine_rank):
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "4"
    mp.spawn(_refine_worker, args=(2, _free_port(), str(tmp_path), fail_rank, no_refine_rank), nprocs=2, join=True)
    return [torch.load(os.path.join(tmp_path, "refine_rank%d.pt" % r)) for r in range(2)]


def test_refine_trainer_two_ranks_one_rank_without_region_stage(tmp_path):
    r0, r1 = _run_refine(tmp_path, fail_rank=1, no_refine_rank=-1)
    assert r0["region_error"] is None and r0["stage2"]
    assert r1["region_error"] is not None and not r1["stage2"]        # rank 1 took the fallback ...
    for k in r0["start"]:
        assert torch.equal(r0["start"][k], r1["start"][k])              # ... both started from rank 0's weights
    moved = 0
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k         # ... and the replicas are still identical
        assert r0["has_grad"][k] == r1["has_grad"][k], k
        moved += int(r0["has_grad"][k])
    # rank 1 received the region network's gradients from rank 0; the never-used layer stays grad-less everywhere
    # same collectives on both ranks: one flag per parameter, then ONE all-reduce of every gradient of BOTH networks
    assert r0["all_reduce_sizes"] == r1["all_reduce_sizes"] == [r0["n_params"], r0["n_grad"]]
    assert r1["has_grad"]["region.extrat_feature_region.conv.weight"]
    assert not r0["has_grad"]["region.extrat_feature_region.linear_cls.weight"]
    assert moved > 80


def test_refine_trainer_two_ranks_one_rank_without_refine_loss(tmp_path):
    r0, r1 = _run_refine(tmp_path, fail_rank=-1, no_refine_rank=0)
    assert r0["stage2"] and not r0["refine"] and r1["stage2"]
    assert r0["all_reduce_sizes"] == r1["all_reduce_sizes"] == [r0["n_params"], r0["n_grad"]]
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
        assert r0["has_grad"][k] == r1["has_grad"][k], k
