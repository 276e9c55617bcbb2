"""Where a training iteration's wall time goes: phases of RefineTrainer.step timed with a device sync after each
(so overlap between phases is lost; the sum is an upper bound of the pipelined step).  python train_phases.py [B]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
from regnet_for_3d_grasping_amd.train_step import RefineTrainer
dev = "cuda:0"
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 25600
pc = synthetic.make_batch(8100, B, N)
records = [synthetic.make_grasp_labels(pc[b].numpy(), 50 + b) for b in range(B)]
target = torch.from_numpy(np.random.default_rng(2).uniform(0, 1, (B, N)).astype(np.float32)).to(dev)
s = ScoreNetwork(training=True); s.load_state_dict(synthetic.seeded_state_dict(s, 3))
r = GripperRegionNetwork(training=True, group_num=256, gripper_num=64, grasp_score_threshold=0.5, radius=0.06, reg_channel=10)
r.load_state_dict(synthetic.seeded_state_dict(r, 4))
t = RefineTrainer(s.to(dev), r.to(dev), pipeline.PARAMS, pipeline.GRIPPER_PARAMS)
t.score_net.train(); t.region_net.train()
pc = pc.to(dev)
np.random.seed(1)
for _ in range(3): t.step(pc, target, records)
acc = {}
def mark(name, t0):
    torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
n = 6
for _ in range(n):
    plan = t.geometry.acquire(t.prefetch(pc), pc.device); torch.cuda.synchronize()
    t0 = time.perf_counter()
    t.opt_score.zero_grad(); t.opt_region.zero_grad()
    all_feature, score, loss = t.score_net(pc, target, None, plan=plan)
    t0 = mark("scorenet forward", t0)
    with contextlib.redirect_stdout(io.StringIO()):
        g = get_grasp_allobj(pc, score, t.params, records)
        t0 = mark("get_grasp_allobj (centres, grouping, labels)", t0)
        res = t.region_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, all_feature, t.gripper_params, g[6], records)
    t0 = mark("region_net forward + losses", t0)
    total = loss.sum() + res[3][0].sum() + (res[13][0].sum() if len(res[13]) > 2 else 0)
    total.backward()
    t0 = mark("backward", t0)
    t.opt_score.step(); t.opt_region.step()
    t0 = mark("two Adam steps", t0)
for k, v in acc.items():
    print("%-48s %7.2f ms" % (k, v / n * 1e3))
print("%-48s %7.2f ms" % ("sum", sum(acc.values()) / n * 1e3))
