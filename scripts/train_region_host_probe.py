"""Host-paced part of a training iteration: the region stage (get_grasp_allobj with labels + GripperRegionNetwork forward
with both losses) alone on an otherwise idle device -- wall time per call, cProfile by cumulative time (our files) and by
own time (everything).  python scripts/train_region_host_probe.py [B]"""
import cProfile, contextlib, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
dev = "cuda:0"
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 25600
pc = synthetic.make_batch(1000, B, N)
records = [synthetic.make_grasp_labels(pc[b].numpy(), 50 + b) for b in range(B)]
target = torch.from_numpy(np.random.default_rng(2).uniform(0, 1, (B, N)).astype(np.float32)).to(dev)
s = ScoreNetwork(training=True); s.load_state_dict(synthetic.seeded_state_dict(s, 7))
r = GripperRegionNetwork(training=True, group_num=256, gripper_num=64, grasp_score_threshold=0.5, radius=0.06, reg_channel=10)
r.load_state_dict(synthetic.seeded_state_dict(r, 11))
synthetic.set_region_head_affine(r)
s, r = s.to(dev).train(), r.to(dev).train()
pc = pc.to(dev)
np.random.seed(1)
with torch.enable_grad():
    all_feature, score, loss = s(pc, target, None)
all_feature = all_feature.detach().requires_grad_(True)
score = score.detach()

def region():
    with contextlib.redirect_stdout(io.StringIO()):
        g = get_grasp_allobj(pc, score, pipeline.PARAMS, records)
        res = r(g[3], g[5], g[2], g[4], g[0], g[1], pc, all_feature, pipeline.GRIPPER_PARAMS, g[6], records)
    total = res[3][0].sum() + (res[13][0].sum() if len(res[13]) > 2 else 0)
    return total

for _ in range(3):
    region()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    tot = region()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("region stage forward (labels, both losses), B=%d: host %.2f ms per call (+ %.2f ms device drain at the end)" % (B, (t1 - t0) / n * 1e3, (t2 - t1) * 1e3))
t0 = time.perf_counter()
for _ in range(n):
    tot = region(); tot.backward()
torch.cuda.synchronize()
print("  ... with its backward: %.2f ms per call" % ((time.perf_counter() - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    region()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats("regnet_for_3d_grasping_amd", 28)
st.sort_stats("tottime").print_stats(25)
