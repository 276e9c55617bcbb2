"""Wall time of one reference-style training iteration (train.py:347-384: ScoreNet + grouping with labels + stage-2
and refine losses, backward, two Adam steps) on one GPU, operator-granular kernels + torch autograd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
from regnet_for_3d_grasping_amd.train_step import RefineTrainer
dev = "cuda:0"
torch.backends.cudnn.benchmark = bool(int(os.environ.get("CUDNN_BENCH", "0")))
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
PRE = bool(int(os.environ.get("PREFETCH", "1")))   # geometry of the next batch on a side stream
ahead = t.prefetch(pc) if PRE else None
def one(ahead):
    nxt = t.prefetch(pc) if PRE else None
    out = t.step(pc, target, records, plan=ahead)
    return out, nxt
for _ in range(2): _, ahead = one(ahead)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 8
for _ in range(n): (loss, parts), ahead = one(ahead)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("prefetch=%d" % PRE, "train step B=%d N=%d: %.1f ms/step, %.1f scenes/s, loss %.4f, peak mem %.1f GB" % (B, N, dt * 1e3, B / dt, float(loss), torch.cuda.max_memory_allocated() / 2**30))
