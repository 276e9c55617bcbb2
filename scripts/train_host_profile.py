"""Host-side cost of a training iteration: cProfile of RefineTrainer.step (no device syncs added), top functions by own
time, plus the host time of the ScoreNet forward alone when the device is not waited for.  python train_host_profile.py [B]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
from regnet_for_3d_grasping_amd.train_step import RefineTrainer
dev = "cuda:0"
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 25600
pc = synthetic.make_batch(8100, B, N)
records = [synthetic.make_grasp_labels(pc[b].numpy(), 50 + b) for b in range(B)]
target = torch.from_numpy(np.random.default_rng(2).uniform(0, 1, (B, N)).astype(np.float32)).to(dev)
s = ScoreNetwork(training=True); s.load_state_dict(synthetic.seeded_state_dict(s, 3))
r = GripperRegionNetwork(training=True, group_num=256, gripper_num=64, grasp_score_threshold=0.5, radius=0.06, reg_channel=10)
r.load_state_dict(synthetic.seeded_state_dict(r, 4))
t = RefineTrainer(s.to(dev), r.to(dev), pipeline.PARAMS, pipeline.GRIPPER_PARAMS)
pc = pc.to(dev)
np.random.seed(1)
ahead = t.prefetch(pc)
for _ in range(4):
    nxt = t.prefetch(pc); t.step(pc, target, records, plan=ahead); ahead = nxt
torch.cuda.synchronize()
# host time of the forward alone (enqueue only)
plan = t.geometry.acquire(t.prefetch(pc), pc.device); torch.cuda.synchronize()
t.score_net.train()
t0 = time.perf_counter()
with torch.enable_grad():
    out = t.score_net(pc, target, None, plan=plan)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("ScoreNet training forward: host enqueue %.2f ms, device finished %.2f ms after that" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
loss = out[2].sum()
t0 = time.perf_counter(); loss.backward(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("ScoreNet backward: host %.2f ms, device finished %.2f ms after that" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
pr = cProfile.Profile()
ahead = t.prefetch(pc)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    nxt = t.prefetch(pc); t.step(pc, target, records, plan=ahead); ahead = nxt
torch.cuda.synchronize()
pr.disable()
print("5 iterations under cProfile: %.1f ms each" % ((time.perf_counter() - t0) / 5 * 1e3))
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(30)
st.print_callees("compute_loss_refine")
st.print_callees("loss.py")
st.print_callees("functional.py.*cross_entropy")
