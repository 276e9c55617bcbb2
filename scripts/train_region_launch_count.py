"""Training iteration, region stage: how many device activities (kernels + copies) does each part issue?  torch.profiler around
get_grasp_allobj (with labels), the region network's forward (both losses) split by a marker, and the backward of the two
losses through the region stage only.   python scripts/train_region_launch_count.py [B]"""
import contextlib, io, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
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

def count(fn):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    ks = [e for e in prof.events() if e.device_type is not None and str(e.device_type).endswith("CUDA")]
    return out, len(ks), collections.Counter((e.name[50:160] if e.name.startswith("void at::native::vectorized_elementwise_kernel") else e.name[:70]) for e in ks)

def grouping():
    with contextlib.redirect_stdout(io.StringIO()):
        return get_grasp_allobj(pc, score, pipeline.PARAMS, records)

def network(g):
    with contextlib.redirect_stdout(io.StringIO()):
        return r(g[3], g[5], g[2], g[4], g[0], g[1], pc, all_feature, pipeline.GRIPPER_PARAMS, g[6], records)

for _ in range(2):
    g = grouping(); res = network(g)
    (res[3][0].sum() + res[13][0].sum()).backward()
torch.cuda.synchronize()
g, n1, c1 = count(grouping)
print("get_grasp_allobj with labels: %d device activities" % n1)
# the network forward as a whole, then (separate passes: profilers do not nest) with one loss function profiled alone
res, n2, c2 = count(lambda: network(g))
parts = {}
for name, attr in (("compute_loss_refine (fused: region_losses.refine_loss)", "compute_loss_refine"),):
    orig = getattr(r, attr)
    def w(*a, _orig=orig, _name=name, **k):
        out, n, c = count(lambda: _orig(*a, **k))
        parts[_name] = (n, c)
        return out
    setattr(r, attr, w)
    res = network(g)
    setattr(r, attr, orig)
inner = sum(v[0] for v in parts.values())
print("region network forward: %d device activities, of which" % n2)
for k, (n, c) in parts.items():
    print("   %-44s %d" % (k, n))
print("   %-44s %d" % ("pooling, heads, crops, gathers", n2 - inner))
total = res[3][0].sum() + res[13][0].sum()
_, n3, c3 = count(lambda: total.backward())
print("backward of both losses through the region stage: %d device activities" % n3)
for name, c in (("the network forward", c2), ("the backward", c3), ("get_grasp_allobj", c1)):
    print("-- most frequent in %s" % name)
    for k, v in c.most_common(14):
        print("   %4d %s" % (v, k))
