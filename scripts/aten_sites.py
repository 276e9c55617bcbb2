"""Which Python lines still launch ATen kernels in one eval forward (ScoreNet with a geometry plan + region stage): torch
profiler with stacks, CUDA time per (op, innermost package frame).  usage: python scripts/aten_sites.py"""
import contextlib, io, os, sys, collections
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
dev = "cuda:0"
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 8, 25600).to(dev)
synthetic.calibrate_score_head(score_net, pc)
np.random.seed(0)


def step():
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        plan = score_net.plan(pc)
        feat, score, _ = score_net(pc, plan=plan)
        g = get_grasp_allobj(pc, score, pipeline.PARAMS, [])
        region_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, feat, pipeline.GRIPPER_PARAMS, None, [])


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_children:
        pass
    if ev.name.startswith("aten::") and ev.device_time <= 0:
        continue
    if not ev.name.startswith("aten::"):
        continue
    site = "?"
    for fr in ev.stack:
        if "regnet_for_3d_grasping_amd" in fr and "/torch/" not in fr:
            site = fr.split("regnet_for_3d_grasping_amd/")[-1]
            break
    a = agg[(ev.name, site)]
    a[0] += ev.device_time
    a[1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
print("device us per step | calls per step | op | site")
for (name, site), (t, n) in rows[:45]:
    print("%10.1f %6.1f  %-28s %s" % (t / 5, n / 5, name, site))
