"""How many launches is the eval-mode region stage?  torch.profiler's kernel list for ONE region stage (centres, groups, heads,
crops, refine) of a batch, grouped by the Python function that issued them (cProfile of the same call beside it).
python scripts/region_launch_count.py [B]"""
import contextlib, io, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, B, 25600, device=dev)
synthetic.calibrate_score_head(score_net, pc)
synthetic.calibrate_region_head(region_net, lambda: pipeline.forward_scenes(score_net, region_net, pc))
np.random.seed(0)
with torch.no_grad():
    feat, score, _ = score_net(pc)

def stage(tag=None):
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        g = get_grasp_allobj(pc, score, pipeline.PARAMS, [])
        if tag: torch.cuda.synchronize()
        res = region_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, feat, pipeline.GRIPPER_PARAMS, None, [])
    return res
for _ in range(3): stage()
torch.cuda.synchronize()
for part in ("grouping", "network"):
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        g = get_grasp_allobj(pc, score, pipeline.PARAMS, [])
        torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            if part == "grouping":
                g = get_grasp_allobj(pc, score, pipeline.PARAMS, [])
            else:
                res = region_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, feat, pipeline.GRIPPER_PARAMS, None, [])
        torch.cuda.synchronize()
    ks = [e for e in prof.events() if e.device_type is not None and str(e.device_type).endswith("CUDA")]
    cnt = collections.Counter(e.name[:70] for e in ks)
    print("== %s: %d device activities (kernels + copies)" % (part, len(ks)))
    for n, c in cnt.most_common(40):
        print("   %3d  %s" % (c, n))
