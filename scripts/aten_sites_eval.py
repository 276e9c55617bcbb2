"""ATen operator calls of one eval forward (ScoreNet with a plan + centre selection + grouping + region / refine networks) by call
site, through a dispatch mode (the profiler's stacks are empty on this build).  usage: python scripts/aten_sites_eval.py"""
import collections, contextlib, io, os, sys, traceback
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
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
        return region_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, feat, pipeline.GRIPPER_PARAMS, None, [])


res = step()
synthetic.calibrate_region_head(region_net, step)
for _ in range(2):
    step()
torch.cuda.synchronize()
agg = collections.Counter()
VIEWS = ("view", "expand", "select", "slice", "unsqueeze", "squeeze", "transpose", "permute", "t.default", "alias", "as_strided",
         "reshape", "detach", "_unsafe_view", "unbind", "split", "empty", "is_pinned", "_local_scalar_dense", "record_stream", "resize")


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        if isinstance(out, torch.Tensor):
            flat.append(out)
        if any(t.is_cuda for t in flat):
            for fr in reversed(traceback.extract_stack(limit=24)):
                if "regnet_for_3d_grasping_amd" in fr.filename:
                    agg[("%s:%d" % (os.path.basename(fr.filename), fr.lineno), str(func).replace("aten.", ""))] += 1
                    break
        return out


STEPS = 3
with Sites():
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
tot = 0
print("calls per step | op | site   (views / allocations left out)")
for (site, name), n in sorted(agg.items(), key=lambda kv: (kv[0][0].split(":")[0], int(kv[0][0].split(":")[1]))):
    if any(name.startswith(v) for v in VIEWS):
        continue
    print("%6.1f  %-30s %s" % (n / STEPS, name, site))
    tot += n
print("total: %.1f per step" % (tot / STEPS))
