"""Wall time of the region stage alone (GPU otherwise idle): grouping (get_grasp_allobj) and grasp-region + refine heads,
with the host<->device synchronisation points counted (torch sync debug mode)."""
import os, sys, time, contextlib, io, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
DEV = "cuda:0"
B = int(os.environ.get("BATCH", 8))
score_net, region_net = pipeline.build_models(DEV)
pc = synthetic.make_batch(1000, B, 25600).to(DEV)
synthetic.calibrate_score_head(score_net, pc)
with torch.no_grad():
    feat, score, _ = score_net(pc)
torch.cuda.synchronize()
np.random.seed(0)
def group():
    return get_grasp_allobj(pc, score, pipeline.PARAMS, [])
def heads(g):
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        return region_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, feat, pipeline.GRIPPER_PARAMS, None, [])
for _ in range(3):
    heads(group())
torch.cuda.synchronize()
tg = th = 0.0
n = 20
for _ in range(n):
    t0 = time.perf_counter(); g = group(); torch.cuda.synchronize(); t1 = time.perf_counter()
    heads(g); torch.cuda.synchronize(); t2 = time.perf_counter()
    tg += t1 - t0; th += t2 - t1
print("batch %d: grouping %.2f ms, heads %.2f ms, total %.2f ms per batch" % (B, tg / n * 1e3, th / n * 1e3, (tg + th) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    heads(group())
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
