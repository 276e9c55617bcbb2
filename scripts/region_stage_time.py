"""Wall time of the region stage alone (GPU otherwise idle): grouping (get_grasp_allobj) and grasp-region + refine heads,
with the host<->device synchronisation points counted (torch sync debug mode)."""
import os, sys, time, contextlib, io, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regnet_for_3d_grasping_amd import get_regiondataset, np_random, pipeline, synthetic
from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
get_regiondataset.DEVICE_DRAWS = os.environ.get("DEVICE_DRAWS", "1") != "0"   # 0: round-2 host draws
defer = np_random.deferred() if os.environ.get("DEFER", "1") != "0" else contextlib.nullcontext()
defer.__enter__()   # as the pipeline's region worker: numpy's generator stays on the device between stages
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
print("device draws %s, deferred hand-back %s" % (get_regiondataset.DEVICE_DRAWS, os.environ.get("DEFER", "1") != "0"))
if get_regiondataset.DEVICE_DRAWS:
    for size, mode, cap, cnt in ((256, 0, 25600, None), (1024, 0, 25600, None)):
        from regnet_for_3d_grasping_amd import region_ops
        g = group()
        radius = get_regiondataset.group_radius(pipeline.WIDTH, pipeline.HEIGHT, pipeline.DEPTH, pipeline.R_TIME_GROUP if size == 256 else pipeline.R_TIME_GROUP_MORE)
        cand, counts = region_ops.radius_candidates(pc, g[0], radius)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            np_random.choice_rows_device(counts, size, mode, cap)
        e1.record(); torch.cuda.synchronize()
        c = counts.cpu().numpy()
        print("choice_rows_device size %d: %.3f ms per call (%d rows, counts mean %.0f max %d, %d rows without replacement)" % (
            size, e0.elapsed_time(e1) / 10, c.size, c.mean(), c.max(), int((c >= size).sum())))
print("batch %d: grouping %.2f ms, heads %.2f ms, total %.2f ms per batch" % (B, tg / n * 1e3, th / n * 1e3, (tg + th) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    heads(group())
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
