"""test.py-scale inference of ONE scene (test.py:68-71: 4000 centres, 256- / 2048-point groups): phase wall times.
python testpy_scale.py [N]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regnet_for_3d_grasping_amd import eval_collision, pipeline, synthetic
from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
params = [4000, 0.5, 256, 0.1, 2048, 0.8, 0.08, 0.01, 0.06]
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 1, N)
synthetic.calibrate_score_head(score_net, pc.to(dev))
pc = pc.to(dev)
np.random.seed(0)
def run():
    t = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        all_feature, score, _ = score_net(pc)
        torch.cuda.synchronize(); t["scorenet"] = time.perf_counter() - t0; t0 = time.perf_counter()
        g = get_grasp_allobj(pc, score, params, [])
        torch.cuda.synchronize(); t["centres + grouping"] = time.perf_counter() - t0; t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            res = region_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, all_feature, pipeline.GRIPPER_PARAMS, None, [])
        torch.cuda.synchronize(); t["region_net + refine"] = time.perf_counter() - t0; t0 = time.perf_counter()
        kept = eval_collision.eval_test(pc[0, :, :3], res[0][:, :8], None, 0.75, pipeline.DEPTH, pipeline.WIDTH, 0)
        torch.cuda.synchronize(); t["collision filter (test.py:147)"] = time.perf_counter() - t0
    return t, int((score > 0.5).sum()), (res[0], kept)
for _ in range(2): run()
# per-call detail of the grouping stage (each wrapped call is followed by a device sync)
from regnet_for_3d_grasping_amd import get_regiondataset as G, np_random, region_ops
detail = {}
def wrap(mod, name, label):
    fn = getattr(mod, name)
    def timed(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize(); detail[label] = detail.get(label, 0.0) + time.perf_counter() - t0
        return out
    setattr(mod, name, timed)
    return fn
saved = [(m, n, wrap(m, n, l)) for m, n, l in [(region_ops, "select_positive", "select_positive"),
         (G._F, "farthest_point_sample", "centre FPS"), (region_ops, "radius_candidates", "radius_candidates"),
         (np_random, "choice_rows", "numpy-stream draws (host)"), (G, "_get_group_pc", "_get_group_pc total")]]
run()
for m, n, fn in saved: setattr(m, n, fn)
for k, v in detail.items(): print("   %-28s %8.2f ms" % (k, v * 1e3))
acc = {}
for _ in range(5):
    t, npos, res = run()
    for k, v in t.items(): acc[k] = acc.get(k, 0) + v / 5
print("N=%d, %d points score > 0.5, %d centres, %d grasps after refine, %d without collision" %
      (N, npos, params[0], res[0].shape[0], res[1].shape[0]))
for k, v in acc.items(): print("%-24s %8.2f ms" % (k, v * 1e3))
print("%-24s %8.2f ms" % ("total", sum(acc.values()) * 1e3))
