"""One scene alone through the forward: where its latency goes.  Device-synchronised wall time of every phase (so the
overlap a pipeline could create is lost on purpose: this is the dependency chain): level-1 / 2 / 3 sampling, the rest of
the geometry (3 ball queries, 3 3-NN searches, the per-centre first-layer terms), the feature stage, the region stage
(centres, groups, heads, crops, refine); and what intra-scene chunking of level 1 could hide at most: the level-1 ball
query + the level-1 block, the only consumers that need level-1 picks only.   python scripts/latency_phases.py [B] [N]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regnet_for_3d_grasping_amd import fused, pipeline, synthetic
from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25600
CHAIN = os.environ.get("NO_CHAIN") != "1"     # NO_CHAIN=1: levels 2-3 sample for real (rounds 1-3)
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, B, N, device=dev)
synthetic.calibrate_score_head(score_net, pc)
synthetic.calibrate_region_head(region_net, lambda: pipeline.forward_scenes(score_net, region_net, pc))
np.random.seed(0)
seg = score_net.extrat_featurePN2
pts = pc[:, :, :6].permute(0, 2, 1)
acc = {}
def lap(name, t0):
    torch.cuda.synchronize()
    t = time.perf_counter()
    acc.setdefault(name, []).append((t - t0) * 1e3)
    return time.perf_counter()
reps = 12
with torch.no_grad():
    for it in range(reps + 3):
        if it == 3:
            acc.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter(); start = t0
        xyz = pts[:, :3, :]
        ctrs = []
        first_tie = None
        for lvl, sa in enumerate(seg.sa_modules):
            ctr, first_tie = fused.sa_sample(sa, xyz, first_tie if CHAIN else None)
            ctrs.append(ctr)
            xyz = torch.gather(xyz, 2, ctr[:, None, :].expand(xyz.shape[0], 3, ctr.shape[1]))
            t0 = lap("sampling level %d" % (lvl + 1), t0)
        geo1 = fused.sa_group(seg.sa_modules[0], pts[:, :3, :], ctrs[0])
        t0 = lap("level-1 centroid gather + ball query", t0)
        new_xyz, feat1 = fused.sa_features(seg.sa_modules[0], pts[:, :3, :], pts[:, 3:6, :], geo1)
        t0 = lap("level-1 block (sa_chain_kernel)", t0)
        plan = score_net.plan(pc, ctrs)
        t0 = lap("whole geometry plan given the picks (incl. level 1 again)", t0)
        all_feature, score, _ = score_net(pc, plan=plan)
        t0 = lap("feature stage (incl. level-1 block again)", t0)
        g = get_grasp_allobj(pc, score, pipeline.PARAMS, [])
        t0 = lap("region: centres + both groupings", t0)
        with contextlib.redirect_stdout(io.StringIO()):
            res = region_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, all_feature, pipeline.GRIPPER_PARAMS, None, [])
        t0 = lap("region: heads + crops + refine", t0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(1):
            pipeline.forward_scenes(score_net, region_net, pc)
        torch.cuda.synchronize()
        acc.setdefault("forward_scenes (no per-phase syncs)", []).append((time.perf_counter() - t) * 1e3)
print("B=%d N=%d, median of %d runs, ms:" % (B, N, reps))
for k, v in acc.items():
    print("  %-62s %7.3f" % (k, float(np.median(v))))
