"""Diagnostic: wall time of the three phases of one forward step (sync after each)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.get_regiondataset import _select_score_center, _get_group_pc
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, B, 25600, device=dev)
synthetic.calibrate_score_head(score_net, pc)
np.random.seed(0)
def sync(): torch.cuda.synchronize()
for it in range(4):
    sync(); t0 = time.perf_counter()
    with torch.no_grad():
        feat, score, _ = score_net(pc)
    sync(); t1 = time.perf_counter()
    c, ci = _select_score_center(pc, score, 64, 0.5); sync(); t2 = time.perf_counter()
    gi, g = _get_group_pc(pc, c, ci, 256, 0.08, 0.01, 0.06, 0.1); sync(); t3 = time.perf_counter()
    gmi, gm = _get_group_pc(pc, c, ci, 1024, 0.08, 0.01, 0.06, 0.8); sync(); t4 = time.perf_counter()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        res = region_net(g, gm, gi, gmi, c, ci, pc, feat, pipeline.GRIPPER_PARAMS, None, [])
    sync(); t5 = time.perf_counter()
    print("iter %d: scorenet %.2f ms | centres %.2f | group256 %.2f | group1024 %.2f | GRN+refine %.2f | total %.2f"
          % (it, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3, (t5-t0)*1e3))
