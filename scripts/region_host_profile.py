"""cProfile of the region stage's host side at a small batch (the stage that paces batches of 1-4 scenes).
python scripts/region_host_profile.py [B] [iterations]"""
import cProfile, contextlib, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(500, B, 25600).to(dev)
synthetic.calibrate_score_head(score_net, pc)
np.random.seed(3)
with torch.no_grad():
    all_feature, score, _ = score_net(pc)

    def stage():
        (center_pc, center_idx, g_idx, g, gm_idx, gm, _) = get_grasp_allobj(pc, score, pipeline.PARAMS, [])
        with contextlib.redirect_stdout(io.StringIO()):
            return region_net(g, gm, g_idx, gm_idx, center_pc, center_idx, pc, all_feature, pipeline.GRIPPER_PARAMS, None, [])
    for _ in range(10):
        stage()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        stage()
    torch.cuda.synchronize()
    print("region stage alone, B=%d: %.3f ms per batch" % (B, (time.perf_counter() - t0) / iters * 1e3))
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(iters):
        stage()
    prof.disable()
    torch.cuda.synchronize()
out = io.StringIO()
st = pstats.Stats(prof, stream=out)
st.sort_stats("tottime").print_stats(45)
print(out.getvalue().replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", ""))
out = io.StringIO()
pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(40)
print(out.getvalue().replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", ""))
