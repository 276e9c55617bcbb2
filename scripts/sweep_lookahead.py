"""Steady-state ms per batch of 8 as a function of the pipeline's look-ahead (max_pending_regions)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
dev = torch.device("cuda:0")
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 8, 25600, device=dev)
synthetic.calibrate_score_head(score_net, pc)
for depth in (1, 2, 3, 4, 6, 8):
    np.random.seed(0)
    pipe = pipeline.ForwardPipeline(score_net, region_net)
    for _ in pipe.run((pc for _ in range(5)), max_pending_regions=depth): pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 150
    for _ in pipe.run((pc for _ in range(n)), max_pending_regions=depth): pass
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("max_pending_regions %d: %.3f ms per batch, %.1f scenes/s" % (depth, dt / n * 1e3, 8 * n / dt))
