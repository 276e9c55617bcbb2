"""Per-batch cadence of ForwardPipeline over a SHORT run (the driver's 20 steps after 5 warm-up steps), no profiler: the GPU time at
which every batch's feature stage ends (timing events on the feature stream), the host time at which its geometry / feature stages
were enqueued, and when the run's last result arrived.  Shows where a 20-step run spends what 20 x the steady-state cadence does not
explain (fill, drain, stalls).     python scripts/pipeline_cadence.py [steps] [max_pending_regions]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
dev = torch.device("cuda:0")
score_net, region_net = pipeline.build_models(dev)
pcs = [synthetic.make_batch(1000 + 8 * k, 8, 25600, device=dev) for k in range(8)]
synthetic.calibrate_score_head(score_net, pcs[0])
np.random.seed(0)
synthetic.calibrate_region_head(region_net, lambda: pipeline.forward_scenes(score_net, region_net, pcs[0]))
np.random.seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
look = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pipe = pipeline.ForwardPipeline(score_net, region_net, first_launch_groups=4)
for rep in range(3):
    for _ in pipe.run((pcs[k % 8] for k in range(5)), max_pending_regions=look):
        pass
    torch.cuda.synchronize()
    log = {"geo": [], "feat": [], "end": [], "out": []}
    og, of = pipe._geometry, pipe._features
    def geo(item):
        log["geo"].append(time.perf_counter()); return og(item)
    def feat(item):
        log["feat"].append(time.perf_counter())
        out = of(item)
        e = torch.cuda.Event(enable_timing=True); e.record(pipe.s_mlp); log["end"].append(e)
        return out
    pipe._geometry, pipe._features = geo, feat
    base = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t_base = time.perf_counter(); base.record(pipe.s_mlp)
    for _ in pipe.run((pcs[k % 8] for k in range(n)), max_pending_regions=look):
        log["out"].append(time.perf_counter())
    torch.cuda.synchronize(); t_end = time.perf_counter()
    pipe._geometry, pipe._features = og, of
    end = [base.elapsed_time(e) for e in log["end"]]
    print("run %d: %d steps in %.2f ms = %.3f ms per step; first feature stage ends at %.2f ms, last at %.2f, last result %.2f ms later" % (
        rep, n, (t_end - t_base) * 1e3, (t_end - t_base) * 1e3 / n, end[0], end[-1], (t_end - t_base) * 1e3 - end[-1]))
    print("   cadence (ms):", " ".join("%.2f" % (b - a) for a, b in zip(end, end[1:])))
    print("   feature stage enqueued before the previous one ended by (ms):", " ".join("%.1f" % (end[j - 1] - (log["feat"][j] - t_base) * 1e3) for j in range(1, n)))
    print("   results handed out at (ms):", " ".join("%.1f" % ((t - t_base) * 1e3) for t in log["out"]))
