"""Gantt of the 3-stage pipeline: per batch, when each stage ran on the GPU (stream events) and
when the host was inside each stage."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
dev = torch.device("cuda:0")
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 8, 25600, device=dev)
synthetic.calibrate_score_head(score_net, pc)
np.random.seed(0)
pipe = pipeline.ForwardPipeline(score_net, region_net)
for _ in pipe.run(pc for _ in range(3)): pass
torch.cuda.synchronize()
ev = {}; host = {}
def E(): return torch.cuda.Event(enable_timing=True)
orig_geo, orig_feat, orig_reg = pipe._geometry, pipe._features, pipe._region; orig_fps = pipe._sample
cnt = {"g": 0, "f": 0, "r": 0, "s": 0}
def fps(pc):
    i = cnt["s"]; cnt["s"] += 1
    h0 = time.perf_counter(); s = E(); s.record(pipe.s_fps); out = orig_fps(pc); e = E(); e.record(pipe.s_fps)
    ev[("fps", i)] = (s, e); host[("fps", i)] = (h0, time.perf_counter()); return out
def geo(pc):
    i = cnt["g"]; cnt["g"] += 1
    h0 = time.perf_counter(); s = E(); s.record(pipe.s_geo); out = orig_geo(pc); e = E(); e.record(pipe.s_geo)
    ev[("geo", i)] = (s, e); host[("geo", i)] = (h0, time.perf_counter()); return out
def feat(item):
    i = cnt["f"]; cnt["f"] += 1
    h0 = time.perf_counter(); s = E(); s.record(pipe.s_mlp); out = orig_feat(item); e = E(); e.record(pipe.s_mlp)
    ev[("mlp", i)] = (s, e); host[("mlp", i)] = (h0, time.perf_counter()); return out
def reg(item):
    i = cnt["r"]; cnt["r"] += 1
    h0 = time.perf_counter(); s = E(); s.record(pipe.s_reg); out = orig_reg(item); e = E(); e.record(pipe.s_reg)
    ev[("reg", i)] = (s, e); host[("reg", i)] = (h0, time.perf_counter()); return out
pipe._geometry, pipe._features, pipe._region, pipe._sample = geo, feat, reg, fps
t0e = E(); t0e.record(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in pipe.run(pc for _ in range(8)): pass
torch.cuda.synchronize(); t1 = time.perf_counter()
print("total %.2f ms for 8 steps" % ((t1 - t0) * 1e3))
for k in sorted(ev, key=lambda k: (k[1], k[0])):
    s, e = ev[k]; h = host[k]
    print("%s %d: gpu %.2f -> %.2f (%.2f ms) | host %.2f -> %.2f" % (k[0], k[1], t0e.elapsed_time(s), t0e.elapsed_time(e),
          s.elapsed_time(e), (h[0] - t0) * 1e3, (h[1] - t0) * 1e3))
