"""How much of the features stage (the MLP stream's work for one batch) is kernel time and how much is gaps between
kernels: events around the stage vs events around every fused op inside it (no profiler attached)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from regnet_for_3d_grasping_amd import fused, pipeline, synthetic
dev = torch.device("cuda:0")
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 8, 25600, device=dev)
synthetic.calibrate_score_head(score_net, pc)
np.random.seed(0)
pipe = pipeline.ForwardPipeline(score_net, region_net)
for _ in pipe.run(pc for _ in range(5)): pass
torch.cuda.synchronize()
stage_ev, op_ev = [], []
orig_feat = pipe._features
def feat(item):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(pipe.s_mlp); out = orig_feat(item); e.record(pipe.s_mlp); stage_ev.append((s, e)); return out
pipe._features = feat
names = ["mlp_layer", "sa_chain3", "sa_premul_layer", "interp_affine", "score_head", "sa_layer12", "sa_layer1", "interp_concat"]
in_stage = {"on": False}
def wrap(name):
    f = getattr(fused, name)
    def g(*a, **k):
        if torch.cuda.current_stream() != pipe.s_mlp: return f(*a, **k)
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); out = f(*a, **k); e.record(); op_ev.append((name, s, e)); return out
    setattr(fused, name, g)
for n in names: wrap(n)
n = 40
t0 = time.perf_counter()
for _ in pipe.run(pc for _ in range(n)): pass
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
stage = sum(s.elapsed_time(e) for s, e in stage_ev) / n
ops = sum(s.elapsed_time(e) for _, s, e in op_ev) / n
print("step %.3f ms | features stage on the MLP stream %.3f ms | sum of fused ops %.3f ms (%d ops/step) | gaps + torch glue %.3f ms" % (dt, stage, ops, len(op_ev) // n, stage - ops))
