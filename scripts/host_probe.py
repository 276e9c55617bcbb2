"""Is the driver thread of ForwardPipeline ahead of the GPU?  Logs, at every enqueue of a feature stage, whether the
previous batch's feature stage had already finished (stream idle = the host is late) and how long each stage's enqueue
took on the host.    python host_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
dev = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 8, 25600)
synthetic.calibrate_score_head(score_net, pc.to(dev))
pc = pc.to(dev)
pipe = pipeline.ForwardPipeline(score_net, region_net)
log = {"late": 0, "n": 0, "t": {"sample_group": 0.0, "geometry": 0.0, "features": 0.0}, "prev": None}
for name in ("_sample_group", "_geometry", "_features"):
    fn = getattr(pipe, name)
    def timed(item, fn=fn, name=name):
        t0 = time.perf_counter()
        if name == "_features":
            log["n"] += 1
            if log["prev"] is not None and log["prev"].query():
                log["late"] += 1
        out = fn(item)
        if name == "_features":
            log["prev"] = out["mlp_done"]
        log["t"][name[1:]] += time.perf_counter() - t0
        return out
    setattr(pipe, name, timed)
np.random.seed(0)
for _ in pipe.run((pc for _ in range(10))): pass
torch.cuda.synchronize()
log.update(late=0, n=0); log["t"] = {k: 0.0 for k in log["t"]}
t0 = time.perf_counter()
for _ in pipe.run((pc for _ in range(steps))): pass
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%.3f ms/step; feature stage enqueued onto an IDLE stream %d of %d times" % (dt / steps * 1e3, log["late"], log["n"]))
print("host enqueue time per step: " + ", ".join("%s %.2f ms" % (k, v / steps * 1e3) for k, v in log["t"].items()))
import sys as _s; print("switch interval", _s.getswitchinterval())
