"""Why does the feature stream idle between two batches?  For every batch: host time when its feature stage was enqueued
(start / end of the enqueue), device time when its first / last kernel ran (events mapped onto the host clock), and the
device time its geometry finished.  idle = first kernel start - previous batch's last kernel end; it is "host-bound"
when the first kernel starts right when its launch was enqueued, "geometry-bound" when it starts when geo_done fires."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regnet_for_3d_grasping_amd import pipeline, synthetic
DEV = "cuda:0"
STEPS = int(os.environ.get("STEPS", 60))
score_net, region_net = pipeline.build_models(DEV)
pc = synthetic.make_batch(1000, 8, 25600).to(DEV)
synthetic.calibrate_score_head(score_net, pc)
np.random.seed(1)
WITH_REGION = os.environ.get("WITH_REGION", "1") != "0"       # 0: the pipeline without its region stage (bench.py --score-only)
if WITH_REGION and os.environ.get("CALIB", "1") != "0":      # 1: region head calibrated, the refine network runs (round 4)
    synthetic.calibrate_region_head(region_net, lambda: pipeline.forward_scenes(score_net, region_net, pc))
rec = []

class Probe(pipeline.ForwardPipeline):
    def _geometry(self, item):
        item = super()._geometry(item)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(self.s_geo)
        item["geo_ev"] = ev
        item["geo_enq"] = time.perf_counter()
        return item

    def _features(self, item):
        s_mlp = self.s_mlps[self._n_featured % len(self.s_mlps)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        s_mlp.wait_event(item["geo_done"])
        e0.record(s_mlp)
        geo_ev, geo_enq = item.pop("geo_ev"), item.pop("geo_enq")
        item = super()._features(item)
        e1.record(s_mlp)
        rec.append((t0, time.perf_counter(), e0, e1, geo_ev, geo_enq))
        return item

pipe = Probe(score_net, region_net, with_region=WITH_REGION, first_launch_groups=4)
for _ in pipe.run((pc for _ in range(5))):
    pass
torch.cuda.synchronize()
rec.clear()
ref = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
ref.record(pipe.s_mlp)
torch.cuda.synchronize()
t_ref = time.perf_counter()
for _ in pipe.run((pc for _ in range(STEPS))):
    pass
torch.cuda.synchronize()
rows = []
for t0, t1, e0, e1, geo_ev, geo_enq in rec:
    rows.append(((t0 - t_ref) * 1e3, (t1 - t_ref) * 1e3, ref.elapsed_time(e0), ref.elapsed_time(e1), ref.elapsed_time(geo_ev), (geo_enq - t_ref) * 1e3))
print("batch | enqueue start..end (host ms) | device start..end (ms) | geometry done (ms) | idle before (ms) | start - enqueue_start | start - geo_done")
idle_tot = 0
for i, (h0, h1, d0, d1, g, genq) in enumerate(rows):
    idle = d0 - rows[i - 1][3] if i else 0.0
    if i >= 25: idle_tot += idle
    if i < 40:
        print("%3d | %8.2f .. %8.2f | %8.2f .. %8.2f | %8.2f (enq %8.2f) | %6.3f | %7.3f | %7.3f" % (i, h0, h1, d0, d1, g, genq, idle, d0 - h0, d0 - g))
n = len(rows) - 25
print("steady state (batches 25..): mean idle %.3f ms per step, mean step %.3f ms" % (idle_tot / n, (rows[-1][3] - rows[24][3]) / n))
