"""Standalone timing of 3-NN / ball query / radius group at the ScoreNet level shapes (B scenes)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from regnet_for_3d_grasping_amd import pn2_ext, region_ops, synthetic
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pc = synthetic.make_batch(1000, B, 25600, device=dev)
xyz = pc.permute(0, 2, 1)[:, :3, :]
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
ctr1 = pn2_ext.farthest_point_sample(xyz, 5120)
x1 = torch.gather(xyz, 2, ctr1[:, None, :].expand(B, 3, 5120))
ctr2 = pn2_ext.farthest_point_sample(x1, 1024)
x2 = torch.gather(x1, 2, ctr2[:, None, :].expand(B, 3, 1024))
for name, thr in (("exhaustive", 1 << 40), ("grid", 2048)):
    pn2_ext.GRID_MIN_POINTS = pn2_ext.GRID_MIN_POINTS_BALL = thr
    print(name)
    print("  three_nn  Q25600 K5120: %.3f ms" % timeit(lambda: pn2_ext.point_search(xyz, x1, 3)))
    print("  three_nn  Q5120  K1024: %.3f ms" % timeit(lambda: pn2_ext.point_search(x1, x2, 3)))
    print("  ball_query N25600 M5120 r.02: %.3f ms" % timeit(lambda: pn2_ext.ball_query(xyz, x1, 0.02, 64)))
    print("  ball_query N5120 M1024 r.08: %.3f ms" % timeit(lambda: pn2_ext.ball_query(x1, x2, 0.08, 64)))
c = pc[:, :64, :].contiguous()
print("radius_group r.008: %.3f ms, r.064: %.3f ms" % (timeit(lambda: region_ops.radius_candidates(pc, c, 0.008)),
                                                      timeit(lambda: region_ops.radius_candidates(pc, c, 0.064))))
