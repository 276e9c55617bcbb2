"""Standalone timing of the FPS kernel at the ScoreNet level shapes."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from regnet_for_3d_grasping_amd import pn2_ext, synthetic
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pc = synthetic.make_batch(1000, B, 25600, device=dev)
xyz = pc.permute(0, 2, 1)[:, :3, :]
def timeit(x, M, reps=5):
    pn2_ext.farthest_point_sample(x, M); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): idx = pn2_ext.farthest_point_sample(x, M)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, idx
for N, M in ((25600, 5120), (25600, 1280), (25600, 320), (16384, 4096), (8192, 2048), (5120, 1024), (1024, 256)):
    x = xyz[:, :, :N]
    ms, idx = timeit(x, M)
    print("N=%5d M=%4d: %.3f ms  %.3f us/round" % (N, M, ms, ms * 1e3 / (M - 1)))
print("-- short runs (centre selection shapes), B=1, contiguous (P,3) rows")
for N, M in ((16215, 64), (16215, 2), (9891, 64), (25600, 64), (4000, 64)):
    x = pc[0, :N, :3].contiguous().view(1, N, 3).transpose(2, 1)
    ms, idx = timeit(x, M, reps=20)
    print("N=%5d M=%4d: %.3f ms" % (N, M, ms))
