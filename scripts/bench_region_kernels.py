"""The region stage's kernels one at a time on an otherwise idle GPU, at the shapes of a bench step (8 x 25 600 points, 64
centres per scene): microseconds per launch by HIP events around REPS back-to-back launches, and the HBM rate of the
algorithmic bytes.  Inside the pipeline the same launches wait for CUs beside the persistent chains (that wait is inside
rocprofv3's duration column); this is what the kernels themselves cost.
    python scripts/bench_region_kernels.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from regnet_for_3d_grasping_amd import pn2_ext, region_ops, synthetic

dev = "cuda:0"
REPS = int(os.environ.get("REPS", 50))
B, N, NC = 8, 25600, 64


def timed(fn, name, bytes_moved, note=""):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / REPS * 1e3
    print("%-44s %8.1f us   %7.1f MB   %6.2f TB/s  %s" % (name, us, bytes_moved / 1e6, bytes_moved / us / 1e6, note))


torch.manual_seed(0)
pc = synthetic.make_batch(1000, B, N).to(dev)                       # (B, N, 6)
for share in (0.5, 0.1):
    score = (torch.rand(B, N, device=dev) < share).float()
    idx, xyz, count = region_ops.select_positive(pc, score, 0.5)
    kept = int(count.sum())
    timed(lambda: region_ops.select_positive(pc, score, 0.5), "select_positive  B8 N25600  %2d %% positive" % (100 * share),
          B * N * 4 + kept * (12 + 8 + 12), "(incl. three torch.empty)")

feat = torch.randn(B * N, 256, device=dev)
for G, R_scene in ((256, NC), (64, NC - 8)):
    local = torch.randint(0, N, (B * NC, G), device=dev)
    row_ids = None if R_scene == NC else torch.arange(B * NC, device=dev).view(B, NC)[:, :R_scene].reshape(-1).contiguous()
    R = B * R_scene
    timed(lambda: region_ops.gather_max_scene(feat, local, row_ids, NC, N), "gather_max_scene R%d G%d F256" % (R, G),
          R * G * 256 * 4 + R * G * 8 + R * 256 * 4, "(random rows of a 210 MB table)")
glob = (torch.randint(0, N, (B * NC, 256), device=dev) + (torch.arange(B * NC, device=dev) // NC * N).view(-1, 1)).contiguous()
timed(lambda: region_ops.gather_max(feat, glob), "gather_max       R512 G256 F256", 512 * 256 * (1024 + 8) + 512 * 1024)

centres = pc[:, :NC, :3].contiguous()
for radius in (0.008, 0.064):
    cand, cnt = region_ops.radius_candidates(pc, centres, radius)
    timed(lambda: region_ops.radius_candidates(pc, centres, radius), "radius_candidates r=%.3f (cap %d)" % (radius, cand.size(2)),
          B * N * 12 + int(cnt.sum()) * 4, "mean members %.0f" % float(cnt.float().mean()))

cnt1 = torch.randint(1, 65, (B * 5120,), device=dev)
timed(lambda: pn2_ext.class_order(cnt1), "class_order      n40960", 40960 * 8 * 3)
pts = pc[:, :, :3].permute(0, 2, 1)
ind = torch.randint(0, N, (B, 5120), device=dev)
timed(lambda: pn2_ext.gather_points(pts, ind), "gather_points    B8 C3 M5120 (strided src)", B * 5120 * (8 + 24))
