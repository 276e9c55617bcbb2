"""The two grasp heads alone: one launch each (fused.HEADS_CHAIN, csrc/heads.hip) against the layer-wise split-K path, microseconds
per call on an idle GPU for n rows.   python scripts/bench_heads.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regnet_for_3d_grasping_amd import fused, pipeline
dev = "cuda:0"
_, net = pipeline.build_models(dev)
fused.HEADS_CHAIN_MAX_ROWS = 1 << 20     # measure the kernel at every n (the product uses it up to 256 rows)
for n in (64, 128, 256, 449, 512, 1024):
    x2 = torch.randn(n, 256, 1, device=dev); x3 = torch.randn(n, 384, 1, device=dev)
    row = []
    for flag in (True, False):
        fused.HEADS_CHAIN = flag
        for name, fn, x in (("twostage", fused.twostage_forward, x2), ("refine", fused.refine_forward, x3)):
            mod = net.extrat_feature_region if name == "twostage" else net.extrat_feature_refine
            with torch.no_grad():
                for _ in range(5): fn(mod, x)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(50): fn(mod, x)
                e.record(); torch.cuda.synchronize()
            row.append("%s %s %6.1f us" % ("chain" if flag else "layers", name, s.elapsed_time(e) / 50 * 1e3))
    print("n=%4d | " % n + " | ".join(row))
fused.HEADS_CHAIN = True
