"""Which lines of the package still call ATen operators on GPU tensors in one steady-state forward step (ScoreNet + region
stage, 8 x 25 600): a TorchDispatchMode logs every operator with the innermost package frame of its Python stack.
    python scripts/aten_dispatch_sites.py [--train]
Prints calls per step and output bytes per step per (operator, file:line) -- the work list of "take the ATen glue out of the
forward path" (VERDICT r4 #6); device durations are in the rocprofv3 kernel stats."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from regnet_for_3d_grasping_amd import pipeline, synthetic

dev = "cuda:0"
score_net, region_net = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 8, 25600).to(dev)
synthetic.calibrate_score_head(score_net, pc)
synthetic.calibrate_region_head(region_net, lambda: pipeline.forward_scenes(score_net, region_net, pc))
np.random.seed(0)
for _ in range(3):
    pipeline.forward_scenes(score_net, region_net, pc)
torch.cuda.synchronize()

log = collections.defaultdict(lambda: [0, 0])
SKIP = ("aten.view", "aten.detach", "aten._unsafe_view", "aten.alias", "aten.expand", "aten.slice", "aten.select", "aten.transpose",
        "aten.permute", "aten.unsqueeze", "aten.squeeze", "aten.as_strided", "aten.t.", "aten.reshape", "aten.unbind", "aten.split",
        "aten.empty", "aten.sym_", "aten._local_scalar_dense", "aten.is_pinned", "aten.lift_fresh", "aten.unfold")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith(SKIP):
            return out
        tensors = [a for a in list(args) + [out] if isinstance(a, torch.Tensor)]
        if not any(t.is_cuda for t in tensors):
            return out
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=24)):
            if "regnet_for_3d_grasping_amd" in fr.filename:
                site = "%s:%d" % (fr.filename.split("regnet_for_3d_grasping_amd/")[-1], fr.lineno)
                break
        nbytes = out.numel() * out.element_size() if isinstance(out, torch.Tensor) else 0
        e = log[(name, site)]
        e[0] += 1
        e[1] += nbytes
        return out


STEPS = 2
with Log():
    for _ in range(STEPS):
        pipeline.forward_scenes(score_net, region_net, pc)
torch.cuda.synchronize()
rows = sorted(log.items(), key=lambda kv: (-kv[1][0], kv[0]))
print("calls/step  out KB/step  operator                          site")
for (name, site), (n, b) in rows:
    print("%8.1f %11.1f  %-33s %s" % (n / STEPS, b / STEPS / 1024.0, name, site))
print("total ATen operator calls on GPU tensors per step: %.1f" % (sum(v[0] for v in log.values()) / STEPS))
