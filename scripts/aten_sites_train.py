"""Which Python lines of the REGION STAGE still launch library kernels in a training iteration (the stage is the iteration's critical
path between the forward and the trunk's backward, and host-paced): torch profiler with stacks over eager iterations, device time
and launches per (op, innermost package frame).  usage: python scripts/aten_sites_train.py"""
import collections, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
from regnet_for_3d_grasping_amd.train_step import RefineTrainer
dev = "cuda:0"
B, N = 8, 25600
batches = []
for k in range(2):
    pc = synthetic.make_batch(1000 + 8 * k, B, N)
    records = [synthetic.make_grasp_labels(pc[b].numpy(), 50 + 8 * k + b) for b in range(B)]
    target = torch.from_numpy(np.random.default_rng(2 + k).uniform(0, 1, (B, N)).astype(np.float32)).to(dev)
    batches.append((pc.to(dev), target, records))
s = ScoreNetwork(training=True); s.load_state_dict(synthetic.seeded_state_dict(s, 7))
r = GripperRegionNetwork(training=True, group_num=256, gripper_num=64, grasp_score_threshold=0.5, radius=0.06, reg_channel=10)
r.load_state_dict(synthetic.seeded_state_dict(r, 11))
synthetic.set_region_head_affine(r)
t = RefineTrainer(s.to(dev), r.to(dev), pipeline.PARAMS, pipeline.GRIPPER_PARAMS, graphs=False)
np.random.seed(1)
for i in range(3):
    t.step(*batches[i % 2])
torch.cuda.synchronize()
STEPS = 3
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
FILES = ("gripper_region_network.py", "get_regiondataset.py", "region_losses.py", "region_ops.py", "heads_train.py", "host_io.py",
         "np_random.py", "pointnet2.py")
agg = collections.Counter()


class Sites(TorchDispatchMode):
    """Every ATen operator call that touches a GPU tensor, by the innermost frame inside the region stage's files (the profiler's
    stacks are empty on this build)."""

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        if isinstance(out, torch.Tensor):
            flat.append(out)
        if any(t.is_cuda for t in flat):
            for fr in reversed(traceback.extract_stack(limit=24)):
                if "regnet_for_3d_grasping_amd" in fr.filename and fr.filename.endswith(FILES):
                    agg[("%s:%d" % (os.path.basename(fr.filename), fr.lineno), str(func).replace("aten.", ""))] += 1
                    break
        return out


VIEWS = ("view", "expand", "select", "slice", "unsqueeze", "squeeze", "transpose", "permute", "t.default", "alias", "as_strided",
         "reshape", "detach", "_unsafe_view", "unbind", "split", "empty", "is_pinned", "_local_scalar_dense", "record_stream", "resize")
with Sites():
    for i in range(STEPS):
        t.step(*batches[i % 2])
    torch.cuda.synchronize()
print("calls per iteration | op | site   (views / allocations left out)")
tot = 0
for (site, name), n in sorted(agg.items(), key=lambda kv: (kv[0][0].split(":")[0], int(kv[0][0].split(":")[1]))):
    if any(name.startswith(v) for v in VIEWS):
        continue
    print("%6.1f  %-34s %s" % (n / STEPS, name, site))
    tot += n
print("total: %.1f per iteration" % (tot / STEPS))
