"""Where does a capture of RefineTrainer's graphs fail?  python scripts/train_graph_probe3.py B N dropout(0|1) steps"""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from test_gpu_train_graphs import _trainer, _batches
B, N, drop, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
batches = _batches(2, B, N, 8100)
t = _trainer(True, 0.0, bool(drop))
np.random.seed(31)
for k in range(steps):
    total, parts = t.step(*batches[k % 2])
    torch.cuda.synchronize()
    print("step", k, float(total), "replays", t.graph_replays, parts.get("region_error"), flush=True)
