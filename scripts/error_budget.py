"""Where the HIP path's distance from the reference comes from: ScoreNet scores of the 8 S8 scenes against the float64
evaluation of the same graph (tests/golden/s8_score_fp64.npz) and against the reference's fp32 scores, with the fused
path's re-associations switched off one at a time."""
import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import golden_util as gu
from regnet_for_3d_grasping_amd import fused, synthetic
DEV = "cuda:0"
m7 = gu.meta_full()
s64 = np.load(os.path.join(gu.GOLDEN, "s8_score_fp64.npz"))["score"]
ref = np.load(os.path.join(gu.GOLDEN, "s8_b8_25600.npz"))["score"].astype(np.float64)
net = gu.build_scorenet_full(m7, DEV)
pc = synthetic.make_batch(2000, 8, 25600).to(DEV)
print("reference (torch CPU fp32) vs fp64: max %.2e mean %.2e" % (np.abs(ref - s64).max(), np.abs(ref - s64).mean()))
def run(tag, **sw):
    old = {k: getattr(fused, k) for k in sw}
    for k, v in sw.items(): setattr(fused, k, v)
    for mod in net.modules():
        for a in ("_regnet_packed", "_regnet_sa_chain", "_regnet_sa3_chain", "_regnet_rowchain", "_regnet_head"):
            if hasattr(mod, a): delattr(mod, a)
    try:
        with torch.no_grad():
            _, s, _ = net(pc)
        s = s.cpu().numpy().astype(np.float64)
    finally:
        for k, v in old.items(): setattr(fused, k, v)
    print("%-58s vs fp64: max %.2e mean %.2e | vs reference: max %.2e mean %.2e" % (
        tag, np.abs(s - s64).max(), np.abs(s - s64).mean(), np.abs(s - ref).max(), np.abs(s - ref).mean()))
run("default (all chains, pre-multiplied first layers)")
run("no pre-multiplied first layers (PREMUL=False)", PREMUL=False)
run("pre-multiplied, coordinates NOT centred", PREMUL_CENTRE=False)
run("no SA3 chain", SA3_CHAIN=False)
run("no rowchain (SA2 / FP3+head layer-wise)", ROWCHAIN=False)
run("no level-1 chain (CHAIN3=False)", CHAIN3=False)
run("no interp prologue", FP_HEAD_INTERP=False)
run("everything layer-wise, no premul", PREMUL=False, SA3_CHAIN=False, ROWCHAIN=False, CHAIN3=False)
fused.ENABLED = False
with torch.no_grad():
    _, s, _ = net(pc)
s = s.cpu().numpy().astype(np.float64)
print("%-58s vs fp64: max %.2e mean %.2e | vs reference: max %.2e mean %.2e" % ("operator-granular path (torch convs on the GPU)", np.abs(s - s64).max(), np.abs(s - s64).mean(), np.abs(s - ref).max(), np.abs(s - ref).mean()))
