"""How much of the HIP path's distance from the float64 evaluation is the LENGTH of its fp32 accumulation chains?
The layer-wise path (every fused chain off) with every layer of K >= 512 evaluated as K / 256 independent partial sums
added once (the split-K kernel of the region heads, forced on) against the same path with one accumulation chain per
output -- S8's scores against tests/golden/s8_score_fp64.npz.  Measurement only."""
import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import golden_util as gu
from regnet_for_3d_grasping_amd import fused, synthetic
from regnet_for_3d_grasping_amd._lib import lib as _L, check as _check
DEV = "cuda:0"
m7 = gu.meta_full()
s64 = np.load(os.path.join(gu.GOLDEN, "s8_score_fp64.npz"))["score"]
ref = np.load(os.path.join(gu.GOLDEN, "s8_b8_25600.npz"))["score"].astype(np.float64)
net = gu.build_scorenet_full(m7, DEV)
pc = synthetic.make_batch(2000, 8, 25600).to(DEV)
orig = fused.mlp_layer
SLAB = [256]
MINK = [512]

def split_layer(A, Ka, layer, P, pool_group=0):
    if pool_group or layer.Kpad < MINK[0]:
        return orig(A, Ka, layer, P, pool_group)
    ksplit = layer.Kpad // SLAB[0]
    out = torch.empty((P, layer.N), dtype=torch.float32, device=A.device)
    ws = torch.empty((_L.regnet_mlp_splitk_workspace_bytes(P, layer.N, ksplit),), dtype=torch.uint8, device=A.device)
    _check(_L.regnet_mlp_layer_splitk_f32(A.data_ptr(), A.stride(0), Ka, layer.W.data_ptr(), layer.Kpad, layer.scale.data_ptr(),
                                          layer.shift.data_ptr(), out.data_ptr(), out.stride(0), P, layer.N, layer.relu, ksplit,
                                          ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "splitk")
    return out

def run(tag, split, **sw):
    old = {k: getattr(fused, k) for k in sw}
    for k, v in sw.items(): setattr(fused, k, v)
    fused.mlp_layer = split_layer if split else orig
    for mod in net.modules():
        for a in ("_regnet_packed", "_regnet_sa_chain", "_regnet_sa3_chain", "_regnet_rowchain", "_regnet_head"):
            if hasattr(mod, a): delattr(mod, a)
    try:
        with torch.no_grad():
            _, s, _ = net(pc)
        s = s.cpu().numpy().astype(np.float64)
    finally:
        fused.mlp_layer = orig
        for k, v in old.items(): setattr(fused, k, v)
    print("%-70s vs fp64: max %.2e mean %.2e | vs reference: max %.2e mean %.2e" % (
        tag, np.abs(s - s64).max(), np.abs(s - s64).mean(), np.abs(s - ref).max(), np.abs(s - ref).mean()), flush=True)

print("reference (torch CPU fp32) vs fp64: max %.2e mean %.2e" % (np.abs(ref - s64).max(), np.abs(ref - s64).mean()))
LW = dict(PREMUL=False, SA3_CHAIN=False, ROWCHAIN=False, CHAIN3=False)
run("default (all chains)", False)
run("layer-wise, one accumulation chain per output", False, **LW)
run("layer-wise, K >= 512 as 256-wide partial sums", True, **LW)
SLAB[0] = 128; MINK[0] = 256
run("layer-wise, K >= 256 as 128-wide partial sums", True, **LW)
SLAB[0] = 64; MINK[0] = 128
run("layer-wise, K >= 128 as 64-wide partial sums", True, **LW)
LWP = dict(SA3_CHAIN=False, ROWCHAIN=False, CHAIN3=False)
SLAB[0] = 256; MINK[0] = 512
run("layer-wise + premul, one chain", False, **LWP)
run("layer-wise + premul, K >= 512 as 256-wide partial sums", True, **LWP)
