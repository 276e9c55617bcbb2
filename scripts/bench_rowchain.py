"""Stand-alone timing of the FP3-tail + head chain (csrc/rowchain.hip) against the same seven layers run one launch
at a time (gemm2 + score head), batch of 8 scenes x 25 600 points; random post-ReLU-like input."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regnet_for_3d_grasping_amd import fused, pipeline
DEV = "cuda:0"
P = int(os.environ.get("ROWS", 8 * 25600))
net, _ = pipeline.build_models(DEV)
seg = net.extrat_featurePN2
fp = seg.fp_modules[-1]
fp_layers = fused._packed_stack(fp, fp.mlp)
head = fused._packed_stack(seg.mlp, seg.mlp)
h1 = torch.relu(torch.randn(P, 256, device=DEV))

def chain():
    return fused.fp_head_chain(h1, seg, fp_layers, P)

def layerwise():
    h = h1
    for layer in list(fp_layers[1:]) + list(head):
        h = fused.mlp_layer(h, layer.K, layer, P)
    return fused.score_head(h, seg, P)

flop = 2.0 * P * 491520
for name, fn in (("chain", chain), ("layerwise", layerwise), ("chain", chain), ("layerwise", layerwise)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print("%-10s %.3f ms  %.1f TFLOP/s" % (name, ms, flop / ms / 1e9))

# ---- level-3 set-abstraction block (8 scenes x 256 neighbourhoods x 64): one chained launch vs the two it replaces
sa = seg.sa_modules[2]
Cf = sa.in_channels
layers = fused._packed_stack(sa, sa.mlp, lambda: torch.cat([torch.arange(3, 3 + Cf), torch.arange(0, 3)]).to(DEV))
B, M, Nsrc = 8, 256, 1024
U = torch.randn(B * Nsrc, 512, device=DEV)
V = torch.randn(B * M, 512, device=DEV) * 0.5
nbr = torch.randint(0, Nsrc, (B, M, 64), device=DEV)

def sa3_chain():
    return fused.sa3_premul_chain(U, V, nbr, sa, layers, B, Nsrc, M)

def sa3_layerwise():
    h = fused.sa_premul_layer(U, V, nbr, layers[1], B, Nsrc, M, 64)
    return fused.mlp_layer(h, layers[2].K, layers[2], B * M * 64, pool_group=64)

flop3 = 2.0 * B * M * 64 * (512 * 512 + 512 * 1024)
for name, fn in (("sa3 chain", sa3_chain), ("sa3 layerwise", sa3_layerwise), ("sa3 chain", sa3_chain), ("sa3 layerwise", sa3_layerwise)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print("%-14s %.3f ms  %.1f TFLOP/s" % (name, ms, flop3 / ms / 1e9))
