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
