"""Grasp heads (stage-2: 7 layers, refine: 5 layers) on an idle chip: heads_chain_kernel (16 rows per workgroup), heads_tree_kernel
(32 rows per workgroup, chunked trunk) and the layer-by-layer split-K path.   python scripts/bench_heads.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regnet_for_3d_grasping_amd import fused, synthetic
from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
dev = "cuda:0"
net = GripperRegionNetwork(training=True, group_num=256, gripper_num=64, grasp_score_threshold=0.5, radius=0.06, reg_channel=10)
net.load_state_dict(synthetic.seeded_state_dict(net, 11))
net = net.to(dev).eval()
rows = [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024, 4000]
MAC2, MAC3 = 256 * 1024 + 2 * (1024 * 256 + 256 * 128) + 128 * 44, 384 * 1024 + 2 * 1024 * 128 + 128 * 12
for n in rows:
    x2, x3 = torch.randn(n, 256, 1, device=dev), torch.randn(n, 384, 1, device=dev)
    line = "rows %5d:" % n
    for name, (chain, tree) in (("chain16", (True, False)), ("tree32", (False, True)), ("layerwise", (False, False))):
        old = (fused.HEADS_CHAIN, fused.HEADS_CHAIN_MAX_ROWS, fused.HEADS_TREE)
        fused.HEADS_CHAIN, fused.HEADS_TREE, fused.HEADS_CHAIN_MAX_ROWS = chain, tree, 1 << 20
        try:
            with torch.no_grad():
                for which, fn, x, mac in (("stage2", lambda: fused.twostage_forward(net.extrat_feature_region, x2, raw_reg=True), x2, MAC2),
                                          ("refine", lambda: fused.refine_forward(net.extrat_feature_refine, x3), x3, MAC3)):
                    for _ in range(3):
                        fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(20):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) / 20 * 1e3
                    line += "  %s %s %.0f us (%.1f TF)" % (name, which, us, 2 * mac * n / us / 1e6)
        finally:
            fused.HEADS_CHAIN, fused.HEADS_CHAIN_MAX_ROWS, fused.HEADS_TREE = old
    print(line)
