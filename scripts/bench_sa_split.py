"""The split-products experiment on the level-1 set-abstraction block (csrc/sa_split.hip, fused.SPLIT_PRODUCTS) against the exact-fp32
sa_chain_kernel: error of both against a float64 evaluation of the same packed layers, and time per launch at 8 x 25 600.
    python scripts/bench_sa_split.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regnet_for_3d_grasping_amd import fused, pipeline, synthetic
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
score_net, _ = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, B, 25600).to(dev)
synthetic.calibrate_score_head(score_net, pc)
seg = score_net.extrat_featurePN2
sa = seg.sa_modules[0]
points = pc[:, :, :6].permute(0, 2, 1)
xyz, feat = points[:, :3, :], points[:, 3:6, :]
with torch.no_grad():
    geo = fused.sa_geometry(sa, xyz)
    Cf = 3
    layers = fused._packed_stack(sa, sa.mlp, lambda: torch.cat([torch.arange(3, 3 + Cf), torch.arange(0, 3)]).to(dev))
    l1, l2, l3 = layers
    M, K = sa.num_centroids, 64

    def run(split, skip=True):
        fused.SPLIT_PRODUCTS = split
        try:
            return fused.sa_chain3(feat, xyz, geo["nbr"], geo["ctr"], l1, l2, l3, B, M, K, geo.get("count") if skip else None,
                                   geo.get("order") if skip else None)
        finally:
            fused.SPLIT_PRODUCTS = False

    exact, split = run(False), run(True)
    # float64 evaluation of the first two scenes
    nb = min(B, 2)
    nbr, ctr = geo["nbr"][:nb], geo["ctr"][:nb]
    x64, f64 = xyz[:nb].double(), feat[:nb].double()
    idx = nbr.reshape(nb, 1, M * K)
    gx = torch.gather(x64, 2, idx.expand(nb, 3, M * K)).view(nb, 3, M, K) - torch.gather(x64, 2, ctr[:, None, :].expand(nb, 3, M)).unsqueeze(-1)
    gf = torch.gather(f64, 2, idx.expand(nb, 3, M * K)).view(nb, 3, M, K)
    h = torch.cat([gf, gx], 1).permute(0, 2, 3, 1).reshape(-1, 6)            # columns [feature | xyz], as the packed first layer
    for L in (l1, l2, l3):
        W = (L.W8[:, :L.K] if L.W8 is not None else L.W[:L.N, :L.K]).double()
        h = h @ W.t() * L.scale.double() + L.shift.double()
        if L.relu:
            h = torch.relu(h)
    ref = h.view(nb * M, K, -1).max(1)[0]
    n = nb * M
    for name, out in (("exact fp32 (sa_chain_kernel)", exact), ("split products (sa_chain_split_kernel)", split)):
        d = (out[:n].double() - ref).abs()
        print("%-42s max|err| vs float64 %.3e  mean %.3e   (|ref| max %.2f)" % (name, float(d.max()), float(d.mean()), float(ref.abs().max())))
    print("split vs exact: max|diff| %.3e" % float((split - exact).abs().max()))

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    print("time per launch at %d x 25 600: exact with tile skipping %.3f ms, exact without %.3f ms; split with (<= 32-member halves) %.3f ms, without %.3f ms" % (
        B, timed(lambda: run(False)), timed(lambda: run(False, False)), timed(lambda: run(True)), timed(lambda: run(True, False))))
