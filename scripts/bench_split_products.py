"""The split-products experiment (csrc/tsplit.hip, conv1x1_train.SPLIT_PRODUCTS) against the exact-fp32 kernels of csrc/tgemm.hip:
error against a float64 evaluation and time per launch, forward / forward on a pending BatchNorm / input gradient, at the
training iteration's layer shapes (8 scenes).     python scripts/bench_split_products.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regnet_for_3d_grasping_amd import conv1x1_train as c
dev = "cuda:0"
torch.manual_seed(0)
SHAPES = [(8, 128, 128, 327680), (8, 256, 128, 327680), (8, 256, 256, 65536), (8, 512, 256, 65536), (8, 512, 512, 16384),
          (8, 1024, 512, 16384), (8, 256, 272, 25600), (8, 512, 256, 25600), (8, 256, 512, 25600), (8, 1024, 1024, 1024)]


def timed(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("%-28s %-8s %9s %9s %7s   %10s %10s" % ("B Co Ci L", "op", "fp32 ms", "split ms", "ratio", "fp32 err", "split err"))
tot = [0.0, 0.0]
for B, Co, Ci, L in SHAPES:
    x = torch.relu(torch.randn(B, Ci, L, device=dev) + 0.3)
    dy = torch.randn(B, Co, L, device=dev)
    w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5
    scale, shift = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.2
    sl = slice(0, min(L, 4096))
    ref = {"fwd": torch.einsum("oc,bcl->bol", w.double(), x[0:1, :, sl].double()),
           "fwd_bn": torch.einsum("oc,bcl->bol", w.double(), torch.relu(scale.double()[None, :, None] * x[0:1, :, sl].double() + shift.double()[None, :, None])),
           "dgrad": torch.einsum("oc,bol->bcl", w.double(), dy[0:1, :, sl].double())}
    for op in ("fwd", "fwd_bn", "dgrad"):
        if op == "fwd_bn" and not (c.pending_ok.__globals__["_native_ok"](B, Co, Ci, L) and Ci <= 512):
            continue
        res = {}
        for flag in (False, True):
            c.SPLIT_PRODUCTS = flag
            if op == "fwd":
                fn = lambda: c.native_fwd(x, w)
            elif op == "fwd_bn":
                fn = lambda: c.native_fwd_bnrelu(x, w, scale, shift, 1)
            else:
                fn = lambda: c.native_dgrad(w, dy)
            out = fn()
            err = float((out[0:1, :, sl].double() - ref[op]).abs().max())
            res[flag] = (timed(fn), err)
        c.SPLIT_PRODUCTS = False
        tot[0] += res[False][0]; tot[1] += res[True][0]
        print("%-28s %-8s %9.3f %9.3f %7.2f   %10.2e %10.2e" % ("%d %d %d %d" % (B, Co, Ci, L), op, res[False][0], res[True][0],
                                                               res[False][0] / res[True][0], res[False][1], res[True][1]))
print("sum: fp32 %.2f ms, split %.2f ms" % tuple(tot))
