"""csrc/bn_train.hip passes alone: microseconds and TB/s of HBM traffic per call, idle GPU.   python scripts/bench_bn_passes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regnet_for_3d_grasping_amd import _lib
L_ = _lib.lib
dev = "cuda:0"


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (B, C, L) in ((8, 128, 327680), (8, 256, 327680), (8, 256, 65536), (8, 512, 16384), (8, 256, 25600), (8, 1024, 1024)):
    x = torch.randn(B, C, L, device=dev); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(L_.regnet_bn_workspace_bytes(C), dtype=torch.uint8, device=dev)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); o = torch.empty(6, C, device=dev)
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    t_stats = timeit(lambda: L_.regnet_bn_train_stats_f32(x.data_ptr(), B, C, L, g.data_ptr(), b.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), ws.data_ptr(), st))
    t_fwd = timeit(lambda: L_.regnet_bn_relu_train_fwd_f32(x.data_ptr(), B, C, L, g.data_ptr(), b.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), 1, 0, y.data_ptr(), None, o[0].data_ptr(), o[1].data_ptr(), ws.data_ptr(), st))
    t_bwd = timeit(lambda: L_.regnet_bn_relu_train_bwd_f32(x.data_ptr(), None, dy.data_ptr(), None, B, C, L, g.data_ptr(), b.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), 1, 0, dx.data_ptr(), o[4].data_ptr(), o[5].data_ptr(), ws.data_ptr(), st))
    S = B * C * L * 4 / 1e6   # MB
    print("B%d C%4d L%6d (%6.0f MB) | stats %7.1f us %4.2f TB/s | stats+apply %7.1f us %4.2f TB/s | backward (reduce + apply) %7.1f us %4.2f TB/s" % (
        B, C, L, S, t_stats, S / t_stats, t_fwd, 3 * S / t_fwd, t_bwd, 5 * S / t_bwd))
    del x, y, dy, dx
