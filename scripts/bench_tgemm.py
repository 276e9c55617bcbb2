"""csrc/tgemm.hip forward / input-gradient contractions alone: microseconds per call and TFLOP/s on an idle GPU for the
training iteration's layer shapes (8 x 25 600 points).   python scripts/bench_tgemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regnet_for_3d_grasping_amd import conv1x1_train
dev = "cuda:0"


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


tot_f = tot_d = 0.0
for (B, Ci, Co, L) in ((8, 128, 128, 327680), (8, 128, 256, 327680), (8, 256, 256, 65536), (8, 256, 512, 65536),
                       (8, 512, 512, 16384), (8, 512, 1024, 16384), (8, 256, 256, 25600), (8, 512, 256, 25600),
                       (8, 1024, 1024, 1024), (8, 512, 512, 5120), (8, 256, 128, 25600), (3, 64, 48, 2064)):
    x = torch.randn(B, Ci, L, device=dev); w = torch.randn(Co, Ci, device=dev) * 0.05
    dy = torch.randn(B, Co, L, device=dev)
    tf = timeit(lambda: conv1x1_train.native_fwd(x, w))
    td = timeit(lambda: conv1x1_train.native_dgrad(w, dy))
    y64 = torch.einsum("oi,bil->bol", w.double(), x[:1, :, :4096].double())
    err = float((conv1x1_train.native_fwd(x, w)[:1, :, :4096].double() - y64).abs().max())
    fl = 2.0 * B * Ci * Co * L
    tot_f += tf; tot_d += td
    print("B%d Ci%4d Co%4d L%6d | fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | err %.1e" % (B, Ci, Co, L, tf, fl / tf / 1e6, td, fl / td / 1e6, err))
    del x, dy
print("sum fwd %.1f us, dgrad %.1f us" % (tot_f, tot_d))
