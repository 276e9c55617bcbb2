"""Stand-alone timing of the training 1x1-convolution kernels (csrc/tgemm.hip: forward, input gradient, weight gradient)
against torch.bmm (rocBLAS) on the layer shapes of a training iteration (batch of 8 scenes x 25 600 points)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regnet_for_3d_grasping_amd import conv1x1_train as c
DEV = "cuda:0"
B = int(os.environ.get("BATCH", 8))
shapes = [(128, 128, 5120 * 64), (128, 256, 5120 * 64), (256, 256, 1024 * 64), (256, 512, 1024 * 64),
          (512, 512, 256 * 64), (512, 1024, 256 * 64), (256, 256, 25600), (256, 512, 25600), (512, 256, 25600),
          (1024, 1024, 1024), (512, 512, 5120)]

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

print("%-26s %10s %10s | %10s %10s | %10s %10s   (TFLOP/s: native / rocBLAS)" % ("Ci->Co x L", "fwd", "fwd-lib", "dgrad", "dgrad-lib", "wgrad", "wgrad-lib"))
for Ci, Co, L in shapes:
    x = torch.randn(B, Ci, L, device=DEV); w = torch.randn(Co, Ci, device=DEV) * 0.05; dy = torch.randn(B, Co, L, device=DEV)
    flop = 2.0 * B * Ci * Co * L
    wexp, wtexp = w.unsqueeze(0).expand(B, -1, -1), w.t().unsqueeze(0).expand(B, -1, -1)
    t = [timeit(lambda: c.native_fwd(x, w)), timeit(lambda: torch.bmm(wexp, x)),
         timeit(lambda: c.native_dgrad(w, dy)), timeit(lambda: torch.bmm(wtexp, dy)),
         timeit(lambda: c.native_wgrad(dy, x)), timeit(lambda: c._Conv1x1.backward.__func__ and _lib_wgrad(dy, x, w)) if False else 0.0]
    # library weight gradient: the split-K bmm of conv1x1_train (REGNET_CONV1X1_NATIVE=0 path)
    def lib_wgrad():
        S = c._chunks(L, ((Co + 127) // 128) * ((Ci + 127) // 128), B)
        Ls = L // S
        part = torch.empty((B, S, Co, Ci), dtype=torch.float32, device=DEV)
        for b in range(B):
            torch.bmm(dy[b].view(Co, S, Ls).transpose(0, 1), x[b].view(Ci, S, Ls).permute(1, 2, 0), out=part[b])
        return part.sum((0, 1))
    t[5] = timeit(lib_wgrad)
    print("%4d->%4d x %-12d %10.1f %10.1f | %10.1f %10.1f | %10.1f %10.1f" % ((Ci, Co, L) + tuple(flop / ms / 1e9 for ms in t)))
    del x, dy
