"""Probe: rocBLAS GEMM formulations of the 1x1 convolution (forward / input gradient / weight gradient) vs F.conv2d."""
import time, torch
import torch.nn.functional as F
dev = "cuda:0"
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (B, Ci, Co, L) in [(4, 6, 128, 327680), (4, 128, 256, 327680), (4, 259, 256, 65536), (4, 515, 512, 16384), (4, 1536, 1024, 1024), (4, 1280, 512, 5120), (4, 515, 256, 25600), (4, 256, 512, 25600), (4, 256, 128, 25600)]:
    x = torch.randn(B, Ci, L, device=dev); w = torch.randn(Co, Ci, device=dev); dy = torch.randn(B, Co, L, device=dev)
    fl = 2.0 * B * Ci * Co * L / 1e9
    r = {}
    r["conv fwd"] = t(lambda: F.conv1d(x, w[:, :, None]))
    r["matmul fwd"] = t(lambda: torch.matmul(w, x))
    r["bmm fwd"] = t(lambda: torch.bmm(w.expand(B, Co, Ci), x))
    r["matmul dgrad"] = t(lambda: torch.matmul(w.t(), dy))
    r["conv dgrad"] = t(lambda: torch.ops.aten.convolution_backward(dy, x, w[:, :, None], None, (1,), (0,), (1,), False, (0,), 1, (True, False, False)))
    r["conv wgrad"] = t(lambda: torch.ops.aten.convolution_backward(dy, x, w[:, :, None], None, (1,), (0,), (1,), False, (0,), 1, (False, True, False)))
    for S in (1, 8, 32, 128) if L >= 16384 else (1, 4):
        Ls = L // S
        def wg():
            part = torch.empty((B, S, Co, Ci), device=dev)
            for b in range(B):
                torch.bmm(dy[b].view(Co, S, Ls).transpose(0, 1), x[b].view(Ci, S, Ls).permute(1, 2, 0), out=part[b])
            return part.sum((0, 1))
        r["bmm wgrad S=%d" % S] = t(wg)
    def wg2():   # one GEMM with K = B*L on channel-major copies
        return torch.matmul(dy.transpose(0, 1).reshape(Co, B * L), x.transpose(0, 1).reshape(Ci, B * L).t())
    r["copy+mm wgrad"] = t(wg2)
    print("B%d Ci%d Co%d L%d (%.0f GF):" % (B, Ci, Co, L, fl), "  ".join("%s %.2f" % kv for kv in r.items()))
