"""A resident VALU-dense workgroup slows the whole feature stage (scripts/side_load_probe.py).  Which resource?  Times a
register-only MFMA kernel (all CUs) and an HBM copy, alone and beside side kernels."""
import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libclock_probe.so"))
lib.side_load.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.mfma_burn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.hbm_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(4, device=dev)
src = torch.ones(1 << 28, device=dev)          # 1 GiB
dst = torch.empty_like(src)
side = torch.cuda.Stream(dev, priority=-1)
main = torch.cuda.current_stream(dev)
ITERS = 20000                                    # 160k MFMAs per wave, 4 waves per SIMD-set: ~2.5 ms


def work(kind):
    if kind == "mfma":
        lib.mfma_burn(256 * 2, ITERS, sink.data_ptr(), main.cuda_stream)
    else:
        lib.hbm_stream(src.data_ptr(), dst.data_ptr(), src.numel() // 4, 256 * 8, main.cuda_stream)


def timed(label, kind, side_args):
    if side_args is not None:
        lib.side_load(side_args[0], side_args[1], 90.0, side_args[2], sink.data_ptr(), side.cuda_stream)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    work(kind)
    s.record()
    for _ in range(8):
        work(kind)
    e.record()
    torch.cuda.synchronize()
    print("%-8s %-44s %.3f ms per launch" % (kind, label, s.elapsed_time(e) / 8))


for kind in ("mfma", "hbm"):
    timed("alone", kind, None)
    timed("beside 1 x 1024 threads of FMAs", kind, (1, 1024, 1))
    timed("beside 8 x 1024 threads of FMAs", kind, (8, 1024, 1))
    timed("beside 64 x 1024 threads of FMAs", kind, (64, 1024, 1))
    timed("beside 8 x 1024 threads sleeping", kind, (8, 1024, 0))
    timed("beside 8 x 1024 threads of s_nop", kind, (8, 1024, 5))
    timed("alone (again)", kind, None)
