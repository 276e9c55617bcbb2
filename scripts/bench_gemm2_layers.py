"""The step's plain shared-MLP layers (regnet_mlp_layer_f32 -> gemm2_kernel), one at a time on an otherwise idle GPU or
beside a resident side kernel that takes SIDE_BLOCKS CUs (scripts/ablate/clock_probe.hip): microseconds per launch.
usage: [SIDE_BLOCKS=8] [REGNET_HIP_LIB=variant.so] python scripts/bench_gemm2_layers.py"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from regnet_for_3d_grasping_amd import _lib

L = _lib.lib
dev = torch.device("cuda:0")
SHAPES = [(131072, 512, 1024, 64), (131072, 512, 512, 0), (40960, 512, 512, 0), (40960, 512, 256, 0), (40960, 256, 512, 0), (40960, 259, 256, 0),
          (8192, 1024, 1024, 0), (8192, 512, 1024, 0), (8192, 1024, 512, 0), (8192, 515, 512, 0), (2048, 1024, 1024, 0)]
side_blocks = int(os.environ.get("SIDE_BLOCKS", 0))
if side_blocks:
    side = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libclock_probe.so"))
    side.side_load_lds.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    sink = torch.zeros(4, device=dev)
    sst = torch.cuda.Stream(dev, priority=-1)
st = torch.cuda.current_stream(dev).cuda_stream
if os.environ.get("SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["SHAPES"].split(",")]
torch.manual_seed(0)
for P, K, N, pool in SHAPES:
    Kpad = (K + 15) // 16 * 16
    Ka = (K + 3) // 4 * 4
    A = torch.randn(P, Ka, device=dev)
    W = torch.zeros((N + 127) // 128 * 128, Kpad, device=dev)
    W[:N, :K] = torch.randn(N, K, device=dev) / K ** 0.5
    if os.environ.get("ZERO_OPERANDS") == "1":      # power probe: the same instruction stream on all-zero data (no toggling in the
        A.zero_(); W.zero_()                        # multipliers): a launch that gets faster was limited by the clock it was given
    scale, shift = torch.ones(W.shape[0], device=dev), torch.zeros(W.shape[0], device=dev)
    C = torch.empty(P // pool if pool else P, N, device=dev)
    stream_mode = os.environ.get("STREAM") == "1" and not pool and L.regnet_mlp_layer_stream_supported(P, A.stride(0), Kpad, N)
    tick = torch.zeros(8 * (int(os.environ.get("REPS", 40)) + 8), dtype=torch.int32, device=dev)
    calls = [0]
    def launch():
        if stream_mode:      # the persistent launch (gemm2_stream_kernel): 8 fresh zeroed ticket words per call
            t = tick[8 * calls[0]:8 * calls[0] + 8]
            calls[0] += 1
            rc = L.regnet_mlp_layer_stream_f32(A.data_ptr(), A.stride(0), Ka, W.data_ptr(), Kpad, scale.data_ptr(), shift.data_ptr(),
                                               C.data_ptr(), C.stride(0), P, N, 1, t.data_ptr(), st)
            assert rc == 0, rc
            return
        rc = L.regnet_mlp_layer_f32(A.data_ptr(), A.stride(0), Ka, W.data_ptr(), Kpad, scale.data_ptr(), shift.data_ptr(),
                                    C.data_ptr(), C.stride(0), P, N, 1, pool, st)
        assert rc == 0, rc
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    reps = int(os.environ.get("REPS", 40))
    reps = min(reps, tick.numel() // 8 - 8)
    if side_blocks:
        side.side_load_lds(side_blocks, 1024, float(os.environ.get("SIDE_MS", 12.0)), int(os.environ.get("SIDE_MODE", 6)), 150 * 1024, sink.data_ptr(), sst.cuda_stream)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        launch()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    import hashlib
    digest = hashlib.sha256(C.cpu().numpy().tobytes()).hexdigest()[:12]     # bit-identity across tile variants (same seed)
    print("P%-7d K%-5d N%-5d %s  %8.1f us  %6.1f TFLOP/s  out %s%s" % (P, K, N, "pool" if pool else "    ", us, 2.0 * P * K * N / us / 1e6, digest, "  [stream]" if stream_mode else ""))
