"""What about a resident side kernel slows the feature stage?  The feature stage (one stream) is timed alone and beside
synthetic side kernels (scripts/ablate/clock_probe.hip:side_load_kernel) that only sleep, only run VALU work, or do an
LDS atomic + barrier per iteration, on 8 workgroups like one level-1 FPS launch -- and beside the FPS launch itself."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from regnet_for_3d_grasping_amd import pipeline, synthetic

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libclock_probe.so"))
lib.side_load.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
score_net, _ = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 8, 25600, device=dev)
sink = torch.zeros(4, device=dev)
side = torch.cuda.Stream(dev, priority=-1)
with torch.no_grad():
    plan = score_net.plan(pc)
    for _ in range(3):
        score_net(pc, plan=plan)
torch.cuda.synchronize()
REPS = 12


def timed(label, launch_side):
    with torch.no_grad():
        if launch_side is not None:
            with torch.cuda.stream(side):
                launch_side()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(REPS):
            score_net(pc, plan=plan)
        e.record()
    torch.cuda.synchronize()
    print("%-46s feature stage %.3f ms per batch" % (label, s.elapsed_time(e) / REPS))


def synth(blocks, threads, mode):
    return lambda: lib.side_load(blocks, threads, 110.0, mode, sink.data_ptr(), side.cuda_stream)


def fps():
    for _ in range(13):
        score_net.sample_level1(pc)


timed("alone", None)
for nb in (1, 2, 4, 8, 16, 32, 64):
    timed("beside %d x 1024 threads of FMAs" % nb, synth(nb, 1024, 1))
timed("beside 8 x 512 threads of FMAs", synth(8, 512, 1))
timed("beside 8 x 256 threads of FMAs", synth(8, 256, 1))
timed("beside 8 x 1024 threads, FMAs ~1/4 duty", synth(8, 1024, 3))
timed("beside 8 x 1024 threads of integer mads", synth(8, 1024, 4))
timed("beside 8 x 1024 threads of s_nop", synth(8, 1024, 5))
timed("beside 8 x 1024 threads sleeping", synth(8, 1024, 0))
timed("beside level-1 FPS (8 x 1024 threads)", fps)
timed("alone (again)", None)
