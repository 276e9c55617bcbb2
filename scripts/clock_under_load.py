"""Does concurrent work lower the shader clock?  A one-wave dependent-FMA chain (scripts/ablate/clock_probe.hip, ~0.4 ms) is
launched on its own high-priority stream (a) alone, (b) beside level-1 FPS launches, (c) beside the feature stage, (d) beside
both; its links per microsecond follow the shader clock.  usage: python scripts/clock_under_load.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from regnet_for_3d_grasping_amd import pipeline, synthetic

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libclock_probe.so"))
lib.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
score_net, _ = pipeline.build_models(dev)
B = 8
pc = synthetic.make_batch(1000, B, 25600, device=dev)
LINKS = 1 << 18
probe_stream = torch.cuda.Stream(dev, priority=-1)
mlp_stream = torch.cuda.Stream(dev)
fps_streams = [torch.cuda.Stream(dev, priority=-1) for _ in range(2)]
with torch.no_grad():
    plan = score_net.plan(pc)
    for _ in range(3):
        score_net(pc, plan=plan)
torch.cuda.synchronize()


def run(label, n_fps, features):
    outs = []
    with torch.no_grad():
        for st in fps_streams[:n_fps]:
            with torch.cuda.stream(st):
                for _ in range(8):
                    score_net.sample_level1(pc)
        if features:
            with torch.cuda.stream(mlp_stream):
                for _ in range(9):
                    score_net(pc, plan=plan)
        time.sleep(0.004)                       # let the load get going before the first probe
        with torch.cuda.stream(probe_stream):
            for _ in range(40):
                o = torch.zeros(3, dtype=torch.int64, device=dev)
                lib.clock_probe(o.data_ptr(), LINKS, probe_stream.cuda_stream)
                outs.append(o)
                time.sleep(0.001)
    torch.cuda.synchronize()
    v = torch.stack(outs).cpu().double()
    per_us = LINKS / (v[:, 1] / 100.0)          # links per microsecond (100 MHz counter)
    ratio = v[:, 0] / v[:, 1]                   # shader counter ticks per 10 ns
    q = torch.quantile(per_us, torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9], dtype=torch.float64))
    print("%-34s links/us: min %.1f  p10 %.1f  p25 %.1f  median %.1f  p75 %.1f  p90 %.1f  max %.1f" %
          (label, per_us.min(), q[0], q[1], q[2], q[3], q[4], per_us.max()))


run("probe alone", 0, False)
run("beside 1 FPS launch", 1, False)
run("beside 2 FPS launches", 2, False)
run("beside the feature stage", 0, True)
run("beside feature stage + 1 FPS", 1, True)
run("beside feature stage + 2 FPS", 2, True)
run("probe alone (again)", 0, False)
