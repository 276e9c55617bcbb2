"""Does the feature stage get slower once ANOTHER stream (HW queue) has been used, even if that stream is idle again?
Times the feature stage alone (10 forwards per measurement) before and after touching other streams."""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from regnet_for_3d_grasping_amd import pipeline, synthetic

side = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libclock_probe.so"))
side.side_load_lds.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
score_net, _ = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 8, 25600, device=dev)
sink = torch.zeros(4, device=dev)
with torch.no_grad():
    plan = score_net.plan(pc)
    for _ in range(3):
        score_net(pc, plan=plan)
torch.cuda.synchronize()


def measure(label, reps=10):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        s.record()
        for _ in range(reps):
            score_net(pc, plan=plan)
        e.record()
    torch.cuda.synchronize()
    print("%-64s %.3f ms per batch" % (label, s.elapsed_time(e) / reps), flush=True)


measure("fresh process, only the default stream so far")
measure("again")
st0 = torch.cuda.Stream(dev, priority=0)
with torch.cuda.stream(st0):
    sink.add_(1.0)
torch.cuda.synchronize()
measure("after one tiny kernel on a second (normal-priority) stream")
measure("again")
st1 = torch.cuda.Stream(dev, priority=-1)
with torch.cuda.stream(st1):
    sink.add_(1.0)
torch.cuda.synchronize()
measure("after one tiny kernel on a high-priority stream")
side.side_load_lds(1, 1024, 5.0, 6, 150 * 1024, sink.data_ptr(), st1.cuda_stream)
torch.cuda.synchronize()
measure("after a 5 ms sleeping workgroup on the high-priority stream (finished)")
with torch.cuda.stream(st1), torch.no_grad():
    score_net.sample_level1(pc)
torch.cuda.synchronize()
measure("after one level-1 FPS launch on the high-priority stream (finished)")
time.sleep(1.0)
measure("after one idle second")
del st0, st1
torch.cuda.synchronize()
measure("after dropping the stream objects")
