"""Feature stage alone (one stream, nothing else on the GPU): per-op HIP-event times at several batch sizes, to separate a
kernel's own time from what the pipeline's other streams add.  usage: python scripts/features_alone.py [batches...]
FPS_STREAMS=n keeps n level-1 sampling launches of the same batch in flight on side streams meanwhile (FPS_PRIORITY=0/-1)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from regnet_for_3d_grasping_amd import pipeline, synthetic

dev = torch.device("cuda:0")
score_net, region_net = pipeline.build_models(dev)
timer = bench.OpTimer(every=1)
bench.install_timers(timer)
for B in [int(a) for a in sys.argv[1:]] or [4, 8, 16]:
    pc = synthetic.make_batch(1000, B, 25600, device=dev)
    with torch.no_grad():
        plan = score_net.plan(pc)
        for _ in range(3):
            score_net(pc, plan=plan)
        torch.cuda.synchronize()
        timer.records.clear(); timer.calls.clear(); timer.enabled = True
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = int(os.environ.get("REPS", 20))
        nside = int(os.environ.get("FPS_STREAMS", 0))
        side = [torch.cuda.Stream(dev, priority=int(os.environ.get("FPS_PRIORITY", -1))) for _ in range(nside)]
        timer.enabled = False
        for st in side:
            with torch.cuda.stream(st):
                sub = pc[:int(os.environ.get("FPS_SCENES", B))]
                for _ in range(int(reps * 9.0 * B / 8 / 9.9) + 3):
                    score_net.sample_level1(sub)
        if os.environ.get("SIDE_BLOCKS"):
            import ctypes
            lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libclock_probe.so"))
            lib.side_load_lds.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p]
            sink = torch.zeros(4, device=dev)
            sst = torch.cuda.Stream(dev, priority=-1)
            lib.side_load_lds(int(os.environ["SIDE_BLOCKS"]), int(os.environ.get("SIDE_THREADS", 1024)), reps * 10.0 * B / 8 + 10, int(os.environ.get("SIDE_MODE", 6)),
                              int(os.environ.get("SIDE_LDS", 150 * 1024)), sink.data_ptr(), sst.cuda_stream)
        timer.enabled = True
        s.record()
        for _ in range(reps):
            score_net(pc, plan=plan)
        e.record(); torch.cuda.synchronize(); timer.enabled = False
    print("B=%d: feature stage %.3f ms per batch" % (B, s.elapsed_time(e) / reps))
    rows = sorted(timer.summary().items(), key=lambda kv: -kv[1][0])
    for (name, meta), (tot, calls) in rows[:int(os.environ.get("ROWS", 12))]:
        fl = [t for t in meta.split() if t.startswith("flop")]
        tf = int(fl[0][4:]) * calls / tot / 1e9 if fl else 0.0
        print("   %-18s %-44s %8.4f ms %7.1f TF" % (name, meta[:44], tot / reps / (calls / reps), tf))
