"""Host (wall) time of each pipeline stage's ENQUEUE per batch, by batch size: where small batches lose their time.
python scripts/host_time_probe.py [B ...]   -> gpurun_out/host_time_probe.txt style lines on stdout"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regnet_for_3d_grasping_amd import pipeline, synthetic

dev = torch.device("cuda:0")
score_net, region_net = pipeline.build_models(dev)
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 8]:
    for with_region in (False, True):
        pipe = pipeline.ForwardPipeline(score_net, region_net, with_region=with_region)
        acc = {}

        def wrap(name):
            fn = getattr(pipe, name)

            def timed(*a, **k):
                t0 = time.perf_counter()
                r = fn(*a, **k)
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
                return r
            setattr(pipe, name, timed)
        for n in ("_sample_group", "_geometry", "_features", "_region"):
            wrap(n)
        batches = [synthetic.make_batch(100 + i, B, 25600).to(dev) for i in range(4)]
        steps = 60
        for _ in pipe.run(batches[i % 4] for i in range(10)):
            pass
        torch.cuda.synchronize()
        acc.clear()
        t0 = time.perf_counter()
        for _ in pipe.run(batches[i % 4] for i in range(steps)):
            pass
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        print("B=%d region=%d: %.3f ms/step (%.0f scenes/s); host ms per batch: %s" % (
            B, with_region, wall, B * 1e3 / wall, ", ".join("%s %.3f" % (k, v / steps * 1e3) for k, v in acc.items())), flush=True)
