"""Step time of the default pipeline as a function of WHICH stream handle is read first (ORDER=fps0,fps1,geo,mlp,reg ... ; the
letters of argv[1]: M = read the feature stream's handle before the first run, A = all in pipeline order, L = after the warm-up,
t / c / e = bench.py's timers installed / critical streams set early / enabled): the HIP runtime binds streams to hardware queues
in order of first use, and a feature stream bound first is 8 % slower.  NOTE: ForwardPipeline now binds its streams itself at
construction, so M and ORDER only show the effect with that loop removed."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
variant = sys.argv[1]
sys.argv = ["bench.py", "--cpu-scenes", "0", "--latency-runs", "0"]
args = bench.parse()
from regnet_for_3d_grasping_amd import sharding
rank, local_rank, world = sharding.env_world()
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
from regnet_for_3d_grasping_amd import pipeline, synthetic
timer = bench.OpTimer(args.time_every)
if "t" in variant:
    bench.install_timers(timer)
score_net, region_net = pipeline.build_models(dev)
seeds = sharding.scene_seeds(rank, world, args.batch)
pc = torch.from_numpy(np.stack([synthetic.make_scene(s_, args.points) for s_ in seeds], 0)).to(dev)
synthetic.calibrate_score_head(score_net, pc)
np.random.seed(1234 + rank)
pipe = pipeline.ForwardPipeline(score_net, region_net, with_region=not args.score_only, fps_streams=args.fps_streams,
                                mlp_streams=args.mlp_streams, fps_group=args.fps_group)
if "c" in variant:
    timer.critical_streams = {m.cuda_stream for m in pipe.s_mlps}
if os.environ.get("ORDER"):
    names = {"fps0": pipe.s_fps[0], "fps1": pipe.s_fps[1], "geo": pipe.s_geo, "mlp": pipe.s_mlp, "reg": pipe.s_reg}
    for nm in os.environ["ORDER"].split(","):
        _ = names[nm].cuda_stream
if "A" in variant:
    _ = [st.cuda_stream for st in list(pipe.s_fps) + [pipe.s_geo] + list(pipe.s_mlps) + [pipe.s_reg]]
if "R" in variant:
    _ = pipe.s_reg.cuda_stream
if "F" in variant:
    _ = pipe.s_fps[0].cuda_stream
if "M" in variant:
    _ = pipe.s_mlp.cuda_stream
def run_steps(n):
    last = None
    for last in pipe.run((pc for _ in range(n)), max_pending_regions=args.lookahead):
        pass
    return last
run_steps(args.warmup)
torch.cuda.synchronize()
if "L" in variant:
    _ = pipe.s_mlp.cuda_stream
if "e" in variant:
    timer.enabled = True
t0 = time.perf_counter()
out = run_steps(args.steps)
torch.cuda.synchronize()
print(variant, "%.3f ms/step" % ((time.perf_counter() - t0) / args.steps * 1e3))
