"""The gap between a batch's last feature kernel (fp_head_chain) and the next batch's first (sa_chain) INSIDE ForwardPipeline,
from a rocprofv3 kernel trace; PIPE_TAIL=0 keeps the chain's last round in the main launch, PIPE_REGION=0 drops the region stage,
PIPE_TIMERS=1 adds bench.py's event brackets.
    rocprofv3 --kernel-trace --stats -d DIR -o g -- python scripts/pipeline_boundary_gaps.py ; python scripts/pipeline_boundary_gaps.py --report DIR"""
import collections, glob, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--report" in sys.argv:
    db = glob.glob(sys.argv[sys.argv.index("--report") + 1] + "/**/*results.db", recursive=True)[0]
    rows = list(sqlite3.connect(db).cursor().execute("select start,end,name,queue_id from kernels order by start"))
    q = collections.Counter(r[3] for r in rows if r[2].startswith("sa_chain_kernel")).most_common(1)[0][0]
    fq = [r for r in rows if r[3] == q]
    sa = [i for i, r in enumerate(fq) if r[2].startswith("sa_chain_kernel")]
    gaps = sorted((fq[i][0] - fq[i - 1][1]) / 1e3 for i in sa[8:])
    span = (fq[sa[-1]][0] - fq[sa[8]][0]) / 1e6 / (len(sa) - 9)
    print("steps %d: %.3f ms per step; gap before sa_chain: median %.1f us, p10 %.1f, p90 %.1f, mean %.1f" % (
        len(gaps), span, gaps[len(gaps) // 2], gaps[len(gaps) // 10], gaps[len(gaps) * 9 // 10], sum(gaps) / len(gaps)))
    sys.exit(0)
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic
dev = torch.device("cuda:0")
score_net, region_net = pipeline.build_models(dev)
pcs = [synthetic.make_batch(1000 + 8 * k, 8, 25600, device=dev) for k in range(4)]
synthetic.calibrate_score_head(score_net, pcs[0])
np.random.seed(0)
if os.environ.get("PIPE_CALIB") == "1":        # as bench.py: the region head calibrated so that the refine network runs
    synthetic.calibrate_region_head(region_net, lambda: pipeline.forward_scenes(score_net, region_net, pcs[0]))
    np.random.seed(0)
if os.environ.get("PIPE_DISTINCT"):
    pcs = [synthetic.make_batch(1000 + 8 * k, 8, 25600, device=dev) for k in range(int(os.environ["PIPE_DISTINCT"]))]
if os.environ.get("PIPE_TIMERS") == "1":
    import bench
    timer = bench.OpTimer(8); bench.install_timers(timer); timer.enabled = True
pipe = pipeline.ForwardPipeline(score_net, region_net, with_region=os.environ.get("PIPE_REGION", "1") != "0")
if os.environ.get("PIPE_TAIL") == "0":
    pipe.split_chain_tail = False
if os.environ.get("PIPE_GEO_SAME") == "1":      # the geometry on the feature stream itself (no concurrency between the two)
    pipe.s_geo = pipe.s_mlps[0]
if os.environ.get("PIPE_NO_EVENT") == "1":      # no completion event behind a feature stage (the region stage then reads too early: timing only)
    import regnet_for_3d_grasping_amd.pipeline as P
    class _NoEvent:
        def record(self, *a): pass
        def query(self): return False
        def synchronize(self): torch.cuda.synchronize()
    orig_features = pipe._features
    def feats(item):
        real = torch.cuda.Event
        out = orig_features(item)
        return out
    # (kept simple: see PIPE_REGION=0 for the variant without consumers)
for _ in pipe.run(pcs[k % len(pcs)] for k in range(5)):
    pass
torch.cuda.synchronize()
for _ in pipe.run(pcs[k % len(pcs)] for k in range(40)):
    pass
torch.cuda.synchronize()
