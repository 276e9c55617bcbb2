"""Gaps between consecutive kernels of the feature stage when it runs ALONE on one stream (sequential forwards with a fixed
geometry plan) -- which kernel boundaries cost what without any other stream.
    rocprofv3 --kernel-trace --stats -d gpurun_out/gaps -o g -- python scripts/chain_boundary_gaps.py ; python scripts/chain_boundary_gaps.py --report gpurun_out/gaps"""
import collections, glob, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--report" in sys.argv:
    db = glob.glob(sys.argv[sys.argv.index("--report") + 1] + "/**/*results.db", recursive=True)[0]
    rows = list(sqlite3.connect(db).cursor().execute("select start,end,name,queue_id from kernels order by start"))
    q = collections.Counter(r[3] for r in rows if r[2].startswith("sa_chain_kernel")).most_common(1)[0][0]
    fq = [r for r in rows if r[3] == q]
    sh = lambda n: n.split("(")[0].replace("void ", "")[:40]
    gap = collections.defaultdict(list)
    first = [i for i, r in enumerate(fq) if r[2].startswith("sa_chain_kernel")][4]
    for i in range(first, len(fq) - 1):
        gap[(sh(fq[i][2]), sh(fq[i + 1][2]))].append((fq[i + 1][0] - fq[i][1]) / 1e3)
    key = ("fp_head_chain_kernel<true>", "sa_chain_kernel")
    if key in gap:
        v = sorted(gap[key])
        print("fp_head -> sa_chain: median %.1f us (min %.1f max %.1f, n %d)" % (v[len(v) // 2], v[0], v[-1], len(v)))
    for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]) / len(kv[1]))[:4]:
        print("%7.1f us avg (min %6.1f max %6.1f, n %d)  %s -> %s" % (sum(v) / len(v), min(v), max(v), len(v), k[0], k[1]))
    sys.exit(0)
import torch
from regnet_for_3d_grasping_amd import fused, pipeline, synthetic
dev = torch.device("cuda:0")
score_net, _ = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 8, 25600, device=dev)
mode = os.environ.get("MODE", "plain")
with torch.no_grad():
    plan = score_net.plan(pc)
    side = torch.cuda.Stream(dev)
    other = torch.cuda.Stream(dev)
    cur = torch.cuda.current_stream(dev)
    if mode == "fps":           # level-1 sampling launches (one CU per scene for ~4 ms each) on a side stream meanwhile
        with torch.cuda.stream(side):
            for _ in range(40):
                score_net.sample_level1(pc)
    for _ in range(20):
        if mode == "tail":      # as the pipeline: the chain's partial last round on a side stream
            fused.TAIL_SINK = []
        if mode == "event":     # a stage boundary as the pipeline's: an event another stream waits for, and a wait for that stream's event
            e1 = torch.cuda.Event(); e1.record(cur); other.wait_event(e1)
            with torch.cuda.stream(other):
                z = torch.zeros(16, device=dev)
                e2 = torch.cuda.Event(); e2.record(other)
            cur.wait_event(e2)
        score_net(pc, plan=plan)
        fused.TAIL_SINK = None
torch.cuda.synchronize()
