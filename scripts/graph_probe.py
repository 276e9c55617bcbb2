"""Can the geometry + feature stages of one batch be captured as hipGraphs, do replays give the same bits, and what does a
replay cost the host?   python scripts/graph_probe.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regnet_for_3d_grasping_amd import fused, pipeline, synthetic

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
score_net, region_net = pipeline.build_models(dev)
pcs = [synthetic.make_batch(300 + i, B, 25600).to(dev) for i in range(3)]
with torch.no_grad():
    fused.prepack(score_net, region_net)
    ctrs = [score_net.sample_levels(pc) for pc in pcs]
    ref = []
    for pc, ctr in zip(pcs, ctrs):
        plan = score_net.plan(pc, ctr)
        f, s, _ = score_net(pc, plan=plan)
        ref.append((f.clone(), s.clone()))
    torch.cuda.synchronize()
    pc_s = pcs[0].clone()
    ctr_s = [c.clone() for c in ctrs[0]]
    side = torch.cuda.Stream(dev)
    g_geo, g_feat = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(g_geo, stream=side):
        plan_s = score_net.plan(pc_s, ctr_s)
    with torch.cuda.graph(g_feat, stream=side):
        f_s, s_s, _ = score_net(pc_s, plan=plan_s)
    print("captured both graphs in %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    for i, (pc, ctr) in enumerate(zip(pcs, ctrs)):
        pc_s.copy_(pc)
        for a, b in zip(ctr_s, ctr):
            a.copy_(b)
        g_geo.replay()
        g_feat.replay()
        torch.cuda.synchronize()
        print("batch %d: feature equal %s, score equal %s" % (i, torch.equal(f_s, ref[i][0]), torch.equal(s_s, ref[i][1])))
    # host cost and device time per replay
    for name, fn in (("eager", None), ("graphs", True)):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            if fn is None:
                plan = score_net.plan(pc_s, ctr_s)
                score_net(pc_s, plan=plan)
            else:
                g_geo.replay()
                g_feat.replay()
        e1.record()
        host = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        print("%s: host %.3f ms per batch, device %.3f ms per batch (one stream, B=%d)" % (name, host, e0.elapsed_time(e1) / n, B))
