"""Bisect of scripts/train_graph_probe.py: which capture layout of ScoreNet's training forward / backward replays correctly?
   python scripts/train_graph_probe2.py MODE [B] [N]      MODE: fwd | one | two_shared | two_private"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from regnet_for_3d_grasping_amd import synthetic, fused
from regnet_for_3d_grasping_amd.score_network import ScoreNetwork

mode = sys.argv[1]
dev = torch.device("cuda:0")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = int(sys.argv[3]) if len(sys.argv) > 3 else 25600
net = ScoreNetwork(training=True)
net.load_state_dict(synthetic.seeded_state_dict(net, 7))
net = net.to(dev).train()
net.extrat_featurePN2.mlp.dropout_prob = 0.0
pcs = [synthetic.make_batch(300 + 10 * i, B, N).to(dev) for i in range(2)]
targets = [torch.from_numpy(np.random.default_rng(i).uniform(0, 1, (B, N)).astype(np.float32)).to(dev) for i in range(2)]
params = [p for p in net.parameters() if p.requires_grad]
names = [n for n, p in net.named_parameters() if p.requires_grad]


def eager(pc, target, plan, backward=True):
    for p in params:
        p.grad = None
    _, score, loss = net(pc, target, None, plan=plan)
    if backward:
        loss.sum().backward()
    return loss.detach().clone(), [None if p.grad is None else p.grad.clone() for p in params], score.detach().clone()


side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    plans = [net.plan(pc) for pc in pcs]
    state = {k: v.clone() for k, v in net.state_dict().items()}
    ref = []
    for pc, t, pl in zip(pcs, targets, plans):
        net.load_state_dict(state)
        ref.append(eager(pc, t, pl))
    # eager against eager: what the atomics' summation order alone does to this metric
    for i in (0, 1):
        net.load_state_dict(state)
        again = eager(pcs[i], targets[i], plans[i])
        worst, where, num, den = 0.0, None, 0.0, 0.0
        for nm, g, r in zip(names, again[1], ref[i][1]):
            if r is None:
                continue
            d = float((g - r).abs().max() / (r.abs().max() + 1e-12))
            num += float(((g - r).double() ** 2).sum()); den += float((r.double() ** 2).sum())
            if not d <= worst:
                worst, where = d, nm
        print("eager vs eager batch %d: worst grad rel diff %.2e (%s), global rel L2 %.2e" % (i, worst, where, (num / den) ** 0.5))
    net.load_state_dict(state)
    pc_s, tg_s = pcs[0].clone(), targets[0].clone()
    plan_s = net.plan(pc_s)
    plan_s_t = fused.plan_tensors(plan_s)
    for p in params:
        p.grad = None
side.synchronize()
import gc
gc.collect()
gc.disable()
with torch.cuda.stream(side):
    g_f, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    if mode == "fwd":
        with torch.cuda.graph(g_f, stream=side), torch.no_grad():
            _, score_s, loss_s = net(pc_s, tg_s, None, plan=plan_s)
        graphs = [g_f]
    elif mode == "one":
        with torch.cuda.graph(g_f, stream=side):
            _, score_s, loss_s = net(pc_s, tg_s, None, plan=plan_s)
            loss_s.sum().backward()
        graphs = [g_f]
    else:
        with torch.cuda.graph(g_f, stream=side):
            _, score_s, loss_s = net(pc_s, tg_s, None, plan=plan_s)
            total_s = loss_s.sum()
        kw = {"pool": g_f.pool()} if mode == "two_shared" else {}
        with torch.cuda.graph(g_b, stream=side, **kw):
            total_s.backward()
        graphs = [g_f, g_b]
    grads_s = [p.grad for p in params]
    for i in (0, 1, 0, 1):
        net.load_state_dict(state)
        pc_s.copy_(pcs[i]); tg_s.copy_(targets[i])
        for a, b in zip(plan_s_t, fused.plan_tensors(plans[i])):
            a.copy_(b)
        for g in graphs:
            g.replay()
        side.synchronize()
        worst, where, num, den = 0.0, None, 0.0, 1e-300
        if mode != "fwd":
            for nm, g, r in zip(names, grads_s, ref[i][1]):
                if r is None or g is None:
                    assert r is None and g is None, nm
                    continue
                d = float((g - r).abs().max() / (r.abs().max() + 1e-12))
                num += float(((g - r).double() ** 2).sum()); den += float((r.double() ** 2).sum())
                if not d <= worst:
                    worst, where = d, nm
        print("   global rel L2 %.2e" % ((num / den) ** 0.5))
        print("%s batch %d: score max|diff| %.3e  loss %.8f / %.8f  worst grad rel diff %.2e (%s)" % (
            mode, i, float((score_s - ref[i][2]).abs().max()), float(loss_s.detach()), float(ref[i][0]), worst, where))
