"""Can ScoreNet's TRAINING forward + backward (train-mode BatchNorm, dropout, the scatter-add backward of the gathers) be
captured as hipGraphs, do replays give the gradients of an eager pass, and what does an iteration cost the host either way?
   python scripts/train_graph_probe.py [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from regnet_for_3d_grasping_amd import synthetic
from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
from regnet_for_3d_grasping_amd import fused

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25600
net = ScoreNetwork(training=True)
net.load_state_dict(synthetic.seeded_state_dict(net, 7))
net = net.to(dev).train()
pcs = [synthetic.make_batch(300 + i, B, N).to(dev) for i in range(2)]
targets = [torch.from_numpy(np.random.default_rng(i).uniform(0, 1, (B, N)).astype(np.float32)).to(dev) for i in range(2)]
params = [p for p in net.parameters() if p.requires_grad]
seg = net.extrat_featurePN2
seg.mlp.dropout_prob = 0.0          # dropout off for the equality check (its mask comes from the generator's state)


def eager(pc, target, plan):
    for p in params:
        p.grad = None
    _, score, loss = net(pc, target, None, plan=plan)
    loss.sum().backward()
    return loss.detach().clone(), [None if p.grad is None else p.grad.clone() for p in params], score.detach().clone()


plans = [net.plan(pc) for pc in pcs]
bn_state = {k: v.clone() for k, v in net.state_dict().items()}
ref = [eager(pc, t, pl) for pc, t, pl in zip(pcs, targets, plans)]
torch.cuda.synchronize()
net.load_state_dict(bn_state)

# static inputs
pc_s, tg_s = pcs[0].clone(), targets[0].clone()
plan_s = net.plan(pc_s)
plan_s_t = fused.plan_tensors(plan_s)
print("plan tensors:", [(tuple(t.shape), str(t.dtype)) for t in plan_s_t])
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        eager(pc_s, tg_s, plan_s)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
net.load_state_dict(bn_state)
for p in params:
    p.grad = None
import gc
gc.collect()
gc.disable()
g_f, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
t0 = time.perf_counter()
grabbed = {}
hook = seg.register_forward_hook(lambda _m, _i, out: grabbed.__setitem__("feat", out[0]))
with torch.cuda.graph(g_f, stream=side):
    _, score_s, loss_s = net(pc_s, tg_s, None, plan=plan_s)
    total_s = loss_s.sum()
hook.remove()
with torch.cuda.graph(g_b, stream=side, pool=g_f.pool()):
    total_s.backward()
print("captured forward + backward in %.1f ms" % ((time.perf_counter() - t0) * 1e3))
gc.enable()
grads_s = [p.grad for p in params]

for i in (0, 1, 0, 1):
    net.load_state_dict(bn_state)
    pc_s.copy_(pcs[i]); tg_s.copy_(targets[i])
    for a, b in zip(plan_s_t, fused.plan_tensors(plans[i])):
        a.copy_(b)
    g_f.replay(); g_b.replay()
    torch.cuda.synchronize()
    worst = 0.0
    for g, r in zip(grads_s, ref[i][1]):
        if r is None or g is None:
            assert r is None and g is None
            continue
        worst = max(worst, float((g - r).abs().max() / (r.abs().max() + 1e-12)))
    print("   score max|diff| %.3e, score mean replay %.5f eager %.5f" % (float((score_s - ref[i][2]).abs().max()), float(score_s.mean()), float(ref[i][2].mean())))
    print("batch %d: loss replay %.8f eager %.8f, worst relative gradient difference %.2e (atomics reorder sums)"
          % (i, float(loss_s), float(ref[i][0]), worst))

for name in ("eager", "graphs"):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        if name == "eager":
            eager_loss = eager(pc_s, tg_s, plan_s)
        else:
            g_f.replay(); g_b.replay()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    print("%s: host %.3f ms, device %.3f ms per forward+backward (B=%d, N=%d)" % (name, host, e0.elapsed_time(e1) / n, B, N))
print("memory: allocated %.1f GB, reserved %.1f GB" % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
