"""Acceptance statistics of exact multi-pick furthest point sampling (csrc/geometry.hip: fps_sorted_kernel<.., KP>, fps_cluster_kernel):
how many candidates per round pass the ordered acceptance test, by candidate count K and record granularity (points per record).
CPU / numpy only: python scripts/fps_multipick_sim.py"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regnet_for_3d_grasping_amd import synthetic
N, M = 25600, 5120
pc = synthetic.make_scene(1000, N)[:, :3].astype(np.float32)
# Morton sort as the kernel does: 16^3 grid
lo, hi = pc.min(0), pc.max(0)
q = np.minimum(15, ((pc - lo) * (16.0 / (hi - lo))).astype(np.int64))
def spread(v):
    return (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6)
code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
order = np.argsort(code, kind='stable')
P = pc[order]
def sim(K, unit):   # unit: points per record group (400 = DPP row of 16 threads x 25, 25 = thread)
    G = N // unit
    dist = np.full(N, np.inf, np.float32)
    cur = [int(np.where(order == 0)[0][0])]
    picks, rounds, hist = 1, 0, {}
    while picks < M:
        for c in cur:
            d = ((P - P[c]) ** 2).sum(1).astype(np.float32)
            dist = np.minimum(dist, d)
        rounds += 1
        g = dist.reshape(G, unit)
        a = g.argmax(1)
        v1 = g[np.arange(G), a]
        g2 = g.copy(); g2[np.arange(G), a] = -1
        v2 = g2.max(1)
        top = np.argsort(-v1)[:K + 1]
        bound = max(v1[top[K]], v2[top[:K]].max())
        acc = [top[0] * unit + a[top[0]]]
        for j in range(1, K):
            cj = top[j] * unit + a[top[j]]
            vj = v1[top[j]]
            if not (vj > bound and vj < v1[top[j - 1]]):
                break
            if any(((P[cj] - P[x]) ** 2).sum() < vj for x in acc):
                break
            acc.append(cj)
        acc = acc[:M - picks]
        hist[len(acc)] = hist.get(len(acc), 0) + 1
        picks += len(acc)
        cur = acc
    return rounds, hist
for K in (2, 4, 6, 8):
    for unit in (400, 100, 25):
        r, h = sim(K, unit)
        print("K=%d unit=%d: %d rounds for %d picks (%.2f picks/round) %s" % (K, unit, r, M, M / r, dict(sorted(h.items()))))
