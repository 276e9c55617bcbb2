"""For each start of a batch's feature stage (sa_chain_kernel) in a kernel trace: the idle time before it on its queue
and when the last geometry kernel (grid 3-NN of level 1) / the last level-1 FPS before it ended -- is the feature stage
waiting for its geometry, or for the host?    python boundary_probe.py <trace dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
chains = [r for r in rows if "sa_chain_kernel" in r[2]]
q = chains[0][3]
mlpq = [r for r in rows if r[3] == q]
geo_last = [r for r in rows if "three_nn_grid_kernel" in r[2]]
fps = [r for r in rows if "fps_sorted_kernel" in r[2] or "fps_multi_kernel" in r[2]]
print("   gap_us  geo_end_before_us  fps_end_before_us   (negative: ended AFTER the feature stage could have started)")
for c in chains[len(chains) // 2: len(chains) // 2 + 14]:
    prev = [r for r in mlpq if r[1] <= c[0]]
    gap = c[0] - prev[-1][1] if prev else 0
    g = [r for r in geo_last if r[1] <= c[0]]
    f = [r for r in fps if r[1] <= c[0]]
    print("%9.1f %18.1f %18.1f" % (gap / 1e3, (c[0] - g[-1][1]) / 1e3 if g else -1, (c[0] - f[-1][1]) / 1e3 if f else -1))
# what ran on the other queues inside those idle windows
import collections
inside = collections.Counter(); dur = collections.Counter(); n = 0
for c in chains[len(chains) // 4: 3 * len(chains) // 4]:
    prev = [r for r in mlpq if r[1] <= c[0]]
    if not prev: continue
    a, b = prev[-1][1], c[0]
    n += 1
    for s, e, name, qq in rows:
        if qq != q and e > a and s < b:
            inside[(name.split("(")[0][:50], qq)] += 1
            dur[(name.split("(")[0][:50], qq)] += min(e, b) - max(s, a)
print("kernels of other queues overlapping the %d idle windows (count, overlapped us per window):" % n)
for k, v in sorted(dur.items(), key=lambda kv: -kv[1])[:14]:
    print("   %5d  %8.1f  %s" % (inside[k], v / 1e3 / n, k))
