"""Per-queue breakdown of a rocprofv3 kernel trace of bench.py: for the queue that runs the MFMA kernels, how much of
a step is MFMA kernels, other kernels, and idle gaps between consecutive kernels of that queue.

    python scripts/stream_breakdown.py gpurun_out/kt
"""
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
is_mfma = lambda n: "mlp_gemm" in n or "sa_chain" in n
g = [r for r in rows if is_mfma(r[2])]
a, b = g[int(len(g) * 0.4)][0], g[int(len(g) * 0.9)][1]
sel = [r for r in rows if r[0] >= a and r[1] <= b]
steps = sum(1 for r in sel if "sa_chain_kernel" in r[2]) or 1
byq = collections.Counter(r[3] for r in sel if is_mfma(r[2]))
q = byq.most_common(1)[0][0]
qs = [r for r in sel if r[3] == q]
mf = sum(e - s for s, e, n, _ in qs if is_mfma(n))
other = collections.Counter()
for s, e, n, _ in qs:
    if not is_mfma(n):
        other[n.split("(")[0][:60]] += e - s
gaps = sum(max(0, qs[i + 1][0] - qs[i][1]) for i in range(len(qs) - 1))
span = qs[-1][1] - qs[0][0]
print("queue %s: %d steps, %.3f ms/step span; MFMA kernels %.3f, other kernels %.3f, gaps %.3f ms/step; %d kernels/step" %
      (q, steps, span / 1e6 / steps, mf / 1e6 / steps, sum(other.values()) / 1e6 / steps, gaps / 1e6 / steps, len(qs) // steps))
for n, v in other.most_common(12):
    print("   %7.3f ms/step  %s" % (v / 1e6 / steps, n))
# where the idle gaps of that queue sit: by position in the step (index of the kernel that follows the gap)
per_pos = collections.defaultdict(list)
pos = 0
for i in range(len(qs) - 1):
    if "sa_chain_kernel" in qs[i][2]:
        pos = 0
    pos += 1
    per_pos[(pos, qs[i][2].split("(")[0][:28], qs[i + 1][2].split("(")[0][:28])].append(max(0, qs[i + 1][0] - qs[i][1]))
print("largest average gaps (us) by position after the level-1 block: (position, kernel before, kernel after)")
for k, v in sorted(per_pos.items(), key=lambda kv: -sum(kv[1]) / max(1, steps))[:12]:
    print("   %6.1f us avg  max %7.1f  n=%d  %s" % (sum(v) / len(v) / 1e3, max(v) / 1e3, len(v), k))
