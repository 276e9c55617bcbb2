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
