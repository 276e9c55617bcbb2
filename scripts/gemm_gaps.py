"""Timeline analysis of a rocprofv3 --kernel-trace CSV of `python bench.py`: how much of the steady-state
wall time has at least one MFMA kernel (mlp_gemm_kernel / sa_chain_kernel) running, and what runs in the gaps.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --cpu-scenes 0 --steps 20
    python scripts/gemm_gaps.py gpurun_out/kt
"""
import collections, csv, glob, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
g_all = [r for r in rows if "mlp_gemm" in r[2] or "sa_chain" in r[2]]
# steady state: from 40 % to 90 % of the GEMM launches (skips model set-up and warm-up)
a, b = g_all[int(len(g_all) * 0.4)][0], g_all[int(len(g_all) * 0.9)][1]
sel = [r for r in rows if r[0] >= a and r[1] <= b]
gemm = [r for r in sel if "mlp_gemm" in r[2] or "sa_chain" in r[2]]
iv = sorted((s, e) for s, e, _ in gemm)
merged = []
for s, e in iv:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
busy = sum(e - s for s, e in merged)
span = merged[-1][1] - merged[0][0]
print("window %.2f ms: >=1 GEMM running %.2f ms (%.1f %%), sum of GEMM durations %.2f ms" %
      (span / 1e6, busy / 1e6, 100.0 * busy / span, sum(e - s for s, e, _ in gemm) / 1e6))
gaps = [(merged[i + 1][0] - merged[i][1], merged[i][1], merged[i + 1][0]) for i in range(len(merged) - 1)]
hist = collections.Counter()
for g, s, e in gaps:
    hist["<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else ">=100us"] += g
print("gap time by gap length (ms):", {k: round(v / 1e6, 3) for k, v in hist.items()}, "n_gaps", len(gaps))
in_gap = collections.Counter()
for g, s, e in gaps:
    if g < 2e4:
        continue
    for ks, ke, name in sel:
        ov = min(ke, e) - max(ks, s)
        if ov > 0:
            in_gap[name] += ov
print("kernels overlapping gaps >= 20 us (ms of overlap):")
for name, v in in_gap.most_common(12):
    print("   %8.3f  %s" % (v / 1e6, name))
