"""Where is the GPU idle?  From a rocprofv3 kernel trace (csv): the union of all kernels' busy intervals over the last
`window` ms, every gap longer than `min_gap` us with the kernel that ended before it and the one that started after it,
and the total by (before -> after) pair.     python scripts/trace_gaps.py <dir> [window_ms=600] [min_gap_us=40]"""
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
window = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
t0 = rows[-1][1] - window * 1e6
sel = [r for r in rows if r[0] >= t0]
busy_end, last_name = sel[0][1], sel[0][2]
gaps = []
busy = 0
cur_start = sel[0][0]
for s, e, n in sel[1:]:
    if s > busy_end:
        gaps.append((s - busy_end, last_name, n, busy_end))
        busy += busy_end - cur_start
        cur_start = s
    if e > busy_end:
        busy_end, last_name = e, n
busy += busy_end - cur_start
span = sel[-1][1] - sel[0][0]
print("window %.1f ms: busy %.1f ms, idle %.1f ms in %d gaps" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps)))
big = [g for g in gaps if g[0] >= min_gap * 1e3]
print("gaps >= %.0f us: %d, total %.1f ms" % (min_gap, len(big), sum(g[0] for g in big) / 1e6))
pair = collections.Counter(); cnt = collections.Counter()
for g in big:
    pair[(g[1], g[2])] += g[0]; cnt[(g[1], g[2])] += 1
for (a, b), v in pair.most_common(25):
    print("%8.2f ms %4d  %-60s -> %s" % (v / 1e6, cnt[(a, b)], a, b))
hist = collections.Counter()
for g in gaps:
    us = g[0] / 1e3
    hist["<5us" if us < 5 else "<20us" if us < 20 else "<40us" if us < 40 else "<200us" if us < 200 else "<1ms" if us < 1000 else ">=1ms"] += g[0]
print("idle by gap length:", {k: round(v / 1e6, 2) for k, v in hist.items()})
