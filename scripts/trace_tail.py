"""Top kernels of the LAST part of a rocprofv3 kernel trace (steady state of a short run): python trace_tail.py <dir> [frac or ms] [rows]"""
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
# frac < 1: that fraction of the trace's span; frac >= 1: that many milliseconds before the last kernel ends
t0 = rows[-1][1] - ((rows[-1][1] - rows[0][0]) * frac if frac < 1 else frac * 1e6)
sel = [r for r in rows if r[0] >= t0]
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in sel:
    tot[n[:90]] += e - s; cnt[n[:90]] += 1
span = (sel[-1][1] - sel[0][0]) / 1e6
print("window %.1f ms, kernel time %.1f ms, %d kernels" % (span, sum(tot.values()) / 1e6, len(sel)))
for n, v in tot.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 22):
    print("%9.2f ms %6d  %s" % (v / 1e6, cnt[n], n))
