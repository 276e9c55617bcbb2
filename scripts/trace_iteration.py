"""One steady-state training iteration out of a compact kernel trace (scripts/trace_compact.py): every kernel with its start offset,
duration, the idle time in front of it, queue and grid.   python scripts/trace_iteration.py <trace.csv.gz> [iteration=20] [queue]"""
import csv, gzip, sys
rows = [r for r in csv.DictReader(gzip.open(sys.argv[1], "rt"))]
R = [(int(r["start_ns"]), int(r["end_ns"]), r["queue"], r["grid"], r["kernel"]) for r in rows]
adam = [r[0] for r in R if "multi_tensor_apply" in r[4]]
starts = [adam[0]]
for a, b in zip(adam, adam[1:]):
    if b - a > 20e6:
        starts.append(b)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
a, b = starts[k], starts[k + 1]
only = sys.argv[3] if len(sys.argv) > 3 else None
print("# iteration %d: %.2f ms; all iterations: %s" % (k, (b - a) / 1e6, " ".join("%.1f" % ((y - x) / 1e6) for x, y in zip(starts, starts[1:]))))
busy_end = a
for s, e, q, g, n in R:
    if not (a <= s < b):
        continue
    gap = max(0.0, (s - busy_end) / 1e3)
    busy_end = max(busy_end, e)
    if only is None or q == only:
        print("%8.3f %7.1f %6.1f q%s g%s %s" % ((s - a) / 1e6, (e - s) / 1e3, gap, q, g, n.replace("void ", "").replace("at::native::", "")[:90]))
