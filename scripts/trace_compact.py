"""rocprofv3 kernel-trace csv -> a compact gzip csv (start_ns, end_ns, queue, grid, workgroup, kernel name) small enough to travel back
through gpurun_out/.     python scripts/trace_compact.py <dir> <out.csv.gz>"""
import csv, glob, gzip, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), r.get("Grid_Size_X", r.get("Grid_Size", "")),
                     r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r["Kernel_Name"][:160]))
rows.sort()
t0 = rows[0][0] if rows else 0
with gzip.open(sys.argv[2], "wt", newline="") as g:
    w = csv.writer(g)
    w.writerow(["start_ns", "end_ns", "queue", "grid", "wg", "kernel"])
    for s, e, q, gr, wg, n in rows:
        w.writerow([s - t0, e - t0, q, gr, wg, n])
print("wrote", len(rows), "rows")
