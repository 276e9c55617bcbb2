#!/usr/bin/env python
"""Turns a rocprofv3 results .db (kernel-trace) into the text summary kept under profiles/.

    python scripts/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.txt
    python scripts/rocprof_summary.py x_results.db --json profiles/rocprof_launch_ms.json --tag rNN_x --steps 37

``--json``: also write the per-FAMILY launch durations (the kernel families of bench.py's ``roofline``: one entry per
device kernel family, scripts/collect_pmc.py's naming) that ``bench.py`` quotes as ``roofline.rocprof_avg_launch_ms`` beside its
own HIP-event figure, so that the line's fraction can be recomputed from a committed file.
"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def rows_of(path):
    cur = sqlite3.connect(path).cursor()
    return list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))


def main(path, top=45, json_path=None, tag=None, steps=None, batch=8, points=25600):
    rows = rows_of(path)
    total = sum(float(r[2]) for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# total kernel time %.3f ms over %d distinct kernels; durations in microseconds" % (total / 1e3, len(rows)))
    print("%-100s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows[:top]:
        print("%-100s %8d %14.1f %12.2f %7.2f" % (str(name)[:100], int(calls), float(tot), float(avg),
                                                  float(pct)))
    if json_path:
        from collect_pmc import family, short_name
        fams = {}
        for name, calls, tot, avg, pct in rows:
            fam = family(str(name))
            if fam is None:
                continue
            f = fams.setdefault(fam, {"calls": 0, "total_us": 0.0, "kernel_names": []})
            f["calls"] += int(calls)
            f["total_us"] += float(tot)
            f["kernel_names"].append(short_name(str(name)))
        for f in fams.values():
            f["avg_launch_ms"] = round(f["total_us"] / max(f["calls"], 1) / 1e3, 5)
            f["total_us"] = round(f["total_us"], 1)
            f["kernel_names"].sort()
        out = {"source": "profiles/%s_pipeline_kernel_stats.txt" % tag if tag else os.path.basename(path),
               "what": "rocprofv3 --kernel-trace --stats of `python bench.py` (scripts/collect_profiles.sh), per kernel family",
               "workload": {"batch": int(batch), "points": int(points)},     # bench.py hands the durations out to such runs only
               "steps_traced_incl_warmup": steps, "total_kernel_ms": round(total / 1e3, 3), "families": fams}
        with open(json_path, "w") as fh:
            json.dump(out, fh, indent=1, sort_keys=True)
            fh.write("\n")


if __name__ == "__main__":
    args = sys.argv[1:]
    opts = {}
    for flag in ("--json", "--tag", "--steps", "--batch", "--points"):
        if flag in args:
            i = args.index(flag)
            opts[flag] = args[i + 1]
            del args[i:i + 2]
    main(args[0], json_path=opts.get("--json"), tag=opts.get("--tag"),
         steps=int(opts["--steps"]) if "--steps" in opts else None, batch=int(opts.get("--batch", 8)),
         points=int(opts.get("--points", 25600)))
