#!/usr/bin/env python
"""Turns a rocprofv3 results .db (kernel-trace) into the text summary kept under profiles/.

    python scripts/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=45):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(float(r[2]) for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# total kernel time %.3f ms over %d distinct kernels; durations in microseconds" % (total / 1e3, len(rows)))
    print("%-100s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows[:top]:
        print("%-100s %8d %14.1f %12.2f %7.2f" % (str(name)[:100], int(calls), float(tot), float(avg),
                                                  float(pct)))


if __name__ == "__main__":
    main(sys.argv[1])
