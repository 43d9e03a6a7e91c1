#!/usr/bin/env python3
"""tools/merge_kernel_stats.py <dir of <workload>.csv from rocprofv3 --kernel-trace --stats> -> one CSV on stdout with a workload column:
per workload the kernels that take >= 0.5 % of its GPU time (generators, fills and torch copies of the set-up included)."""
import csv, glob, os, sys
w = csv.writer(sys.stdout)
w.writerow(["workload", "kernel", "calls", "total_us", "average_us", "min_us", "max_us", "percent_of_workload"])
for path in sorted(glob.glob(os.path.join(sys.argv[1], "*.csv"))):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        pct = float(r.get("Percentage", 0) or 0)
        if pct < 0.5:
            continue
        ns = lambda k: float(r.get(k, 0) or 0) / 1e3
        w.writerow([os.path.splitext(os.path.basename(path))[0], r.get("Name", "?")[:160], r.get("Calls"), round(ns("TotalDurationNs"), 1),
                    round(ns("AverageNs"), 2), round(ns("MinNs"), 2), round(ns("MaxNs"), 2), pct])
