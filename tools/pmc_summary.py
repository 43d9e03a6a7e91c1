#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: per kernel name x counter -> mean value per dispatch."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = (row.get("Kernel_Name", "?")[:90], row.get("Counter_Name", "?"))
        acc[k][0] += float(row.get("Counter_Value", 0))
        acc[k][1] += 1
print(f"{'kernel':90s} {'counter':28s} {'dispatches':>10s} {'mean_per_dispatch':>20s}")
for (k, c), (tot, n) in sorted(acc.items()):
    print(f"{k:90s} {c:28s} {n:10d} {tot / n:20.1f}")
