#!/usr/bin/env python3
"""Per-kernel sums of every counter in a rocprofv3 --pmc counter_collection.csv (kernels whose name contains argv[2])."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]
    if pat not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, {c: "%.3g" % v for c, v in sorted(d.items())}, "calls", max(calls[(k, c)] for c in d))
