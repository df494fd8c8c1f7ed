#!/usr/bin/env python3
"""Trim a rocprofv3 *_kernel_stats.csv to a readable table (kernel names cut to 64 chars)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
title = sys.argv[2] if len(sys.argv) > 2 else ""
print("# " + title)
print("%-66s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print("%-66s %6s %12.0f %12.1f %7.2f" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                              float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
