#!/usr/bin/env python3
"""Idle gaps on one queue of a rocprofv3 kernel_trace.csv over the last `frac` of the trace: total busy / idle time and the
largest gaps with the kernels on either side."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
byq = {}
for r in rows:
    byq.setdefault(r["Queue_Id"], []).append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
    gaps = []
    for a, b in zip(rs, rs[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g > 0:
            gaps.append((g, a["Kernel_Name"].split("(")[0][-24:], b["Kernel_Name"].split("(")[0][-24:]))
    gaps.sort(reverse=True)
    print("queue %s: %d kernels, span %.1f ms, busy %.1f ms, idle %.1f ms (gaps > 20 us: %d, sum %.1f ms)" % (
        q, len(rs), span / 1e6, busy / 1e6, (span - busy) / 1e6, sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
    for g in gaps[:12]:
        print("    %8.1f us  after %-24s before %s" % (g[0] / 1e3, g[1], g[2]))
