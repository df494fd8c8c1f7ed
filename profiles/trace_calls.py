#!/usr/bin/env python3
"""Per-call view of a rocprofv3 kernel_trace.csv: for every pz kernel, calls / total / mean over the LAST `frac`
of the trace (skips warm-up), plus the per-call list for the MSM sort/accumulate kernels."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
acc = collections.OrderedDict()
for r in rows:
    if int(r["Start_Timestamp"]) < cut: continue
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    a = acc.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in acc.values())
print("%-30s %6s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for n, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%-30s %6d %12.0f %10.1f %6.2f" % (n, a[0], a[1], a[1] / a[0], 100 * a[1] / tot))
