#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE / WRITE_SIZE from two rocprofv3 --pmc passes (counter_collection.csv).
Units: the counters are in KiB (MI355X_MICROARCH.md: hbm_bytes = value * 1024); on gfx950 FETCH_SIZE
reads exactly half of a WIDE coalesced stream's bytes (16 B/lane) -- other access widths are
uncalibrated, so both the raw and the doubled read figure are printed."""
import csv, sys, collections
def load(path, name):
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"].split("(")[0][:28]
        a = acc[k]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return acc
f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
print("%-30s %6s %14s %14s %14s %10s" % ("kernel", "calls", "fetch_MB/call", "fetch_x2_MB", "write_MB/call", "ms/call"))
for k in sorted(f, key=lambda k: -f[k][2]):
    if not k.startswith("k_") and not k.startswith("void k_"): continue
    n = f[k][0]; fm = f[k][1] * 1024 / n / 1e6; wm = (w[k][1] * 1024 / max(1, w[k][0]) / 1e6) if k in w else float("nan")
    print("%-30s %6d %14.2f %14.2f %14.2f %10.3f" % (k, n, fm, 2 * fm, wm, f[k][2] / n))
