#!/usr/bin/env python3
"""Per-kernel sums of every counter of a rocprofv3 --pmc run (counter_collection.csv found under argv[1]), with the summed
dispatch time and the effective clock where GRBM_GUI_ACTIVE was collected (MI355X_MICROARCH.md: GRBM_GUI_ACTIVE / 8 / time)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(dict)
meta = {}
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
        if pat not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        meta[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size"))
for k, d in sorted(acc.items()):
    t = sum(dur[k].values())
    line = {c: "%.4g" % v for c, v in sorted(d.items())}
    extra = ""
    if t and "GRBM_GUI_ACTIVE" in d:
        extra = " eff_clock_GHz %.3f" % (d["GRBM_GUI_ACTIVE"] / 8 / t / 1e9)
    print(k, "dispatches", len(dur[k]), "time_s %.6f" % t, "vgpr/agpr/sgpr/lds/scratch/grid", meta[k], line, extra)
