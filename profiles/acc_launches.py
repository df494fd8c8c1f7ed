#!/usr/bin/env python3
"""per-launch durations of one kernel from a rocprofv3 *_kernel_trace.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(sys.argv[2])]
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print("%s grid=(%s,%s) wg=%s vgpr=%s dur_ms=%.3f" % (r["Kernel_Name"][:24], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], r.get("VGPR_Count", "?"), d))
