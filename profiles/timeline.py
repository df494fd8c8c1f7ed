#!/usr/bin/env python3
"""Timeline view of a rocprofv3 kernel_trace.csv over the LAST `frac` of the trace: wall span, time with no kernel in
flight, time with kernels of >= 2 streams in flight, and per kernel class its span, its busy time and how much of that
ran beside a kernel of another stream."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
ev = []
iv = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < cut:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:26]
    q = r.get("Queue_Id", r.get("Stream_Id", "0"))
    iv.append((s, e, name, q))
    ev.append((s, 1, q))
    ev.append((e, -1, q))
ev.sort()
span0, span1 = iv[0][0], max(i[1] for i in iv)
active = collections.Counter()
last = span0
idle = multi = 0
# piecewise-constant sets of active queues
pieces = []
for t, d, q in ev:
    if t > last:
        nq = sum(1 for v in active.values() if v > 0)
        if nq == 0:
            idle += t - last
        elif nq >= 2:
            multi += t - last
        pieces.append((last, t, frozenset(k for k, v in active.items() if v > 0)))
        last = t
    active[q] += d
print("span %.1f ms   idle %.1f ms   >=2 queues in flight %.1f ms   queues %s" % (
    (span1 - span0) / 1e6, idle / 1e6, multi / 1e6, sorted(set(i[3] for i in iv))))
import bisect
starts = [p[0] for p in pieces]
acc = collections.OrderedDict()
for s, e, name, q in iv:
    a = acc.setdefault((name, q), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e6
    i = bisect.bisect_right(starts, s) - 1
    while i < len(pieces) and pieces[i][0] < e:
        ps, pe, qs = pieces[i]
        lo, hi = max(ps, s), min(pe, e)
        if hi > lo and len(qs - {q}) > 0:
            a[2] += (hi - lo) / 1e6
        i += 1
print("%-28s %-6s %6s %10s %12s" % ("kernel", "queue", "calls", "busy_ms", "beside_other"))
for (name, q), a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:22]:
    print("%-28s %-6s %6d %10.1f %12.1f" % (name, q, a[0], a[1], a[2]))
