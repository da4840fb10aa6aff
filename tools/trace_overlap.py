#!/usr/bin/env python3
"""How much do the kernels of a rocprofv3 --kernel-trace CSV overlap?  Over the busiest contiguous stretch of the trace (the
last `--tail` fraction of the launches): wall time, summed kernel time, time with >= 1 / >= 2 / >= 4 kernels running.
    python tools/trace_overlap.py TRACE_kernel_trace.csv [--tail 0.5]
"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--tail", type=float, default=0.5)
a = ap.parse_args()
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(a.csv))))
rows = rows[int(len(rows) * (1 - a.tail)):]
ev = []
for s, e in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
t0, t1 = rows[0][0], max(e for _, e in rows)
busy = {1: 0, 2: 0, 4: 0, 8: 0}
n, last = 0, t0
for t, d in ev:
    for k in busy:
        if n >= k:
            busy[k] += t - last
    n += d
    last = t
wall = t1 - t0
tot = sum(e - s for s, e in rows)
print("launches %d  wall %.1f ms  summed kernel time %.1f ms (x%.2f)  >=1 running %.2f  >=2 %.2f  >=4 %.2f  >=8 %.2f of the wall" %
      (len(rows), wall / 1e6, tot / 1e6, tot / wall, busy[1] / wall, busy[2] / wall, busy[4] / wall, busy[8] / wall))
