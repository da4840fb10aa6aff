#!/usr/bin/env python3
"""How full are the hardware queues, and with what?  From a rocprofv3 --kernel-trace CSV of a bench run: the histogram of
kernels running at once, the share of the wall clock with at least one chip-filling hash launch running (k_level_hash with
>= 1536 workgroups' worth of entries cannot be told from the grid -- the grid is an upper bound -- so: duration-weighted, by
level, from the launch order), and per kernel name: launches, average duration, sum of durations / wall (= queue slots held).

    python tools/trace_concurrency.py TRACE_kernel_trace.csv [--tail-ms 150] [--json out.json]
"""
import argparse
import collections
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--tail-ms", type=float, default=150.0, help="the last so many ms of the trace (steady state)")
    ap.add_argument("--json")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.csv)):
        n = r["Kernel_Name"]
        short = n.split("(")[0].replace("void ", "").replace("zk::", "")
        wi = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) * max(1, int(r.get("Grid_Size_Z", "1") or 1))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, wi, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    t1 = max(b for _, b, *_ in rows)
    t0 = t1 - int(a.tail_ms * 1e6)
    rows = [x for x in rows if x[0] >= t0]
    wall = t1 - min(x[0] for x in rows)
    ev = []
    for s, e, n, wi, q, st in rows:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort()
    cnt = 0
    hist = collections.Counter()
    hash_on = 0
    nhash = 0
    last = ev[0][0]
    for t, d, n in ev:
        hist[cnt] += t - last
        if nhash:
            hash_on += t - last
        last = t
        cnt += d
        if n.startswith("k_level_hash") or n.startswith("k_keccak"):
            nhash += d
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n, wi, q, st in rows:
        per[n][0] += 1
        per[n][1] += e - s
    queues = collections.Counter(q for *_, q, _ in rows)
    out = {"wall_ms": wall / 1e6, "kernels_at_once": {k: round(v / wall, 4) for k, v in sorted(hist.items())},
           "mean_kernels_at_once": sum(k * v for k, v in hist.items()) / wall,
           "share_of_wall_with_a_hash_launch_running": hash_on / wall,
           "queues": dict(queues),
           "per_kernel": {n: {"launches": c, "avg_us": d / c / 1e3, "queue_slots_held": d / wall} for n, (c, d) in
                          sorted(per.items(), key=lambda kv: -kv[1][1])}}
    s = json.dumps(out, indent=1)
    if a.json:
        open(a.json, "w").write(s)
    print(s)


if __name__ == "__main__":
    main()
