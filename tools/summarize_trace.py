#!/usr/bin/env python3
"""Summaries of a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv):

    python tools/summarize_trace.py TRACE.csv --stats            # per-kernel calls / total / average, by share
    python tools/summarize_trace.py TRACE.csv --last-build       # every launch of the last Merkle build in the file, in order
"""
import argparse
import csv
import re


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("zk::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--last-build", action="store_true")
    ap.add_argument("--out")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    lines = []
    if a.stats:
        tot = {}
        for r in rows:
            k = short(r["Kernel_Name"])
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            c, t = tot.get(k, (0, 0))
            tot[k] = (c + 1, t + d)
        s = sum(t for _, t in tot.values())
        lines.append("kernel,calls,total_us,avg_us,share")
        for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
            lines.append("%s,%d,%.1f,%.2f,%.4f" % (k, c, t / 1e3, t / c / 1e3, t / s))
    if a.last_build:
        starts = [i for i, r in enumerate(rows) if "k_runs_stage<true>" in r["Kernel_Name"] or "k_keccak_leaves" in r["Kernel_Name"]]
        i0 = starts[-1] if starts else 0
        t0 = int(rows[i0]["Start_Timestamp"])
        lines.append("kernel,workitems,start_us,duration_us")
        for r in rows[i0:]:
            k = short(r["Kernel_Name"])
            lines.append("%s,%s,%.1f,%.1f" % (k, r.get("Grid_Size", r.get("Grid_Size_X", "")), (int(r["Start_Timestamp"]) - t0) / 1e3,
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
            if "k_gather_nodes" in k:
                break
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    else:
        print(text, end="")


if __name__ == "__main__":
    main()
