#!/usr/bin/env python3
"""Per kernel class of a rocprofv3 --kernel-trace CSV: launches, sum of durations, UNION of the launches' intervals (time with at
least one launch of the class on the GPU), and -- for k_level_hash -- the `roofline.frac` that follows from them.

    python tools/trace_union.py TRACE_kernel_trace.csv [--perms-per-proof 3735131 --instr 3976] [--json out.json] [--intervals out.csv.gz]

This is the recipe profiles/README.md gives for re-deriving `roofline.frac` of a bench line from the profiler's own trace of
the same command: a proof of the bench trace makes 13 k_level_hash launches (levels 0 .. v - 8 at v = 20) and
hashes `--perms-per-proof` nodes with them (config.keccak_permutations_per_proof minus the 255 x 43 of k_merkle_top), so
    permutations = launches / 13 * perms-per-proof,   achieved = permutations * instr / union,   frac = achieved / 78.64 T
(instr = 3976, the algorithmic count per permutation bench.py uses: the instructions of the cheapest kernel that performs one).
The union cannot exceed the wall clock; the sum of durations of launches that share the chip does (round 3 divided by it).
"""
import argparse
import collections
import csv
import gzip
import json

CLASSES = (("k_level_hash", "level_hash"), ("k_runs_stage", "structure"), ("k_cons_", "structure"), ("k_merkle_top", "top"),
           ("k_radix_fold", "eval"), ("k_keccak_leaves", "dense"), ("k_keccak_level", "dense"), ("k_keccak_small", "tables"))
PEAK = 256 * 4 * 32 * 2.4e9


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > ce:
            tot += ce - cs
            cs, ce = a, b
        elif b > ce:
            ce = b
    return tot + ce - cs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--perms-per-proof", type=float, default=0.0)
    ap.add_argument("--launches-per-proof", type=int, default=13)
    ap.add_argument("--instr", type=float, default=3976.0, help="algorithmic VALU instructions per permutation (bench.py: IC_ALG)")
    ap.add_argument("--json")
    ap.add_argument("--intervals", help="write (class, start_ns, end_ns) of every classified launch, gzip CSV")
    ap.add_argument("--last", type=int, default=0, help="only the last N launches of every class -- with N = roofline.launches of the "
                    "bench line of the same --no-extras run: the launches of its roofline leg, the last thing such a run does")
    a = ap.parse_args()
    per = collections.defaultdict(list)
    if a.csv.endswith(".gz"):  # an intervals file written by an earlier run of this tool (the full trace is not kept: ~10 MB)
        for r in csv.DictReader(gzip.open(a.csv, "rt")):
            per[r["class"]].append((int(r["start_ns"]), int(r["end_ns"])))
    else:
        for r in csv.DictReader(open(a.csv)):
            name = r["Kernel_Name"]
            for key, cls in CLASSES:
                if key in name:
                    per[cls].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
                    break
    if a.intervals:  # (all of them, before --last)
        t00 = min(s for v in per.values() for s, _ in v)
        with gzip.open(a.intervals, "wt") as f:
            f.write("class,start_ns,end_ns\n")
            for cls, iv in per.items():
                for s, e in sorted(iv):
                    f.write("%s,%d,%d\n" % (cls, s - t00, e - t00))
    if a.last:
        n_lh = len(per.get("level_hash", []))
        for cls in list(per):
            k = max(1, round(a.last * len(per[cls]) / max(n_lh, 1)))  # the same share of every class's launches
            per[cls] = sorted(per[cls])[-k:]
    allv = [x for v in per.values() for x in v]
    t0, t1 = min(s for s, _ in allv), max(e for _, e in allv)
    out = {"source": a.csv.split("/")[-1], "last": a.last or None, "wall_us_first_to_last_launch": (t1 - t0) / 1e3, "classes": {}}
    for cls, iv in per.items():
        u = union(iv)
        out["classes"][cls] = {"launches": len(iv), "sum_of_durations_us": sum(e - s for s, e in iv) / 1e3, "union_us": u / 1e3,
                               "avg_launch_us": sum(e - s for s, e in iv) / 1e3 / len(iv), "overlap": sum(e - s for s, e in iv) / u}
    if a.perms_per_proof and "level_hash" in per:
        lh = out["classes"]["level_hash"]
        perms = lh["launches"] / a.launches_per_proof * a.perms_per_proof
        lh["permutations"] = perms
        lh["achieved_T_lane_instr_s"] = perms * a.instr / (lh["union_us"] / 1e6) / 1e12
        lh["frac"] = perms * a.instr / (lh["union_us"] / 1e6) / PEAK
        lh["frac_by_sum_of_durations"] = perms * a.instr / (lh["sum_of_durations_us"] / 1e6) / PEAK
    print(json.dumps(out, indent=1))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
