#!/usr/bin/env python3
"""Copies what tools/regen_profiles.sh left under gpurun_out/fin_* into profiles/ (round-1 names) and derives
profiles/bind_traffic.json from the two PMC passes.  Run from the repo root after the gpurun call has merged its output."""
import collections
import csv
import glob
import json
import os
import shutil

ROUND = "r01"


def newest(pattern):
    files = glob.glob(pattern)
    assert files, pattern
    return max(files, key=os.path.getmtime)


def counter(path, name):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            out[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]),
                                                        int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    ff = newest("gpurun_out/fin_pmc_FETCH_SIZE/*/*counter_collection.csv")
    wf = newest("gpurun_out/fin_pmc_WRITE_SIZE/*/*counter_collection.csv")
    F, W = counter(ff, "FETCH_SIZE"), counter(wf, "WRITE_SIZE")
    k = [x for x in F if "k_radix_fold" in x][0]
    f, w = F[k], W[k]
    n = 1 << 20
    alg = 43 * (n * 4 + 16 * 1024 * 8)
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md): x2 on the fetch side
    per = (sum(x[0] for x in f) * 2048 + sum(x[0] for x in w) * 1024) / len(f)
    json.dump({"kernel": "k_radix_fold",
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1 "
                         "--batch 1 --no-cpu-baseline; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of "
                         "wide coalesced reads), KB -> bytes x1024",
               "launches": len(f), "hbm_bytes_per_launch": per, "algorithmic_bytes_per_launch": alg,
               "avg_launch_us_under_pmc": sum(x[1] for x in f) / len(f) / 1e3}, open("profiles/bind_traffic.json", "w"), indent=1)
    print("k_radix_fold: %.2f MB per launch vs %.2f MB algorithmic (%.4f), %.1f us under PMC" %
          (per / 1e6, alg / 1e6, per / alg, sum(x[1] for x in f) / len(f) / 1e3))
    shutil.copy(ff, "profiles/%s_pmc_fetch_size.csv" % ROUND)
    shutil.copy(wf, "profiles/%s_pmc_write_size.csv" % ROUND)
    shutil.copy(newest("gpurun_out/fin_b1/*/*kernel_stats.csv"), "profiles/%s_b1_kernel_stats.csv" % ROUND)
    shutil.copy(newest("gpurun_out/fin_bdef/*/*kernel_stats.csv"), "profiles/%s_bdef_kernel_stats.csv" % ROUND)
    for src, dst in (("fin_bench.json", "final_bench.json"), ("fin_bench_b1.json", "final_bench_b1.json"),
                     ("fin_bench_b6.json", "final_bench_b6.json"), ("fin_bench_b8.json", "final_bench_b8.json"), ("fin_bench_dedup8.json", "final_bench_dedup_b8.json"),
                     ("fin_extra.json", "extra.json"), ("fin_configs.jsonl", "configs.jsonl"),
                     ("fin_merkle_rate.txt", "merkle_rate.txt"), ("fin_fold_rate.txt", "fold_rate.txt"),
                     ("fin_valu_rate.txt", "valu_rate.txt"), ("fin_bank_rate.txt", "bank_rate.txt"),
                     ("fin_valu2_rate.txt", "valu2_rate.txt")):
        shutil.copy(os.path.join("gpurun_out", src), "profiles/%s_%s" % (ROUND, dst))
    for tag in ("b1", "bdef"):
        print(tag)
        for r in list(csv.DictReader(open("profiles/%s_%s_kernel_stats.csv" % (ROUND, tag))))[:6]:
            print("  %-55s calls=%5s avg_us=%10.2f" % (r["Name"][:55], r["Calls"], float(r["AverageNs"]) / 1e3))
    for name in ("final_bench", "final_bench_b1", "final_bench_b6", "final_bench_b8", "final_bench_dedup_b8"):
        d = json.load(open("profiles/%s_%s.json" % (ROUND, name)))
        r = d["roofline"]
        print("%-22s %6.1f M steps/s  %.2f ms/proof  roofline %.3f (uncontended %.3f, %.1f us)  merkle %.2f ms" %
              (name, d["value"] / 1e6, d["config"]["ms_per_proof_per_gpu"], r["frac"], r["uncontended"]["frac"],
               r["avg_launch_us"], d["kernels"]["uncontended"]["merkle_build_ms"]))


if __name__ == "__main__":
    main()
