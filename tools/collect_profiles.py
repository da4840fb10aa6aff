#!/usr/bin/env python3
"""Copies what tools/regen_profiles.sh left under gpurun_out/p2_* into profiles/ (r02_ names) and derives the per-kernel HBM
traffic (profiles/r02_traffic.json, profiles/keccak_traffic.json) from the two PMC passes.  Run from the repo root after the
gpurun call has merged its output."""
import collections
import csv
import glob
import json
import os
import shutil

ROUND = "r02"


def newest(pattern):
    files = glob.glob(pattern)
    assert files, pattern
    return max(files, key=os.path.getmtime)


def counter(path, name):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            out[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    ff = newest("gpurun_out/p2_pmc_FETCH_SIZE/*/*counter_collection.csv")
    wf = newest("gpurun_out/p2_pmc_WRITE_SIZE/*/*counter_collection.csv")
    vf = newest("gpurun_out/p2_pmc_VALU/*/*counter_collection.csv")
    F, W = counter(ff, "FETCH_SIZE"), counter(wf, "WRITE_SIZE")
    V1, V2 = counter(vf, "SQ_INSTS_VALU"), counter(vf, "SQ_ACTIVE_INST_VALU2")
    kern = json.load(open("gpurun_out/p2_kernels.json"))["kernels"]
    # The per-kernel leg launches, per kernel name, a known sequence: the measured launches of bench.py --kernels are the
    # LAST `launches` dispatches of each (kernel name, grid) group; set-up launches (fills, the cache-flush sweeps of
    # k_block_sums over the 1 GiB buffer) are told apart by their counter value, so per-kernel figures below take the
    # launches whose byte count is closest to the algorithmic one.
    traffic = {}
    N = 1 << 20
    want = {  # kernel name prefix -> (label, algorithmic bytes per launch)
        "k_bind_vec<false>": ("k_bind_vec[43x2^20]", 6 * 43 * N),
        "k_bind_vec<true>": ("k_bind_vec_sums[43x2^20]", 6 * 43 * N),
        "k_radix_fold<true>": ("k_radix_fold[43x2^20]", 43 * (4 * N + 16 * 1024 * 8)),
        "k_block_sums": ("k_half_sums[43x2^20]", 4 * 43 * N),
        "k_keccak_leaves": ("k_keccak_leaves[43x2^20]", 43 * N * 36),
        "k_keccak_level<4>": ("k_keccak_level[43x2^20]", 43 * (N // 2) * 96),
    }
    for key, (label, alg) in want.items():
        fk = [k for k in F if k.replace("void zk::", "").replace("zk::", "").startswith(key)]
        wk = [k for k in W if k.replace("void zk::", "").replace("zk::", "").startswith(key)]
        if not fk or not wk:
            continue
        # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md): x2 on the fetch side
        fetch = [x[0] * 2048 for k in fk for x in F[k]]
        write = [x[0] * 1024 for k in wk for x in W[k]]
        # launches of the 43 x 2^20 shape: fetch + write closest to the algorithmic bytes (within a factor of 2)
        pairs = [(f, w) for f, w in zip(sorted(fetch), sorted(write)) if 0.5 * alg < f + w < 2.0 * alg]
        if not pairs:
            pairs = list(zip(sorted(fetch), sorted(write)))[-3:]
        per = sum(f + w for f, w in pairs) / len(pairs)
        traffic[label] = {"hbm_bytes_per_launch": per, "fetch_bytes": sum(f for f, _ in pairs) / len(pairs),
                          "write_bytes": sum(w for _, w in pairs) / len(pairs), "algorithmic_bytes_per_launch": alg,
                          "ratio": per / alg, "launches_used": len(pairs)}
    src = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --kernels --kernel-iters 10; "
           "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads), KB -> bytes x1024")
    json.dump({"source": src, "kernels": traffic}, open("profiles/%s_traffic.json" % ROUND, "w"), indent=1)
    if "k_keccak_leaves[43x2^20]" in traffic:
        t = traffic["k_keccak_leaves[43x2^20]"]
        json.dump({"kernel": "k_keccak_leaves", "source": "profiles/%s_pmc_fetch_size.csv + %s_pmc_write_size.csv" % (ROUND, ROUND),
                   "hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": t["algorithmic_bytes_per_launch"],
                   "hashes_per_launch": 43 * N, "hbm_bytes_per_hash": t["hbm_bytes_per_launch"] / (43 * N), "algorithmic_bytes_per_hash": 36},
                  open("profiles/keccak_traffic.json", "w"), indent=1)
    valu = {}
    for k in V1:
        if "keccak" in k and k in V2:
            a, b = sum(x[0] for x in V1[k]), sum(x[0] for x in V2[k])
            valu[k.replace("void zk::", "")] = {"SQ_INSTS_VALU": a, "SQ_ACTIVE_INST_VALU2": b, "valu2_share": b / a if a else None}
    json.dump(valu, open("profiles/%s_valu2_share.json" % ROUND, "w"), indent=1)
    for k, v in traffic.items():
        print("%-28s %8.2f MB per launch vs %8.2f MB algorithmic (%.3f)" % (k, v["hbm_bytes_per_launch"] / 1e6, v["algorithmic_bytes_per_launch"] / 1e6, v["ratio"]))
    for k, v in valu.items():
        print("%-40s valu2 share %.3f" % (k[:40], v["valu2_share"]))
    shutil.copy(ff, "profiles/%s_pmc_fetch_size.csv" % ROUND)
    shutil.copy(wf, "profiles/%s_pmc_write_size.csv" % ROUND)
    shutil.copy(vf, "profiles/%s_pmc_valu.csv" % ROUND)
    for tag in ("kernels", "b1", "bdef", "lasso", "sumcheck"):
        shutil.copy(newest("gpurun_out/p2_%s/*/*kernel_stats.csv" % tag), "profiles/%s_%s_kernel_stats.csv" % (ROUND, tag))
    for src_, dst in (("p2_kernels.json", "kernels.json"), ("p2_kernels_plain.json", "kernels_unprofiled.json"),
                      ("p2_bench.json", "final_bench.json"), ("p2_bench_b1.json", "final_bench_b1.json"),
                      ("p2_bench_s0.json", "final_bench_own_thread_transcripts.json"),
                      ("p2_bench_s0_b8.json", "final_bench_own_thread_transcripts_b8.json"),
                      ("p2_bench_s4.json", "final_bench_5x8.json"), ("p2_bench_s6.json", "final_bench_7x8.json"),
                      ("p2_bench_tables_b8.json", "final_bench_merkle_tables_b8.json"),
                      ("p2_bench_dense_b8.json", "final_bench_merkle_dense_b8.json"), ("p2_bench_all.json", "final_bench_merkle_all.json"),
                      ("p2_gpu_bound.txt", "gpu_bound_rate.txt"),
                      ("p2_bench_gpus2_rehearsal.json", "bench_gpus2_rehearsal.json"), ("p2_lasso.json", "lasso.json"),
                      ("p2_sumcheck.json", "sumcheck.json"), ("p2_extra.json", "extra.json"), ("p2_configs.jsonl", "configs.jsonl")):
        if os.path.exists(os.path.join("gpurun_out", src_)):
            shutil.copy(os.path.join("gpurun_out", src_), "profiles/%s_%s" % (ROUND, dst))
    for tag in ("kernels", "b1", "bdef"):
        print(tag)
        for r in list(csv.DictReader(open("profiles/%s_%s_kernel_stats.csv" % (ROUND, tag))))[:8]:
            print("  %-60s calls=%5s avg_us=%10.2f pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r.get("Percentage")))
    one = glob.glob("gpurun_out/p2_one/**/*kernel_trace.csv", recursive=True)
    if one:  # the launches of the last of three lone proofs, in order: name, grid, start (us from the first), duration (us)
        rows = sorted(csv.DictReader(open(max(one, key=os.path.getmtime))), key=lambda r: int(r["Start_Timestamp"]))
        rows = [r for r in rows if "keccak" in r["Kernel_Name"] or "k_runs" in r["Kernel_Name"] or "k_cons" in r["Kernel_Name"]]
        last = rows[-(len(rows) // 3):]
        t0 = int(last[0]["Start_Timestamp"])
        with open("profiles/%s_one_proof_launches.csv" % ROUND, "w") as f:
            f.write("kernel,grid_x,grid_y,start_us,duration_us\n")
            for r in last:
                f.write("%s,%s,%s,%.1f,%.1f\n" % (r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Grid_Size_X", ""),
                                                  r.get("Grid_Size_Y", ""), (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                  (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    for name in ("final_bench", "final_bench_b1", "final_bench_own_thread_transcripts", "final_bench_own_thread_transcripts_b8",
                 "final_bench_5x8", "final_bench_7x8", "final_bench_merkle_tables_b8", "final_bench_merkle_dense_b8",
                 "final_bench_merkle_all"):
        pth = "profiles/%s_%s.json" % (ROUND, name)
        if not os.path.exists(pth):
            continue
        d = json.load(open(pth))
        r = d["roofline"]
        print("%-28s %6.1f M steps/s  %.2f ms/proof  keccak frac %.3f  bind %.3f  single %.1f ms  pcie %.1f M" %
              (name, d["value"] / 1e6, d["config"]["ms_per_proof_per_gpu"], r.get("frac", 0), r.get("bind_hbm_frac", 0),
               d["config"].get("single_proof_ms", 0), d["config"].get("pcie_inclusive_value", 0) / 1e6))


if __name__ == "__main__":
    main()
