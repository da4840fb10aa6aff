#!/usr/bin/env python3
"""Copies what tools/regen_profiles_r04.sh left under gpurun_out/p4_* into profiles/ (r04_ names) and derives, per kernel of ONE
lone 2^20 proof, launches / time / HBM traffic / VALU instructions (profiles/r04_one_proof_kernels.json), the traffic record
bench.py reads for `roofline.traffic` (profiles/r04_traffic.json), and the HBM traffic of the MLE kernels of the per-kernel leg
against their algorithmic bytes (profiles/r04_mle_traffic.json).  Run from the repo root after the gpurun calls have merged
their output."""
import collections
import csv
import glob
import json
import os
import re
import shutil

ROUND = "r04"

def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    assert files, pattern
    return max(files, key=os.path.getmtime)


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("zk::", "")


def counters(path, names):
    """per dispatch (in dispatch order): (kernel, {counter: value})"""
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in names:
            key = int(r["Dispatch_Id"])
            rows.setdefault(key, (short(r["Kernel_Name"]), {}))[1][r["Counter_Name"]] = float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


def last_build(disp):
    """the dispatches of the last Merkle build in the file (tools/trace_one_proof.py proves three times)"""
    starts = [i for i, (k, _) in enumerate(disp) if k.startswith("k_runs_stage<true>")]
    i0 = starts[-1]
    out = []
    for k, c in disp[i0:]:
        out.append((k, c))
        if k.startswith("k_job_summary"):
            break
    return out


def mle_traffic():
    """HBM bytes per launch of the MLE kernels of the per-kernel leg (bench.py --kernels under --pmc FETCH_SIZE / WRITE_SIZE,
    separate passes) against their algorithmic bytes: the launches of the 43 x 2^20 shape are those whose byte count is within
    a factor of two of it (the leg also launches 1 x 2^24 shapes and 1 GiB cache-flush sweeps of k_block_sums)."""
    def per_kernel(path, name, scale):
        out = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == name:
                out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * scale)
        return out
    try:
        F = per_kernel(newest("gpurun_out/p4_kernels_FETCH/**/*counter_collection.csv"), "FETCH_SIZE", 2048)  # KB, halved on gfx950
        W = per_kernel(newest("gpurun_out/p4_kernels_WRITE/**/*counter_collection.csv"), "WRITE_SIZE", 1024)
    except AssertionError:
        return
    N = 1 << 20
    want = {"k_bind_vec<false>": ("k_bind_vec[43x2^20]", 6 * 43 * N), "k_bind_vec<true>": ("k_bind_vec_sums[43x2^20]", 6 * 43 * N),
            "k_radix_fold<true>": ("k_radix_fold[43x2^20]", 43 * (4 * N + 16 * 1024 * 8)), "k_block_sums": ("k_half_sums[43x2^20]", 4 * 43 * N)}
    res = {}
    for key, (label, alg) in want.items():
        f = sorted(x for k, v in F.items() if k.startswith(key) for x in v)
        w = sorted(x for k, v in W.items() if k.startswith(key) for x in v)
        pairs = [(a, b) for a, b in zip(f, w) if 0.5 * alg < a + b < 2.0 * alg]
        if not pairs:
            continue
        per = sum(a + b for a, b in pairs) / len(pairs)
        res[label] = {"hbm_bytes_per_launch": per, "fetch_bytes": sum(a for a, _ in pairs) / len(pairs),
                      "write_bytes": sum(b for _, b in pairs) / len(pairs), "algorithmic_bytes_per_launch": alg, "ratio": per / alg,
                      "launches_used": len(pairs)}
        print("%-28s %8.2f MB per launch vs %8.2f MB algorithmic (%.3f)" % (label, per / 1e6, alg / 1e6, per / alg))
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py --kernels "
                         "--kernel-iters 4 on round-4 code (non-temporal loads in k_block_sums / k_radix_fold); FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads), KB -> bytes x1024", "kernels": res},
              open("profiles/%s_mle_traffic.json" % ROUND, "w"), indent=1)


def main():
    mle_traffic()
    one = newest("gpurun_out/p4_one/**/*kernel_trace.csv")
    rows = sorted(csv.DictReader(open(one)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_runs_stage<true>" in r["Kernel_Name"]]
    build = []
    for r in rows[starts[-1]:]:
        build.append(r)
        if "k_job_summary" in r["Kernel_Name"]:
            break
    t0 = int(build[0]["Start_Timestamp"])
    with open("profiles/%s_one_proof_launches.csv" % ROUND, "w") as f:
        f.write("kernel,workitems,start_us,duration_us\n")
        for r in build:
            f.write("%s,%s,%.1f,%.1f\n" % (short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")),
                                         (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for r in build:
        d = dur[short(r["Kernel_Name"])]
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    F = last_build(counters(newest("gpurun_out/p4_one_FETCH/**/*counter_collection.csv"), {"FETCH_SIZE"}))
    W = last_build(counters(newest("gpurun_out/p4_one_WRITE/**/*counter_collection.csv"), {"WRITE_SIZE"}))
    V = last_build(counters(newest("gpurun_out/p4_one_VALU/**/*counter_collection.csv"), {"SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU2"}))
    per = collections.defaultdict(lambda: {"launches": 0, "us": 0.0, "fetch_bytes": 0.0, "write_bytes": 0.0, "valu_wave_instr": 0.0,
                                           "valu2_wave_instr": 0.0})
    for k, (n, us) in dur.items():
        per[k]["launches"], per[k]["us"] = n, us
    for k, c in F:  # FETCH_SIZE is in KB and, on gfx950, counts half of the bytes of wide coalesced reads (MI355X_MICROARCH.md)
        per[k]["fetch_bytes"] += c.get("FETCH_SIZE", 0.0) * 2048
    for k, c in W:
        per[k]["write_bytes"] += c.get("WRITE_SIZE", 0.0) * 1024
    for k, c in V:
        per[k]["valu_wave_instr"] += c.get("SQ_INSTS_VALU", 0.0)
        per[k]["valu2_wave_instr"] += c.get("SQ_ACTIVE_INST_VALU2", 0.0)
    # what the hashing kernels hashed in that proof: the bench trace at 2^20 (stats of the same build through the API)
    stats_path = "gpurun_out/p4_one_stats.json"
    st = json.load(open(stats_path)) if os.path.exists(stats_path) else {}
    hashes = {"k_level_hash": st.get("list_hash_perms")}
    N = 1 << 20
    alg = {  # algorithmic HBM bytes of one lone proof's launches, per kernel family
        "k_level_hash": None if not hashes["k_level_hash"] else 96.0 * hashes["k_level_hash"],  # 64 B of children in, 32 B out
        "k_runs_stage<true>": 4.0 * 33 * N,                       # the 33 run-aware columns, read once
        "k_cons_leaf_insert": 4.0 * 10 * N,                       # the group's 10 columns, read once per pass
        "k_cons_pass<true>": 4.0 * 10 * N + 4.0 * N,              # ... and once more + one slot per leaf written
        "k_cons_pass<false>": 12.0 * N,                           # levels 1..12: 8 B of children's slots in, 4 B out per node
        "k_radix_fold<true>": 43 * (4.0 * N + 16 * 1024 * 8),
    }
    out = {}
    fam = collections.defaultdict(lambda: {"launches": 0, "us": 0.0, "hbm_bytes": 0.0, "valu_wave_instr": 0.0, "valu2_wave_instr": 0.0})
    for k, v in per.items():
        f = "k_level_hash" if k.startswith("k_level_hash") else k
        fam[f]["launches"] += v["launches"]
        fam[f]["us"] += v["us"]
        fam[f]["hbm_bytes"] += v["fetch_bytes"] + v["write_bytes"]
        fam[f]["valu_wave_instr"] += v["valu_wave_instr"]
        fam[f]["valu2_wave_instr"] += v["valu2_wave_instr"]
    for f, v in sorted(fam.items(), key=lambda kv: -kv[1]["us"]):
        rec = dict(v)
        if alg.get(f):
            rec["algorithmic_bytes"] = alg[f]
            rec["traffic_ratio"] = v["hbm_bytes"] / alg[f]
        if v["valu_wave_instr"]:
            rec["valu2_share"] = v["valu2_wave_instr"] / v["valu_wave_instr"]
        if f == "k_level_hash" and hashes[f]:
            rec["hashes"] = hashes[f]
            rec["hbm_bytes_per_hash"] = v["hbm_bytes"] / hashes[f]
            rec["valu_lane_instr_per_hash"] = v["valu_wave_instr"] * 64 / hashes[f]
            rec["gperm_per_s"] = hashes[f] / 1e9 / (v["us"] / 1e6)
        out[f] = rec
    src = ("rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 (separate passes) "
           "-- python3 tools/trace_one_proof.py: the launches of ONE lone 2^20 proof of the bench trace; FETCH_SIZE doubled per "
           "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads), KB -> bytes x1024")
    json.dump({"source": src, "kernels": out}, open("profiles/%s_one_proof_kernels.json" % ROUND, "w"), indent=1)
    if "k_level_hash" in out and "hbm_bytes_per_hash" in out["k_level_hash"]:
        r = out["k_level_hash"]
        json.dump({"source": "profiles/%s_one_proof_kernels.json" % ROUND,
                   "k_level_hash": {"hbm_bytes_per_hash": r["hbm_bytes_per_hash"], "algorithmic_bytes_per_hash": 96.0,
                                    "valu_lane_instr_per_hash": r.get("valu_lane_instr_per_hash"), "valu2_share": r.get("valu2_share"),
                                    "hashes_in_the_profiled_proof": r["hashes"]}},
                  open("profiles/%s_traffic.json" % ROUND, "w"), indent=1)
    for f, r in out.items():
        print("%-26s launches %3d  %8.1f us  %9.2f MB HBM%s%s" % (
            f, r["launches"], r["us"], r["hbm_bytes"] / 1e6,
            ("  (%.2f x algorithmic)" % r["traffic_ratio"]) if "traffic_ratio" in r else "",
            ("  valu2 %.2f" % r["valu2_share"]) if "valu2_share" in r else ""))
    shutil.copy(newest("gpurun_out/p4_one_FETCH/**/*counter_collection.csv"), "profiles/%s_pmc_fetch_size.csv" % ROUND)
    shutil.copy(newest("gpurun_out/p4_one_WRITE/**/*counter_collection.csv"), "profiles/%s_pmc_write_size.csv" % ROUND)
    shutil.copy(newest("gpurun_out/p4_one_VALU/**/*counter_collection.csv"), "profiles/%s_pmc_valu.csv" % ROUND)
    for tag in ("kernels", "b1", "bdef", "lasso", "sumcheck"):
        shutil.copy(newest("gpurun_out/p4_%s/**/*kernel_stats.csv" % tag), "profiles/%s_%s_kernel_stats.csv" % (ROUND, tag))
    for src_, dst in (("p4_kernels.json", "kernels.json"), ("p4_kernels_plain.json", "kernels_unprofiled.json"),
                      ("p4_bench.json", "final_bench.json"), ("p4_bench_b1.json", "final_bench_b1.json"),
                      ("p4_bench_ctx_per_lane.json", "final_bench_context_per_lane.json"),
                      ("p4_bdef.json", "bdef_bench.json"), ("p4_bdef_union.json", "bdef_union.json"),
                      ("p4_bdef_intervals.csv.gz", "bdef_intervals.csv.gz"), ("p4_stress_soak.log", "stress_soak.log"),
                      ("p4_event_semantics.txt", "event_semantics.txt"), ("p4_event_semantics_profiled.txt", "event_semantics_profiled.txt"),
                      ("p4_bench_s0.json", "final_bench_own_thread_transcripts.json"), ("p4_gpu_bound.txt", "gpu_bound_rate.txt"),
                      ("p4_lasso.json", "lasso.json"), ("p4_sumcheck.json", "sumcheck.json"), ("p4_extra.json", "extra.json"),
                      ("p4_configs.jsonl", "configs.jsonl"), ("p4_bdef_concurrency.json", "bdef_concurrency.json"),
                      ("p4_stream_concurrency.txt", "stream_concurrency.txt"), ("p4_prio_probe.txt", "prio_probe.txt"),
                      ("p4_icache_corun.txt", "icache_corun.txt"), ("p4_bench_upload16.json", "bench_upload_16_byte_records.json"),
                      ("p4_bench_upload32.json", "bench_upload_32_byte_records.json"),
                      ("p4_bench_batch4.json", "bench_2p20_jobs_of_up_to_4.json")):
        if os.path.exists(os.path.join("gpurun_out", src_)):
            shutil.copy(os.path.join("gpurun_out", src_), "profiles/%s_%s" % (ROUND, dst))
    if os.path.exists("profiles/%s_bdef_union.json" % ROUND) and os.path.exists("profiles/%s_bdef_bench.json" % ROUND):
        u = json.load(open("profiles/%s_bdef_union.json" % ROUND))["classes"]["level_hash"]
        b = json.loads(open("profiles/%s_bdef_bench.json" % ROUND).read().strip().splitlines()[-1])["roofline"]
        print("bdef: roofline.frac of the line (events, its roofline leg) %.3f; re-derived from the profiler's trace of the same run %.3f "
              "(ratio %.3f); avg launch %.1f us (events) / %.1f us (trace)" % (b["frac"], u["frac"], b["frac"] / u["frac"],
                                                                            b["avg_launch_us"], u["avg_launch_us"]))
    for name in ("final_bench", "final_bench_b1", "final_bench_context_per_lane", "final_bench_own_thread_transcripts"):
        pth = "profiles/%s_%s.json" % (ROUND, name)
        if not os.path.exists(pth):
            continue
        d = json.loads(open(pth).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-36s %7.1f M steps/s  %.3f ms/proof  %s frac %.3f  in-proof %s  bind %s" %
              (name, d["value"] / 1e6, d["config"]["ms_per_proof_per_gpu"], r["kernel"], r["frac"], r.get("in_proof_frac"),
               r.get("bind_hbm_frac")))


if __name__ == "__main__":
    main()
