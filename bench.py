#!/usr/bin/env python3
"""bench.py -- trace steps/sec proved on the synthetic RV64I ADD/XOR loop at a 2^20 trace (BASELINE config 3).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--nv 20] [--mode traces|shard]
                    [--no-cpu-baseline] [--kernels]

A "step" is ONE full pass of the prover hot path over one batch of `--batch` independent 2^20-step traces (default: 8.8 per
sponge-server thread -- 8 slots each and a tenth more proofs than slots --, 10 servers on a 16-CPU share of the host;
every trace gets its own complete proof) whose 43 witness
columns each are already resident in HBM when the timed region starts.  Per trace: the exact Fiat-Shamir schedule of
Prover.prove (public inputs, SUMCHECK_BEGIN, one LASSO_TABLE absorption per lookup step, POLY_COMMITMENTS, 43*v challenges,
OPENING_CLAIMS), 43 SHA3 Merkle commits, 43 MLE evaluations, 43 openings, packagePublicIO and the ZIGZ v1 serialisation of
the proof.  Nothing is skipped or cached between steps.

Why a batch: one proof is bounded by its own sequential SHA3 transcript on ONE host core (19 bytes absorbed per lookup step,
~27 ms at 2^20) while its GPU work takes ~0.55 ms, so a proving service keeps the GPU busy by running many proofs per GPU
concurrently (one host thread + one HIP stream each), and advances their transcripts 8 per host thread in lock step
(zigz_host_sponge_servers: 8-way AVX-512 Keccak-f, same bytes absorbed) so that the host cores carry them.

The JSON line: metric / value / ... as the contract says; `roofline` = the kernel class that dominates the TIMED REGION
(k_level_hash: permutations x VALU instructions per hash / the sum of its launches' own durations / the int-VALU issue peak;
flat keys, the named ones first); `config` = the workload plus, measured right after the timed region in the same process,
single_proof_ms, pcie_inclusive_value (trace upload + witness kernels inside the loop), the same batch on three other traces
(register_worst_case_value, config4_mixed_value, straight_line_value), under the dense Merkle build (dense_merkle_value) and at
the other trace sizes of the north-star (value_nv16 / _nv22 / _nv24); `cpu_baseline`; nested records under `detail`
(merkle_variants, self_check: SHA-256 of every lane's proof identical under every Merkle build and transcript path; host
phases; per-class kernel time).  DESIGN.md s6 says what each number is.

N > 1: `python bench.py --gpus N` starts the N ranks itself (child `python -m torch.distributed.run`, before this process
touches torch or HIP) and relays rank 0's JSON line; when a launcher has already set WORLD_SIZE the process is a rank.  One
rank per GPU, backend nccl (= RCCL), each rank pinned to the cores of its GPU's NUMA node; the path shards by independent
traces (one batch of proofs per rank, no data-path collective): "scaling": "weak", value = all ranks' trace steps /
max-over-ranks time.  `--mode shard` (one proof per step, its 43 columns sharded over the ranks) is the strong-scaling variant.

`--kernels` runs only the per-kernel leg (cold-HBM launches of the MLE and Keccak kernels with kernel timestamps) and prints
its own JSON line; that is the command profiles/r03_kernels_* were taken from.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
# int-VALU issue peak the Keccak kernels are priced against: 256 CUs x 4 SIMDs x 32 lane-instructions per clock
# (both VALU pipes) x 2.4 GHz.  Only v_bitop3 (and a few other simple ops) can use the second pipe; rotates cannot,
# so the ceiling for the Keccak instruction mix is ~0.65 of this.
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12


def isa_counts():
    """VALU instructions per hash of the Keccak kernels, counted from the gfx950 code (tools/isa_counts.py)."""
    with open(os.path.join(ROOT, "profiles", "isa_counts.json")) as f:
        k = json.load(f)["kernels"]
    return {"leaves": k["k_keccak_leaves"]["valu"], "level": k["k_keccak_level<4>"]["valu"],
            "level_hash": k["k_level_hash<false, true>"]["valu"], "level_hash_leaf": k["k_level_hash<true, true>"]["valu"],
            "top": k["k_merkle_top"]["valu"]}


# ------------------------------------------------------------------ N > 1: self-launch (parent never touches the GPU)
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


MERKLE_BUILDS = {  # --merkle: what config.merkle_build says
    "cons": "as struct, and the ten columns that are functions of the instruction at pc (pc, x0, opcode, rd, rs1, rs2, funct3, "
            "funct7, imm, is_read) as a content-addressed group: per level a device hash table finds the first node with the same "
            "content in all ten, only those representatives are hashed (a loop of P steps has at most P distinct nodes per "
            "level); probed on the leaves first and dropped for a trace that does not repeat; identical trees",
    "struct": "leaf + level-1 digests of the 8 structurally small-domain columns (x0, opcode, rd, rs1, rs2, funct3, funct7, "
              "is_read: values < 128 by construction) from constant tables; large levels of the 31 register columns x1..x31 "
              "(at most one of them changes per step, whatever the program) and of mem.address / mem.value (0 on every step "
              "that is not a LOAD / STORE) run-aware: a node that repeats its left neighbour is neither hashed nor written, "
              "decided from the values on the device; pc and imm dense; identical trees",
    "regs": "as struct, but only the 31 register columns run-aware (mem.address / mem.value dense)",
    "all": "as struct, with every column that is not small-domain run-aware (pc and imm too)",
    "tables": "dense; leaf + level-1 digests of the 8 structurally small-domain columns from constant tables, identical trees",
    "dense": "dense: every node of every column hashed",
}


def cgroup_cpu_quota():
    """CPUs the cgroup grants this process tree (cpu.max), or None."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(quota) // int(period))
    except (OSError, ValueError):
        pass
    return None


def host_cpus():
    """CPUs this process may use: the cgroup quota (cpu.max) if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    q = cgroup_cpu_quota()
    return min(n, q) if q else n


def rank_cpus(world, pinned):
    """CPUs ONE rank may count on.  A rank that has pinned itself (zigz_amd/placement.py) already holds its slice of the
    cores in its affinity mask -- dividing that by the world size again would leave an 8-GPU run two sponge servers per GPU;
    an unpinned rank takes its share of the common mask.  A cgroup quota is shared by all ranks of the launch."""
    n = len(os.sched_getaffinity(0))
    if not pinned:
        n //= max(world, 1)
    q = cgroup_cpu_quota()
    if q:
        n = min(n, q // max(world, 1))
    return max(1, n)


def has_avx512f():
    try:
        return " avx512f" in open("/proc/cpuinfo").read()
    except OSError:
        return False


def default_batch(ncpu):
    """Lanes per GPU: every lane keeps one host core busy with its proof's sequential SHA3 sponge; two cores per rank are
    left for the helper threads (serialisation, Lasso commitments) and this interpreter (used without the sponge service)."""
    return max(1, min(14, ncpu - 2))


def launch_ranks(n, argv):
    """Start n ranks as a child torch.distributed.run and relay rank 0's JSON line.  Nothing in this process has
    imported torch or loaded HIP at this point (tests/test_bench_launch.py asserts it)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["ZIGZ_BENCH_LAUNCHED"] = "1"
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        t = ln.strip()
        if t.startswith("{") and t.endswith("}"):
            line = t
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    if rc == 0 and not line:
        sys.stderr.write("bench.py: the ranks exited without printing a result line\n")
        rc = 1
    return rc


# ------------------------------------------------------------------ CPU baseline (child process; the only oracle user)
def cpu_baseline_child(nv, sample_cols):
    """Runs in a process of its own, BEFORE the parent initialises the GPU: the reference algorithm on one host core
    (oracle = literal C restatement, -O2, 1 thread) on a bounded sample of the same workload, every part timed for real:
    `sample_cols` of the 43 columns through commit + eval + open exactly as prover.zig:405-431 does it (2N-1 hashes,
    two naive O(v 2^v) evals, recompute-on-open), scaled x43/sample_cols, plus the complete sequential transcript of
    steps [4/6]-[5/6] (prover.zig:229-363, one absorption pair per lookup step) run once in full."""
    import ctypes as C
    import oracle_lib as O
    import programs
    P = O.P_BB
    N = 1 << nv
    prog = programs.add_xor_loop((N - 3) // 4)
    cols, nv_o, ns = O.witness_from_program(P, prog, 0x1000, None, 2 * N)
    assert nv_o == nv
    num_lookups = O.vm_trace(prog, 0x1000, None, 2 * N)["num_lookups"]
    pts = O.splitmix64_field(99, sample_cols * nv).reshape(sample_cols, nv)
    t0 = time.perf_counter()
    for c in range(sample_cols):
        O.commit_column_literal(P, cols[(c * 43) // sample_cols], pts[c])
    t_cols = time.perf_counter() - t0
    f = O.lib.orc_prove_transcript_only
    f.restype = C.c_uint64
    f.argtypes = [C.c_uint64, C.c_char_p, C.c_uint64, C.c_size_t, C.c_size_t, C.c_size_t]
    t0 = time.perf_counter()
    f(P, bytes(32), 0x1000, ns, nv, num_lookups)
    t_tr = time.perf_counter() - t0
    total = t_cols * (43.0 / sample_cols) + t_tr
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return ({
        "value": ns / total, "unit": "trace steps/s", "cores": 1, "kind": "port",
        "sample": "%d/43 columns at 2^%d: literal commit + 2 naive evals + open, x43/%d; + full transcript; oracle, gcc -O2, 1 thread"
                  % (sample_cols, nv, sample_cols),
        "sample_detail": "%d of 43 columns at 2^%d through literal commit + 2 naive evals + recompute-on-open (%.1f s measured), "
                         "scaled x43/%d; + the full sequential transcript of steps 4-5 (%d lookup absorptions) run for real (%.3f s "
                         "measured); oracle/zigz_oracle.c, gcc -O2, 1 thread" % (sample_cols, nv, t_cols, sample_cols, num_lookups, t_tr),
        "seconds_per_proof": total, "transcript_seconds": t_tr, "columns_seconds_scaled": t_cols * 43.0 / sample_cols,
        "host_cpu": model, "host_nproc": os.cpu_count(), "threads_used": 1})


def run_cpu_baseline(nv, sample_cols):
    """In a child process (this process then never loads the oracle).  Where a child cannot be started -- under rocprofv3
    the profiler has attached to this process and the box refuses to start other programs from it -- the same function
    runs here instead, still before this process does any GPU work of its own."""
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--nv", str(nv),
                              "--cpu-sample-cols", str(sample_cols)], stdout=subprocess.PIPE, text=True, check=True)
        return json.loads(out.stdout.strip().split("\n")[-1])
    except Exception as e:
        sys.stderr.write("bench.py: cpu_baseline child failed (%r); running it in this process\n" % (e,))
        r = cpu_baseline_child(nv, sample_cols)
        r["sample_detail"] += " (run inside the bench process: no child could be started)"
        return r


# ------------------------------------------------------------------ per-kernel leg
def kernel_leg(ctx, nv, ncols, iters, cold=True, big_nv=24):
    """Cold-HBM launches of the hot kernels on synthetic resident tables, kernel timestamps (zigz_bench_kernel)."""
    ic = isa_counts()
    res = {}

    def hbm(name, knv, kcols):
        r = ctx.bench_kernel(name, knv, kcols, iters, cold)
        gbs = r["algorithmic_bytes"] / 1e9 / (r["avg_us"] / 1e6)
        res["%s[%dx2^%d]" % (name, kcols, knv)] = {
            "bound": "hbm", "avg_us": r["avg_us"], "min_us": r["min_us"], "max_us": r["max_us"], "launches": r["launches"],
            "algorithmic_bytes": r["algorithmic_bytes"], "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS,
            "best_frac": r["algorithmic_bytes"] / 1e9 / (r["min_us"] / 1e6) / HBM_PEAK_GBS}
        return res["%s[%dx2^%d]" % (name, kcols, knv)]

    def valu(name, knv, kcols, instr):
        r = ctx.bench_kernel(name, knv, kcols, iters, cold)
        gperm = r["units"] / 1e9 / (r["avg_us"] / 1e6)
        res["%s[%dx2^%d]" % (name, kcols, knv)] = {
            "bound": "valu", "avg_us": r["avg_us"], "min_us": r["min_us"], "max_us": r["max_us"], "launches": r["launches"],
            "permutations": r["units"], "gperm_per_s": gperm, "valu_instr_per_hash": instr,
            "achieved_Tinstr_s": gperm * instr / 1e3, "frac": gperm * instr / 1e3 / VALU_PEAK_TOPS,
            "hbm_frac": r["algorithmic_bytes"] / 1e9 / (r["avg_us"] / 1e6) / HBM_PEAK_GBS}
        return res["%s[%dx2^%d]" % (name, kcols, knv)]

    hbm("k_bind_vec", nv, ncols)        # partialEval of all columns at once: the batched bind of eval-by-folds
    hbm("k_bind_vec_sums", nv, ncols)   # fused with the next round's half sums (sumcheck_core)
    hbm("k_half_sums", nv, ncols)   # roundPolynomial / sumOverHypercube
    hbm("k_radix_fold", nv, ncols)      # eval: top v-10 variables in one pass
    if big_nv:
        hbm("k_bind_vec", big_nv, 1)        # one 2^24 table (config 5 row count)
        hbm("k_bind_vec_sums", big_nv, 1)
        hbm("k_half_sums", big_nv, 1)
        hbm("k_block_sums", big_nv, 1)      # radix sumcheck pass 1
    valu("k_keccak_leaves", nv, ncols, ic["leaves"])
    valu("k_keccak_level", nv, ncols, ic["level"])
    return res


def lasso_leg(ctx, reps=5):
    """LassoProver.prove (src/lookups/lasso_prover.zig:103-173) end to end through zigz_lasso_prove: the reference's 8-bit
    XOR table (2^16 rows x 3 fields) and Q random valid queries, host buffers in, proof out (uploads included), plus the
    fingerprint kernel alone and the sumcheck of the same size on a resident table."""
    import numpy as np
    a = np.arange(256, dtype=np.uint64)
    tab = np.stack([np.repeat(a, 256), np.tile(a, 256), np.repeat(a, 256) ^ np.tile(a, 256)], axis=1)
    rng = np.random.default_rng(7)
    out = {"table_rows": len(tab), "sizes": {}}
    for lq in (16, 18, 20):
        q = tab[rng.integers(0, len(tab), size=1 << lq)]
        ctx.lasso_prove(tab, q)  # warm-up: workspaces
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.lasso_prove(tab, q)
        ms = (time.perf_counter() - t0) / reps * 1e3
        d = ctx.dev_alloc((1 << lq) * 4)
        ctx.upload(np.arange(1 << lq, dtype=np.uint64) % 2013265921, d)
        ctx.dev_sumcheck_prove(d, 1 << lq)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.dev_sumcheck_prove(d, 1 << lq)
        ms_sc = (time.perf_counter() - t0) / reps * 1e3
        ctx.dev_free(d)
        sponge_bytes = ((1 << lq) + len(tab)) * 8
        out["sizes"]["2^%d" % lq] = {"lasso_prove_ms": ms, "queries_per_s": (1 << lq) / (ms / 1e3),
                                     "sumcheck_resident_ms": ms_sc, "flat_commit_bytes": sponge_bytes}
    k = ctx.bench_kernel("k_lasso_fingerprints", 20, 1, 10, True)
    out["k_lasso_fingerprints[2^20 rows x 3]"] = {"avg_us": k["avg_us"], "min_us": k["min_us"],
                                                  "rows_per_s": k["units"] / (k["avg_us"] / 1e6),
                                                  "hbm_frac": k["algorithmic_bytes"] / 1e9 / (k["avg_us"] / 1e6) / HBM_PEAK_GBS}
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)  # (1.1 s of timed region: the lanes start in lock step, and the first cycles of a
    # region are its slowest -- 10 steps read 3-4 % under 20, 20 within 1 % of 40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="independent traces proven concurrently per GPU per step (0 = pick: 8.8 per "
                    "sponge server, bounded by free HBM; without the service this rank's share of the host CPUs minus 2, at most "
                    "14).  One proof alone is bound by its sequential host transcript (~27 ms on one core) against ~0.55 ms of GPU "
                    "work with the default Merkle build, so a proving service keeps many proofs in flight, one host thread + one "
                    "HIP stream each: 10 sponge servers x 8 lanes ~1.6-1.8 G steps/s on a 16-CPU share; with one core per "
                    "transcript 14 lanes fit (~0.5 G)")
    ap.add_argument("--sponge-servers", type=int, default=-1, help="host threads of the sponge service (zigz_host_sponge_servers): each "
                    "advances up to 8 proofs' transcripts in lock step with one 8-way AVX-512 permutation per block while the "
                    "proofs' own threads sleep.  0 = every proof absorbs its transcript on its own thread (the lane count is then "
                    "bounded by the host cores: 14 lanes, ~480 M steps/s); -1 (default) = 5/8 of this rank's CPUs, at most 12 "
                    "(10 on a 16-CPU share: 80 proofs in flight, ~50 GiB of HBM), or 0 without AVX-512F")
    ap.add_argument("--mode", choices=["traces", "shard"], default="traces",
                    help="traces (default, the headline): every GPU proves its own traces, no data-path collective, weak "
                    "scaling = independent-trace throughput.  shard: ONE proof per step, its 43 columns sharded over the "
                    "GPUs (two all-gathers of 43 x 32 B and 43 x (24 + 33 v) B per proof), strong scaling; bounded by the "
                    "sequential host transcript that every rank replays (DESIGN.md s7)")
    ap.add_argument("--exchange", choices=["shm", "rccl"], default="shm",
                    help="transport of the sharded paths (--mode shard, and the one-proof-over-all-ranks leg when N > 1): shm = the "
                    "shared-memory mailbox (zigz_shm_comm: 2-5 us per exchange between the ranks of one node); rccl = RCCL bound "
                    "natively (zigz_rccl_comm, no torch in the loop: ~60 us per exchange of <= 8 KiB, for ranks on several nodes).  "
                    "rccl needs one GPU per rank")
    ap.add_argument("--slots", type=int, default=-1, help="GPU slots: contexts (HIP stream + tree / list workspaces) the lanes share; a "
                    "lane holds one only for its proof's GPU phases (begin -> roots -> challenges -> open_all -> end), so proofs in "
                    "flight are not bounded by HBM for workspaces.  -1 (default): lanes / 5 + 2, between 4 and 16; 0: a context per "
                    "lane (round 3's form: the build overlaps the proof's own transcript, and every proof in flight holds its "
                    "workspaces throughout)")
    ap.add_argument("--nv", type=int, default=20, help="log2 of the padded trace length (BASELINE config 3: 20)")
    ap.add_argument("--upload", action="store_true", help="A/B only (never the headline: the metric's inputs are resident): the timed "
                    "region uploads every proof's compact trace and builds its witness inside the proof's GPU slot, as the "
                    "PCIe-inclusive leg of the default run does")
    ap.add_argument("--trace", choices=["bench", "mixed", "worst", "straight"], default="bench",
                    help="the program whose traces the lanes prove: bench (default, BASELINE config 3: the RV64I ADD/XOR loop); mixed "
                    "(config 4's RV64IM mix); worst (31 registers written in turn); straight (a program that never loops).  The "
                    "default run measures the other three in legs of their own; this switch makes one of them THE workload "
                    "(profiling, A/B)")
    ap.add_argument("--merkle", choices=["cons", "struct", "regs", "all", "tables", "dense"], default="cons",
                    help="Merkle build of the 43 columns (identical trees and proofs in every mode).  cons (the product's default): "
                    "struct + the ten instruction-determined columns as a content-addressed group (repeat wherever the program "
                    "loops).  struct: leaf + level-1 digests of the 8 structurally small-domain columns from constant tables, and the "
                    "columns that are piecewise constant by construction -- the 31 registers x1..x31 (at most one of them changes "
                    "per step) and mem.address / mem.value (0 on every step without a memory access) -- with run-aware large "
                    "levels (a node that repeats its left neighbour is copied, not hashed).  regs: the registers only.  all: "
                    "every column that is not small-domain.  tables: the tables only (round 2's first default).  dense: hash "
                    "every node")
    ap.add_argument("--dedup", action="store_true", help="same as --merkle all")
    ap.add_argument("--dense-merkle", action="store_true", help="same as --merkle dense")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cols", type=int, default=12, help="columns of the CPU baseline sample (~1.1 s each)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--kernels", action="store_true", help="per-kernel leg only (cold-HBM launches, kernel timestamps)")
    ap.add_argument("--lasso", action="store_true", help="Lasso leg only: zigz_lasso_prove end to end (8-bit XOR table, 2^16 rows) "
                    "with 2^16 .. 2^20 queries + the fingerprint kernel, prints its own JSON line")
    ap.add_argument("--kernel-iters", type=int, default=10)
    ap.add_argument("--no-extras", action="store_true", help="skip the single-proof, PCIe-inclusive and per-kernel legs")
    ap.add_argument("--dry-run", action="store_true", help="launcher rehearsal without a GPU: the ranks rendezvous over gloo, "
                    "all-reduce a token and rank 0 prints {dry_run, n_gpus}; nothing is measured")
    args = ap.parse_args()

    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline_child(args.nv, args.cpu_sample_cols)))
        return 0

    # ---- N > 1 without a launcher: become the launcher.  No torch / HIP import has happened in this process.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        assert "torch" not in sys.modules and "zigz_amd" not in sys.modules
        return launch_ranks(args.gpus, sys.argv[1:])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE); refusing to report a wrong "
                         "n_gpus\n" % (args.gpus, world))
        return 2

    # ---- rank placement: pin this process to the cores of its GPU's NUMA node BEFORE anything initialises HIP (the runtime's
    # helper threads, the lanes' threads and the sponge servers all inherit the mask).  sysfs only; zigz_amd/placement.py is
    # loaded as a plain file so that not even the package's libraries are open yet.
    import importlib.util
    _ps = importlib.util.spec_from_file_location("zigz_placement", os.path.join(ROOT, "zigz_amd", "placement.py"))
    placement = importlib.util.module_from_spec(_ps)
    _ps.loader.exec_module(placement)
    pin = {"pinned": 0, "numa_node": -1, "why": "ZIGZ_BENCH_NO_PIN"} if os.environ.get("ZIGZ_BENCH_NO_PIN") else \
        placement.pin_rank(local_rank, local_world=int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))

    if args.dry_run:
        tok = 1.0
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import torch
            import torch.distributed as dist
            dist.init_process_group(backend="gloo")
            t = torch.tensor([1.0], dtype=torch.float64)
            dist.all_reduce(t)
            tok = float(t.item())
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": tok,
                              "launched_by_bench": os.environ.get("ZIGZ_BENCH_LAUNCHED") == "1"}), flush=True)
        return 0

    # Only the result line may reach stdout: libraries print banners there (RCCL writes its version block to stdout when the
    # first communicator is created), so fd 1 is pointed at stderr for the whole run and the line goes to the saved fd.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(result_fd, (line + "\n").encode())

    # ---- CPU baseline first (rank 0, N = 1 only), in a child of its own: the GPU phase that follows is then one
    # contiguous stretch, and this process never loads the oracle
    cpu = None
    if world == 1 and not args.no_cpu_baseline and args.cpu_sample_cols > 0 and not args.kernels:
        cpu = run_cpu_baseline(args.nv, args.cpu_sample_cols)

    dist = None
    torch = None
    gpu_share = 1  # ranks of this launch that share this rank's GPU (a rehearsal of N ranks on a 1-GPU box): HBM budgets are divided by it
    backend = os.environ.get("ZIGZ_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N > 1 path on a 1-GPU box
    if world > 1 or os.environ.get("ZIGZ_BENCH_FORCE_DIST") == "1":  # (the env switch: the RCCL code path with one rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()  # does not initialise HIP
        if backend == "nccl" and ndev < world:
            backend = "gloo"  # fewer GPUs than ranks (rehearsal on a 1-GPU box): RCCL cannot put two ranks on one device
        local_rank = local_rank % max(ndev, 1)
        gpu_share = max(1, (world + max(ndev, 1) - 1) // max(ndev, 1))
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    tdev = "cuda" if backend == "nccl" else "cpu"

    if args.dense_merkle:
        args.merkle = "dense"
    if args.dedup:
        args.merkle = "all"

    def set_merkle_mode(mode):  # Prover reads these when a proof starts (host/prover.cpp)
        os.environ.pop("ZIGZ_DENSE_MERKLE", None)
        os.environ.pop("ZIGZ_RUN_AWARE", None)
        if mode == "dense":
            os.environ["ZIGZ_DENSE_MERKLE"] = "1"
        else:
            os.environ["ZIGZ_RUN_AWARE"] = {"cons": "cons", "struct": "struct", "regs": "regs", "all": "all", "tables": "off"}[mode]
    set_merkle_mode(args.merkle)
    import zigz_amd
    from zigz_amd import host
    import programs

    from concurrent.futures import ThreadPoolExecutor

    nv = args.nv
    N = 1 << nv

    if args.kernels:
        ctx = zigz_amd.Context(local_rank)
        res = kernel_leg(ctx, nv, 43, args.kernel_iters)
        ctx.close()
        emit(json.dumps({"kernels": res, "cold": True, "iters": args.kernel_iters,
                         "hbm_peak_GBs": HBM_PEAK_GBS, "valu_peak_Tinstr_s": VALU_PEAK_TOPS}))
        return 0

    if args.lasso:
        emit(json.dumps(lasso_leg(zigz_amd.Context(local_rank))))
        return 0

    shard = args.mode == "shard"
    ncpu = rank_cpus(world, pin.get("pinned", 0))  # what this rank sizes its threads for
    servers = args.sponge_servers
    if servers < 0:
        # a server is one busy core per 8 proofs in flight; the lanes' own threads, the serialiser helpers and the runtime's
        # event thread need ~2.2 ms of CPU per proof on top.  On a 16-CPU share (round 4, lanes on shared GPU slots, one box,
        # three interleaved runs each): 10 servers 2.03-2.06 G steps/s at 13.1 CPUs busy, the GPU slots mostly free (host-bound);
        # 11: 2.06-2.11 G at 14.4, proofs queue for the GPU (GPU-bound); 12: 2.07-2.11 G at 15.0-15.6; 13: 1.98-2.10 G, over the quota
        servers = 0 if shard or not has_avx512f() else max(1, min(12, ncpu * 11 // 16))
    if servers > 0:
        zigz_amd._ffi.lib.zigz_host_sponge_servers(servers)
        servers = servers if zigz_amd._ffi.lib.zigz_host_sponge_batching() else 0
    # 8 transcripts per server are in lock step; a tenth more proofs than sponge slots keeps the slots full while a proof is in
    # its GPU phases (roots, challenges, openings: 2-4 of its 40-odd ms): 10 x 8 slots, 88 lanes -- 2.00-2.03 G steps/s against
    # 1.95-1.98 G with 80 on one box, 14.0 of 16 CPUs busy (96 / 104 lanes: 2.02 / 2.05 G at 15.0 / 15.7 CPUs)
    use_slots_ = not shard and args.slots != 0
    # lanes: with slots a lane's GPU phases are no longer underneath its own transcript, so three proofs for ten sponge slots on
    # top keep the slots full (104 lanes on 80 slots; 93: -1 % resident, -12 % with the trace uploads inside the slots; a tenth
    # more than slots with a context per lane)
    B = 1 if shard else (args.batch if args.batch > 0 else
                         (8 * servers + max(1, 8 * servers * (3 if use_slots_ else 1) // 10) if servers else default_batch(ncpu)))

    def lane_bytes(nv_l):   # what a lane holds between its proofs: the 43 resident columns
        return 43 * 4 * (1 << nv_l) + (32 << 20)

    def slot_bytes(nv_l, worst=True):  # what a GPU slot's workspaces grow to: lists + digests in list order (0.45 GiB at 2^20 on the
        # bench trace, measured), slabs for a dropped group and longer lists on a trace that never loops (2.2 GiB)
        return int((2.4 if worst else 0.47) * (1 << 30) * (1 << max(nv_l - 20, 0)))
    if use_slots_ and args.batch <= 0 and args.nv <= 17:
        # small traces: a proof spends more of its cycle in its (shared) GPU phases than in its 2.4 ms sponge: three lanes for two
        # sponge slots (2^16, 16 proofs per job: 93 lanes 1.10 G, 128 lanes 1.26 G steps/s at 15.6 of 16 CPUs busy)
        B = B * 11 // 8
    hbm_free = None
    if not shard and args.batch <= 0:
        probe = zigz_amd.Context(local_rank)
        hbm_free = probe.mem_info()[0] // gpu_share
        probe.close()
        if use_slots_:
            k_guess = args.slots if args.slots > 0 else (max(4, min(12, B // 8 + 1)) if args.nv < 24 else 6)
            # (the legs on traces that do not loop run at 2^20 and below; above, the budget is the bench trace's)
            B = max(1, min(B, (int(hbm_free * 0.92) - k_guess * slot_bytes(args.nv, worst=args.nv < 22)) // lane_bytes(args.nv)))
        else:
            # (a context per lane: ~0.6 GiB at 2^20 on the bench trace with the digests in list order; a trace that does not loop
            # needs ~2.3 GiB -- slabs for the dropped group, longer lists -- and the legs below run such traces on the same lanes)
            B = max(1, min(B, int(hbm_free * 0.9) // (slot_bytes(args.nv) + lane_bytes(args.nv))))
    blocking = B + servers + 2 > ncpu  # more threads than cores: wait for the GPU asleep, not spinning
    if os.environ.get("ZIGZ_BENCH_BLOCKING_SYNC"):
        blocking = os.environ["ZIGZ_BENCH_BLOCKING_SYNC"] == "1"
    if blocking:
        rc = zigz_amd._ffi.lib.zigz_device_set_blocking_sync(local_rank, 1)
        if rc != 0:
            sys.stderr.write("bench.py: zigz_device_set_blocking_sync -> %d (threads will spin while waiting)\n" % rc)
            blocking = False

    # ---- GPU slots (zigz_amd/csrc/host/zigz_host.hpp: GpuSlots).  A proof needs the GPU for a few milliseconds and a sponge
    # slot for tens: the lanes therefore share K contexts (stream + tree / list workspaces each) and a lane holds one only
    # between the end of its transcript's step 5 and the end of its openings; between proofs a lane holds its resident
    # witness and nothing else.  Proofs in flight are then bounded by host cores and witness bytes, not by HBM for
    # workspaces (21 lanes at 2^24 with a context per lane).  --slots 0: a context per lane, as in round 3.
    use_slots = use_slots_
    slots = None
    setup_ctx = zigz_amd.Context(local_rank)  # allocations, witness uploads (set-up; its workspaces are given back after it)

    def make_slots(k, nv_l):
        sl = host.Slots(local_rank, k)
        if nv_l <= int(os.environ.get("ZIGZ_BENCH_BATCH_NV", "17")) and os.environ.get("ZIGZ_BENCH_NO_BATCHING") != "1":
            # small traces: proofs that reach their GPU phase together share one commit job (GpuBatcher): at 2^16 a proof's ~35
            # launches are 13 us of work each; the lanes of a sponge server leave their transcripts together anyway
            sl.set_batching(int(os.environ.get("ZIGZ_BENCH_BATCH_MAX", "16")), float(os.environ.get("ZIGZ_BENCH_BATCH_LINGER_US", "200")),
                            int(os.environ.get("ZIGZ_BENCH_BATCH_NV", "17")))
        return sl

    def slot_count(nl, nv_l):
        """Slots for nl lanes at 2^nv_l: a lane spends ~1/10 of its cycle in a slot (2-4 of ~42 ms at 2^20; the ratio holds
        at the other sizes, GPU phase and sponge both scale with the trace); twice that keeps queueing short, and a slot is
        cheap while it is not needed (the most recently released one is handed out first: unused ones never grow)."""
        if args.slots > 0:
            return args.slots
        # (A/B at 2^20, 93 lanes: 8 slots 1.94 G, 12 1.96 G, 16 1.89-1.99 G, 24 1.81 G, 32 1.75 G -- more builds in flight get in
        # each other's way; the lanes of one sponge server leave their transcripts together, so fewer than 8 makes them queue)
        if nv_l <= 17:  # small traces share commit jobs, 16 proofs each (make_slots): 8 slots carry 128 proofs at once
            return 8
        # (2^24, 8 steps: 8 slots / 75 lanes 1.75 G, 7 / 77 1.80 G, 6 / 80 1.83 G -- a slot's 7 GiB of trees and lists are worth 2.5 lanes)
        return max(4, min(12, nl // 8 + 1)) if nv_l < 24 else 6

    _straight = {}

    def main_program(n_rows, i):
        if args.trace == "mixed":
            return programs.mixed_loop((n_rows - 8) // 12 - i)
        if args.trace == "worst":
            return programs.register_round_robin((n_rows - 2) // 31 - i)
        if args.trace == "straight":
            if n_rows not in _straight:
                _straight[n_rows] = programs.straight_line_program(1000 + rank, int(0.95 * n_rows))
            return _straight[n_rows][:4 * (int(0.95 * n_rows) - i % 1000)]
        return programs.add_xor_loop((n_rows - 3) // 4 - i)

    class Lane:  # one trace + its resident witness; proves through the shared GPU slots (or, --slots 0 / --mode shard, a context of its own)
        def __init__(self, k, nv_l=None, prog=None, pin=True, trace=None):
            self.nv = nv if nv_l is None else nv_l
            self.N = 1 << self.nv
            self.ctx = None if use_slots else zigz_amd.Context(local_rank)  # raises NoDevice: the product has no CPU path
            c = setup_ctx if use_slots else self.ctx
            # synthetic RV64I ADD/XOR loop; every lane / rank proves a different trace
            self.prog = prog if prog is not None else main_program(self.N, 0 if shard else rank * B + k)  # shard: the same trace everywhere
            # [1/6] VM execution: outside the timed region (`trace`: executed beforehand, by several threads for the large sizes)
            self.trace = trace if trace is not None else host.Trace(self.prog, 0x1000, None, 2 * self.N)
            assert self.trace.num_vars == self.nv, (self.trace.num_vars, self.nv)
            self.d_cols = c.dev_alloc(43 * self.N * 4)
            if pin:
                self.trace.pin(c)  # page-locked trace records (48 B per step, and the 32 / 16 B forms a service uploads): PCIe rate
            self.trace.witness_to_device(c, self.d_cols, self.N)   # [2/6] witness resident in HBM before timing
            c.synchronize()
            if not use_slots:
                c.release_workspaces()  # (the upload's staging, 48 B per step, is not needed while the witness stays resident)
            self.proof = None
            self.up_ctx = None
            self.d_next = None
            self.k = k
            self.want_log = False   # slots: fetch the launch log of every proof (the roofline leg)
            self.logs = []
            self.timing = False     # per-launch timestamps on the next proof?  (--slots 0)
            self.timing_every = 0   # 0: leave the context's timing mode alone
            self.proofs_done = 0

        def retrace(self, prog):  # the same lane on another program of the same size (overwrites the resident witness)
            c = setup_ctx if use_slots else self.ctx
            self.prog = prog
            self.trace = host.Trace(prog, 0x1000, None, 2 * self.N)
            assert self.trace.num_vars == self.nv, (self.trace.num_vars, self.nv)
            self.trace.witness_to_device(c, self.d_cols, self.N)
            c.synchronize()

        def drop_upload_context(self):  # (after the PCIe-inclusive leg: the second context + stream of the lane go away)
            if self.up_ctx is not None:
                self.up_ctx.synchronize()
                self.up_ctx.dev_free(self.d_next)
                self.up_ctx.close()
                self.up_ctx = None
                self.d_next = None

        def close(self):
            self.drop_upload_context()
            if self.d_cols is not None:
                (setup_ctx if use_slots else self.ctx).dev_free(self.d_cols)
                self.d_cols = None
            if self.ctx is not None:
                self.ctx.close()
                self.ctx = None

        def prove(self):
            if use_slots:
                self.proof, st, log = self.trace.prove_slots(slots, self.d_cols, self.N, want_log=self.want_log)
                if log:
                    self.logs.append(log)
                st["_timed"] = 1 if self.want_log else 0
                return st, host.last_timings()
            if self.timing_every:  # per-launch timestamps on every timing_every-th proof of this lane
                self.proofs_done += 1
                on = (self.proofs_done + self.k) % self.timing_every == 0
                if on != self.timing:
                    self.timing = on
                    self.ctx.enable_timing(on)
            if shard and dist is not None:
                self.proof = self.trace.prove_sharded(self.ctx, self.d_cols, self.N, dist, allgather_hook)
            else:
                self.proof = self.trace.prove(self.ctx, self.d_cols, self.N, want_bytes="borrow")
            st = self.ctx.stats()
            st["_timed"] = 1 if self.timing else 0
            return st, host.last_timings()

        def prove_and_digest(self):  # self-check steps (untimed): SHA-256 of the proof, taken on the proving thread while
            # the borrowed buffer is still this proof's
            r = self.prove()
            self.digest = hashlib.sha256(self.proof.tobytes()).hexdigest()
            return r

        def upload_and_prove(self):  # PCIe-inclusive: one upload of the compact trace (16 B per step + code table) + one run of the witness
            # kernels per proof, inside the loop.  With slots: inside the proof's GPU slot, into a column buffer the slot owns
            # (upload, expansion and builds on the slot's stream; the other slots' kernels run underneath the copy) -- a lane
            # then needs no resident witness at all.  --slots 0: pipelined as round 3 did -- while this proof runs, the NEXT
            # proof's trace crosses PCIe on a second stream into the lane's other column buffer.
            if use_slots:
                self.proof, st, _ = self.trace.prove_slots(slots, None, 0)
                st["_timed"] = 0
                return st, host.last_timings()
            if self.up_ctx is None:
                self.up_ctx = zigz_amd.Context(local_rank)
                self.d_next = self.up_ctx.dev_alloc(43 * N * 4)
                self.trace.witness_to_device(self.up_ctx, self.d_next, N, wait=False)  # primes the pipeline (set-up call)
            self.up_ctx.synchronize()                            # the upload issued during the previous proof has landed
            self.d_cols, self.d_next = self.d_next, self.d_cols  # prove from it ...
            self.trace.witness_to_device(self.up_ctx, self.d_next, N, wait=False)  # ... while the next one crosses
            return self.prove()

    # --mode shard: the two exchanges of a column-sharded proof are host-resident and a few KiB -> the shared-memory hook
    # (one node); ZIGZ_BENCH_SHARD_HOOK=torch binds torch.distributed (RCCL / gloo) instead
    def make_comm(tag, timeout_s):
        """The exchange transport of a sharded path: every rank calls this (creation is a collective)."""
        if args.exchange == "rccl":
            if backend != "nccl":
                raise RuntimeError("--exchange rccl needs one GPU per rank (this launch runs over %s)" % backend)
            from zigz_amd.shard import RcclComm
            uid = torch.zeros(128, dtype=torch.uint8, device=tdev)
            if rank == 0:
                uid = torch.frombuffer(bytearray(RcclComm.unique_id()), dtype=torch.uint8).to(tdev)
            dist.broadcast(uid, src=0)
            return RcclComm(local_rank, bytes(uid.cpu().numpy().tobytes()), rank, world, max_bytes=1 << 16)
        from zigz_amd.shard import ShmComm
        # the name carries a nonce agreed on by the ranks of THIS launch: ports and job ids get reused
        nonce = torch.tensor([float(int.from_bytes(os.urandom(6), "little"))], dtype=torch.float64, device=tdev)
        dist.all_reduce(nonce, op=dist.ReduceOp.MAX)
        return ShmComm("zigz_bench_%s_%s_%x" % (tag, os.environ.get("MASTER_PORT", "0"), int(nonce.item())), rank, world,
                       max_bytes=1 << 16, timeout_s=timeout_s)

    allgather_hook = None
    if shard and dist is not None:
        if os.environ.get("ZIGZ_BENCH_SHARD_HOOK") == "torch":
            allgather_hook = host.make_allgather(dist)
        else:
            allgather_hook = make_comm("shard", 120.0)
    if use_slots:
        slots = make_slots(slot_count(B, nv), nv)
    lanes = [Lane(k) for k in range(B)]
    setup_ctx.release_workspaces()
    pool = ThreadPoolExecutor(max_workers=B * 11 // 8 + 8)  # (the small-trace leg runs more lanes than the main region)

    def all_contexts():
        if use_slots:
            return slots.contexts() if slots is not None else []
        return [l.ctx for l in lanes if l.ctx is not None]

    def sync_all():
        for c in all_contexts():
            c.synchronize()
        if torch is not None:
            if backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    def gather(futs):
        """Results of all futures; if some failed, the first failure -- but only after EVERY lane has finished, so that what
        handles it (guard) finds no lane in the middle of a proof."""
        out, first = [], None
        for f in futs:
            try:
                out.append(f.result())
            except BaseException as e:  # noqa: BLE001
                first = first or e
        if first is not None:
            raise first
        return out

    def run_step(fn=Lane.prove, which=None):
        ls = lanes if which is None else which
        return gather([pool.submit(fn, l) for l in ls])

    def timed(steps, fn=Lane.prove, which=None):
        """Exactly `steps` steps (= steps x len(lanes) proofs) between barrier + synchronize brackets.  Every lane proves its
        `steps` traces back to back on its own host thread + HIP stream -- a proving service does not run its lanes in
        lockstep -- and the region ends when the last lane has finished.  Returns (seconds, accumulated stats, phases)."""
        ls = lanes if which is None else which
        acc = {}
        acc_t = {}  # the same sums over the lanes that run in timing mode only
        n_t = 0
        phases = {}

        def lane_loop(l):
            if use_slots and fn is Lane.prove and not l.want_log:
                # the lane's `steps` proofs back to back inside the library (zigzh_prove_trace_slots_repeat): ~70 us of
                # interpreter time per proof x 93 lane threads behind one interpreter lock cap a process at ~14 k proofs/s, which
                # the small traces exceed (2^16: 1.0 -> 1.3 G steps/s); sums of the statistics and phase timings come back
                l.proof, st, tm = l.trace.prove_slots_repeat(slots, steps, l.d_cols, l.N)
                st["_timed"] = 0
                return [(st, tm)]
            if use_slots and fn is Lane.upload_and_prove:  # (the same lane loop, every proof uploading its trace inside its slot)
                l.proof, st, tm = l.trace.prove_slots_repeat(slots, steps, None, 0)
                st["_timed"] = 0
                return [(st, tm)]
            out = []
            for _ in range(steps):
                out.append(fn(l))
            return out
        sync_all()
        c0 = time.process_time()
        t0 = time.perf_counter()
        for l, res in zip(ls, gather([pool.submit(lane_loop, l) for l in ls])):
            for st, ph in res:
                for k, v in st.items():
                    acc[k] = acc.get(k, 0) + v
                    if st.get("_timed"):
                        acc_t[k] = acc_t.get(k, 0) + v
                n_t += 1 if st.get("_timed") else 0
                for k, v in ph.items():
                    phases[k] = phases.get(k, 0.0) + v
        sync_all()
        acc["_timed"] = acc_t
        acc["_timed_proofs"] = n_t
        acc["process_cpu_s"] = time.process_time() - c0  # all threads of this process (lanes, sponge servers, helpers)
        return time.perf_counter() - t0, acc, phases

    main_fn = Lane.upload_and_prove if (args.upload and use_slots) else Lane.prove
    run_step(main_fn)  # set-up, not a step: first-use allocation of every lane's workspaces (2.7 GiB of tree each), thread start-up
    for _ in range(args.warmup):
        run_step(main_fn)
    # Per-launch kernel timestamps (HIP events on every launch's own stream) cost host CPU -- two events per launch, collected
    # per proof, and their completion handlers on the runtime's event thread: ~0.5 ms of the ~8 ms of CPU per proof on a
    # host-bound box.  THE timed region therefore carries none (VERDICT r3 #3d); `roofline` comes from a short leg right after it
    # -- the same lanes, the same workload, every launch of every proof timed -- whose own rate stands beside it
    # (`roofline.leg_value`).  With a context per lane (--slots 0) ZIGZ_BENCH_TIMING_EVERY = n times every n-th proof of each
    # lane inside the timed region instead, as round 3 did.
    timing_every = int(os.environ.get("ZIGZ_BENCH_TIMING_EVERY", "0")) if not use_slots else 0
    for k, l in enumerate(lanes):
        if l.ctx is not None and timing_every:
            l.timing_every = timing_every
            l.timing = timing_every == 1
            l.ctx.enable_timing(l.timing)
    if os.environ.get("ZIGZ_BENCH_THREAD_CPU"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import thread_cpu
        snap0 = thread_cpu.snapshot()
    dt, acc, phases = timed(args.steps, main_fn)  # ---- THE timed region
    if os.environ.get("ZIGZ_BENCH_THREAD_CPU"):
        rows, by = thread_cpu.diff(snap0, thread_cpu.snapshot())
        sys.stderr.write("thread CPU over the timed region (%.2f s wall):\n" % dt)
        for comm, (n, sec) in by:
            sys.stderr.write("  %-20s threads %3d  cpu %.2f s\n" % (comm, n, sec))
        allrows, _ = thread_cpu.diff(snap0, thread_cpu.snapshot(), top=10000)
        sys.stderr.write("  per thread, descending: " + " ".join("%.2f" % r[0] for r in allrows) + "\n")
        for r in allrows[:14]:
            try:
                wchan = open("/proc/self/task/%d/wchan" % r[2]).read().strip()
            except OSError:
                wchan = "?"
            sys.stderr.write("    tid %d: user %.2f s  sys %.2f s  wchan %s\n" % (r[2], r[3], r[4], wchan))
    nproofs = args.steps * B
    local_steps = float(sum(l.trace.num_steps for l in lanes))
    trace_steps, trace_lookups, prog = lanes[0].trace.num_steps, lanes[0].trace.num_lookups, lanes[0].prog
    proof = lanes[0].proof.tobytes()  # (the legs below reuse the borrowed buffers and, at the end, the lanes' traces)

    # ---- the roofline leg: the timed region once more, shorter, with every launch of every proof carrying its own begin / end
    # on ONE time axis (zigz_ctx_set_epoch / zigz_ctx_launch_log: microseconds since a common epoch, comparable across the
    # slots' streams).  What is taken from it per kernel class is the UNION of its launches' intervals -- the time during which
    # at least one launch of the class was on the GPU -- next to their count and the sum of their durations: dozens of proofs
    # share the chip, so the sum exceeds the wall clock, the union cannot (VERDICT r3 #1b).
    def union_us(iv):
        if not iv:
            return 0.0
        iv = sorted(iv)
        tot, cs, ce = 0.0, iv[0][0], iv[0][1]
        for a, b in iv[1:]:
            if a > ce:
                tot += ce - cs
                cs, ce = a, b
            elif b > ce:
                ce = b
        return tot + (ce - cs)

    roof_leg = None
    if use_slots and not shard:
        try:
            ctxs = slots.contexts()
            ctxs[0].set_epoch()
            for c in ctxs[1:]:
                c.set_epoch(ctxs[0])
            for c in ctxs:
                c.enable_timing(True)
            for l in lanes:
                l.want_log, l.logs = True, []
            steps_r = max(2, min(args.steps, 4))
            dt_r, acc_r, _ = timed(steps_r)
            recs = [r for l in lanes for lg in l.logs for r in lg]
            # per level of k_level_hash: a proof's class-5 launches are its levels 0, 1, .. in order
            by_level = {}
            for l in lanes:
                for lg in l.logs:
                    lv = 0
                    for c_, _p, a, b in lg:
                        if c_ == 5:
                            by_level.setdefault(lv, []).append((a, b))
                            lv += 1
            roof_leg = {"dt": dt_r, "steps": steps_r, "acc": acc_r, "recs": recs, "by_level": by_level}
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("bench.py: roofline leg failed: %r\n" % (e,))
        finally:
            for c in slots.contexts():
                c.enable_timing(False)
            for l in lanes:
                l.want_log, l.logs = False, []

    # ---- legs outside the timed region (same run, same resident data).  Each runs under a guard: a leg that fails (out of
    # HBM on an unusual box, say) is recorded in detail.leg_errors and left out; the line with `value` goes out regardless.
    extras = not args.no_extras and not shard
    solo = kern = gpu_one = None
    legs = {}         # name -> {"dt", "steps", "trace_steps", ...}: what is reduced over the ranks below
    leg_errors = {}
    self_check = {}

    def guard(name, fn):
        t_leg = time.perf_counter()
        try:
            r = fn()
            if rank == 0:  # (one progress line per leg on stderr: a long default run is not silent)
                try:
                    free_gib = setup_ctx.mem_info()[0] / 2.0**30
                except Exception:
                    free_gib = float("nan")
                sys.stderr.write("bench.py: leg %s: %.1f s, %.0f GiB of HBM free\n" % (name, time.perf_counter() - t_leg, free_gib))
                sys.stderr.flush()
            return r
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001
            leg_errors[name] = repr(e)[:200]
            sys.stderr.write("bench.py: leg %s failed: %r\n" % (name, e))
            try:  # what the backend recorded (the failing HIP call, if any)
                msgs = {zigz_amd._ffi.lib.zigz_last_error(c.h).decode(errors="replace") for c in all_contexts()}
                sys.stderr.write("bench.py:   backend: %s\n" % "; ".join(sorted(m for m in msgs if m))[:600])
            except Exception:
                pass
            try:
                for c in all_contexts():
                    c.synchronize()
                    c.release_workspaces()
            except Exception:
                pass
            return None

    if extras:
        def leg_solo():
            ksolo = max(3, min(args.steps, 8))
            run_step(which=lanes[:1])
            for c in all_contexts():  # (per-launch timestamps: this leg's stats feed detail.single_proof)
                c.enable_timing(True)
            try:
                dts, accs, _ = timed(ksolo, which=lanes[:1])       # one proof at a time on the GPU
            finally:
                for c in all_contexts():
                    c.enable_timing(False)
            r = {"dt": dts, "n": ksolo, "acc": accs}
            if rank == 0:  # Prover.prove from program bytes: VM + compact trace upload + witness kernels + proof + serialisation
                host.prove(setup_ctx, lanes[0].prog, 0x1000, None, 2 * N)
                t0 = time.perf_counter()
                for _ in range(2):
                    host.prove(setup_ctx, lanes[0].prog, 0x1000, None, 2 * N)
                r["from_program_ms"] = (time.perf_counter() - t0) / 2 * 1e3
                t0 = time.perf_counter()
                host.Trace(lanes[0].prog, 0x1000, None, 2 * N)
                r["vm_ms"] = (time.perf_counter() - t0) * 1e3
            setup_ctx.release_workspaces()
            return r
        solo = guard("single_proof", leg_solo)

        # What the GPU alone needs per proof, and what of it is hashing: the commit path of every lane -- Merkle builds, roots,
        # 43 evals + openings, on the resident columns -- back to back WITHOUT the host transcript, three times: complete;
        # with the hash launches left out; with the structure passes left out too (option debug_skip: wrong trees, measurement
        # only).  The kernel-time shares of `roofline` sum launch durations, which count a small kernel's wait for a wave slot
        # next to 80 proofs' hash waves as its time; these are wall-clock differences.
        def gpu_bound(skip, nl=28, timing=False):
            """nl worker threads, each: take a context (a GPU slot, or its lane's own), commit job on a lane's resident
            columns -- begin, roots, 43 evals + openings, end -- back to back, no transcript.  timing: per-launch timestamps
            (one worker: the launches of ONE proof at a time on a GPU that stays busy, i.e. at its clocks)."""
            regs, small = 0x7fffffff << 2, (1 << 1) | (0x3f << 33) | (1 << 42)
            want = {"small_domain_mask": small, "run_aware_mask": regs | (3 << 40), "cons_group_mask": 1 | (1 << 1) | (0x7f << 33) | (1 << 42)}
            import numpy as np
            rng_p = np.random.default_rng(4242)
            pts = rng_p.integers(0, 2013265921, size=(43, nv), dtype=np.uint64)
            gl = lanes[:min(len(lanes), nl)]  # (28 workers keep the GPU full; more get in each other's way: 0.57 against 0.49 ms)
            cs = all_contexts() if use_slots else [l.ctx for l in gl]
            saved = [{k: c.get_option(k) for k in want} for c in cs]
            iters = max(4, min(2 * args.steps, 20))
            acc_g = {}

            def once(l):
                if use_slots:
                    with slots.borrow() as c:
                        return once_on(c, l)
                return once_on(l.ctx, l)

            def loop_in_library(l):  # (no interpreter in the loop: ~70 us per job per worker behind one lock would be in the figure)
                slots.commit_path_repeat(l.d_cols, N, nv, pts, (want["small_domain_mask"], want["run_aware_mask"], want["cons_group_mask"]), iters)
                return []

            def once_on(c, l):
                job = zigz_amd.CommitJob(c, d_cols=l.d_cols, ncols=43, nv=nv, col_stride=N)
                job.roots()
                job.open_all(pts)
                job.end()
                return c.stats() if timing else None

            def loop(l):
                out = []
                for _ in range(iters):
                    out.append(once(l))
                return out
            try:
                for c in cs:
                    c.enable_timing(timing)
                    for k, v in want.items():
                        c.set_option(k, v)
                gather([pool.submit(once, l) for l in gl])
                for c in cs:
                    c.set_option("debug_skip", skip)
                gather([pool.submit(once, l) for l in gl])
                sync_all()
                t0 = time.perf_counter()
                res = gather([pool.submit(loop_in_library if (use_slots and not timing) else loop, l) for l in gl])
                dtg = time.perf_counter() - t0
                for r in res:
                    for st in r:
                        for k, v in (st or {}).items():
                            acc_g[k] = acc_g.get(k, 0) + v
            finally:
                for c, sv in zip(cs, saved):
                    c.enable_timing(False)
                    c.set_option("debug_skip", 0)
                    for k, v in sv.items():
                        c.set_option(k, v)
                for l in gl:
                    l.timing = False
            return {"dt": dtg, "steps": iters, "lanes": len(gl), "trace_steps": float(sum(l.trace.num_steps for l in gl)), "acc": acc_g}
        if args.merkle == "cons":
            for skip, name in ((0, "gpu_all"), (1, "gpu_nohash"), (2, "gpu_nohash_nostruct")):
                legs[name] = guard(name, lambda skip=skip: gpu_bound(skip))
            gpu_one = guard("gpu_one_at_a_time", lambda: gpu_bound(0, nl=1, timing=True))

        # (after the legs above: this one gives every lane a second context and stream for its uploads -- dropped again right
        # after it: a process with twice the streams runs everything a few percent slower)
        def leg_pcie():
            nonlocal slots
            if use_slots:  # a slot is held ~2 ms longer per proof (the 33 MB upload + the witness kernels): a few more of them
                slots.close()
                slots = make_slots(min(16, slot_count(B, nv) + 4), nv)
            try:
                run_step(Lane.upload_and_prove)
                run_step(Lane.upload_and_prove)
                dtp, _, _ = timed(args.steps, Lane.upload_and_prove)   # batch B again, trace upload + witness kernels inside
            finally:
                if use_slots:
                    slots.close()
                    slots = make_slots(slot_count(B, nv), nv)
            form, nbytes = lanes[0].trace.upload_form()  # the record the host mirror chose for these traces, bytes over PCIe per proof
            return {"dt": dtp, "steps": args.steps, "trace_steps": local_steps, "upload_form": form, "upload_bytes": nbytes}
        legs["pcie"] = guard("pcie_inclusive", leg_pcie)
        guard("drop_upload_contexts", lambda: [l.drop_upload_context() for l in lanes])
        guard("release", lambda: [c.release_workspaces() for c in all_contexts()])  # (the slots' column buffers and staging)


        digests = []

        def leg_self_check():  # untimed: every lane's proof under this build ...
            run_step(Lane.prove_and_digest)
            digests.extend(l.digest for l in lanes)
            run_step(Lane.prove_and_digest, which=lanes[:1])  # ... lane 0 once more ALONE (its transcript then runs through the
            self_check["lane0_alone_equals_lane0_in_batch"] = lanes[0].digest == digests[0]  # single-state code, not the 8-way)
            return True
        have_digests = guard("self_check", leg_self_check)

        # the same batch under the other Merkle builds (identical proofs), for the record.  A build without the hints keeps
        # node-addressed trees, 2.75 GiB per proof in flight at 2^20: these legs run on as many of the lanes as fit -- the GPU
        # bounds them long before that -- and give the memory back afterwards
        def variant_lanes():
            for c in all_contexts():
                c.release_workspaces()
            if use_slots:  # (node-addressed trees live in the SLOTS' workspaces now: every lane takes part)
                return lanes
            return lanes[:max(1, min(B, int(setup_ctx.mem_info()[0] * 0.85) // int(3.6 * (1 << 30) * (1 << max(nv - 20, 0)))))]
        vl = guard("merkle_variants", variant_lanes) if have_digests else None
        for mode in ("cons", "struct", "regs", "all", "tables", "dense"):
            if mode == args.merkle or not vl:
                continue

            def leg_variant(mode=mode):
                for c in all_contexts():  # (workspaces only grow: what one build kept must not add to what the next one needs)
                    c.release_workspaces()
                if rank == 0:
                    sys.stderr.write("bench.py: %s on %d lanes, %.0f GiB of HBM free\n" % (mode, len(vl), setup_ctx.mem_info()[0] / 2.0**30))
                set_merkle_mode(mode)
                run_step(Lane.prove_and_digest, which=vl)
                # ... must be byte-identical under every other build (dense hashes every node of every tree)
                self_check["lanes_equal_under_" + mode] = [l.digest for l in vl] == digests[:len(vl)]
                ks = 3 if mode in ("cons", "struct", "regs", "all") else 2
                dtv, accv, _ = timed(ks, which=vl)
                return {"dt": dtv, "steps": ks, "perms": accv["keccak_permutations"] / (ks * len(vl)), "lanes": len(vl),
                        "trace_steps": float(sum(l.trace.num_steps for l in vl))}
            legs["variant:" + mode] = guard("merkle_variant_" + mode, leg_variant)
            set_merkle_mode(args.merkle)
        guard("release", lambda: [c.release_workspaces() for c in all_contexts()])
        if not all(self_check.values()):
            raise SystemExit("bench.py: proofs differ between builds / transcript paths: %r" % self_check)
        if rank == 0:
            kern = guard("kernel_leg", lambda: kernel_leg(setup_ctx, nv, 43, max(3, min(args.kernel_iters, 10)),
                                                          big_nv=24 if nv <= 22 else 0))
            guard("release", lambda: setup_ctx.release_workspaces())
        # The structure-aware levels make the GPU time depend on the trace, so the same batch also runs on three other traces
        # (same lanes, same everything else; the lanes' witness buffers are overwritten: these are the last legs on them):
        #   worst     the worst case BY CONSTRUCTION for the run-aware register levels: a loop that writes 30 different
        #             registers in turn, so the <= N change points of the 31 register columns are spread evenly over all of them
        #   mixed     BASELINE config 4's RV64IM mix (MUL / DIVU / REM / LD / SD / *W in a 12-step loop) at this size
        #   straight  a program that never loops (~2^20 different instructions, each executed once): the content-addressed group
        #             finds nothing and is dropped on the device, its columns are built from the tables / densely
        # (8 timed steps: the lanes leave the two learning steps in lockstep -- all on the GPU, then all in their transcripts --
        # and need a step or two to spread out again, which is the state a service runs in)
        OTHER_STEPS = 8
        def other_trace(make_prog, steps_l):
            for k, l in enumerate(lanes):
                l.retrace(make_prog(rank * B + k))
            # two untimed steps: a context that meets a new kind of trace learns the room its lists need (a build repeated in
            # zigz_commit_roots) and, where the group finds nothing twice in a row, stops trying it; what is timed is the
            # service's steady state on that trace, and `rebuilds` says how many builds were repeated inside the timed steps
            def repeated():  # (a context counts the builds it had to repeat since it was created)
                return sum(c.stats()["rebuilds"] for c in all_contexts())
            r0 = repeated()
            run_step()
            run_step()
            r1 = repeated()
            dtl, accl, _ = timed(steps_l)
            return {"dt": dtl, "steps": steps_l, "perms": accl["keccak_permutations"] / (steps_l * B),
                    "rebuilds": (repeated() - r1) / (steps_l * B), "rebuilds_while_learning": (r1 - r0) / (2.0 * B),
                    "trace_steps": float(sum(l.trace.num_steps for l in lanes))}
        legs["worst"] = guard("register_worst_case", lambda: other_trace(lambda i: programs.register_round_robin((N - 2) // 31 - i), OTHER_STEPS))
        legs["mixed"] = guard("config4_mixed", lambda: other_trace(lambda i: programs.mixed_loop((N - 8) // 12 - i), OTHER_STEPS))

        def leg_straight():  # (last: a context whose group was dropped twice in a row stops trying it for its next 15 jobs)
            base = programs.straight_line_program(1000 + rank, int(0.95 * N))  # one program per rank, a prefix per lane
            return other_trace(lambda i: base[:4 * (int(0.95 * N) - (i - rank * B))], OTHER_STEPS)
        legs["straight"] = guard("straight_line", leg_straight)

    # ---- N > 1, traces mode: also ONE proof per step sharded by column over the N ranks (the strong-scaling variant of
    # --mode shard), reported in the same line.  Exchanges go through the shared-memory hook (host-resident payloads of a
    # few KiB; no device collective), every wait has a timeout, and any failure is recorded instead of losing the line.
    shard_leg = None
    if world > 1 and not shard and not args.no_extras:
        comm = None
        sd = None
        ok = 1.0
        try:  # set-up: a rank that fails before the transport exists is noticed by the others as a timeout of its creation
            sctx = setup_ctx
            sprog = programs.add_xor_loop((N - 3) // 4)  # the same trace on every rank
            strace = host.Trace(sprog, 0x1000, None, 2 * N)
            sd = sctx.dev_alloc(43 * N * 4)
            strace.witness_to_device(sctx, sd, N)
        except Exception as e:
            ok = 0.0
            shard_leg = {"error": repr(e)[:300]}
        try:
            comm = make_comm("leg", 60.0)  # (collective: every rank gets here, whatever happened above)
        except Exception as e:
            ok = 0.0
            shard_leg = {"error": repr(e)[:300]}
        t = torch.tensor([ok], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)  # every rank takes part; the leg runs only if all of them are ready
        if float(t.item()) == 1.0:
            try:
                strace.prove_sharded(sctx, sd, N, None, comm)  # warm-up
                ks = max(3, min(args.steps, 10))
                t0 = time.perf_counter()
                for _ in range(ks):
                    sp = strace.prove_sharded(sctx, sd, N, None, comm)  # the exchanges inside synchronise the ranks
                sdt = time.perf_counter() - t0
                shard_leg = {"exchange": args.exchange, "ms_per_proof": sdt / ks * 1e3, "steps_per_s": strace.num_steps * ks / sdt, "proofs": ks,
                             "proof_bytes": len(sp), "accepts": host.verify(sp.tobytes(), sprog) == "Accept" if rank == 0 else None}
            except Exception as e:  # a peer died: the shared-memory waits time out on every rank alike
                shard_leg = {"error": repr(e)[:300]}
        elif shard_leg is None:
            shard_leg = {"error": "another rank could not set the leg up"}
        try:
            if sd is not None:
                setup_ctx.dev_free(sd)
            if comm is not None:
                comm.close()
        except Exception:
            pass

    # ---- the other trace sizes of the north-star (2^16 .. 2^24), same bench trace, same build: small step counts, after the
    # main lanes have given their HBM back.  With GPU slots a lane holds its witness only (2.75 GiB at 2^24) and the tree / list
    # workspaces belong to the few slots (~7 GiB each there): 64+ proofs in flight at 2^24 instead of 21.
    if extras:
        for l in lanes:
            try:
                l.close()
            except Exception:
                pass
        lanes_main, lanes = lanes, []
        if use_slots:
            try:
                slots.close()
            except Exception:
                pass
            slots = None
        setup_ctx.release_workspaces()
        free_now = setup_ctx.mem_info()[0] // gpu_share
        for nv_s, steps_s in ((16, 40), (22, 6), (24, 8)):
            if nv_s == nv:
                continue
            if use_slots:
                k_s = slot_count(B, nv_s)
                while k_s > 4 and k_s * slot_bytes(nv_s, worst=False) > 0.35 * free_now:
                    k_s -= 1
                b_s = B * 11 // 8 if nv_s <= 17 and nv > 17 else B  # (small traces: more lanes than sponge slots, see above)
                nl = max(1, min(b_s, (int(free_now * 0.92) - k_s * slot_bytes(nv_s, worst=False)) // lane_bytes(nv_s)))
            else:
                # (a context per lane: the bench trace holds 0.66 / 2.5 / 9.9 GiB per proof in flight at 2^20 / 2^22 / 2^24 --
                # resident columns, lists and digests after the first build's learning -- measured; below 2^20 the fixed
                # workspaces dominate)
                per_lane = int(0.66 * (1 << 30) * (1 << max(nv_s - 20, 0))) if nv_s >= 20 else int(0.3 * (1 << 30))
                k_s = 0
                nl = max(1, min(B, int(free_now * 0.8) // per_lane))
            ls = []

            def leg_size(nv_s=nv_s, steps_s=steps_s, nl=nl, ls=ls, k_s=k_s):
                nonlocal slots
                if use_slots:
                    slots = make_slots(k_s, nv_s)
                # (the VM runs of the lanes' traces -- 0.3 s each at 2^24 -- on a few threads; the device part one lane after the other)
                progs_s = [main_program(1 << nv_s, rank * B + k) for k in range(nl)]
                with ThreadPoolExecutor(max_workers=min(8, max(1, ncpu // 2))) as tp:
                    traces_s = list(tp.map(lambda pr: host.Trace(pr, 0x1000, None, 2 << nv_s), progs_s))
                ls.extend(Lane(k, nv_s, prog=progs_s[k], pin=False, trace=traces_s[k]) for k in range(nl))
                del traces_s
                setup_ctx.release_workspaces()
                for _ in range(2 if nv_s > 17 else 6):  # (small traces: until every slot has met a full batch and sized its workspaces)
                    run_step(which=ls)
                dts_, _, _ = timed(steps_s, which=ls)
                return {"dt": dts_, "steps": steps_s, "lanes": nl, "slots": k_s, "trace_steps": float(sum(l.trace.num_steps for l in ls))}
            legs["nv%d" % nv_s] = guard("value_nv%d" % nv_s, leg_size)
            for l in ls:
                try:
                    l.close()
                except Exception:
                    pass
            if use_slots and slots is not None:
                try:
                    slots.close()
                except Exception:
                    pass
                slots = None
            setup_ctx.release_workspaces()
    else:
        lanes_main = lanes

    # every leg of the fixed list is reduced by EVERY rank whether its own attempt succeeded or not (the number of collectives
    # must not depend on what failed where); a leg that failed on any rank is dropped everywhere
    leg_names = ["pcie", "gpu_all", "gpu_nohash", "gpu_nohash_nostruct"] + \
                ["variant:" + m for m in ("cons", "struct", "regs", "all", "tables", "dense")] + \
                ["worst", "mixed", "straight", "nv16", "nv22", "nv24"]
    if torch is not None:
        def allmax(x):
            t = torch.tensor([x], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        def allsum(x):
            t = torch.tensor([x], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return float(t.item())
        dt = allmax(dt)
        if extras:
            for name in leg_names:
                rec = legs.get(name)
                everyone = -allmax(0.0 if rec else 1.0) == 0.0  # nobody missing
                dtm = allmax(rec["dt"] if rec else 0.0)
                tsm = allsum(rec["trace_steps"] if rec else 0.0)
                if rec and everyone:
                    rec["dt"], rec["trace_steps"] = dtm, tsm
                else:
                    legs[name] = None
        s = torch.tensor([local_steps], dtype=torch.float64, device=tdev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_steps = local_steps if shard else float(s.item())  # shard: all ranks worked on the same trace
    else:
        total_steps = local_steps

    if rank == 0:
        assert host.verify(proof, prog) == "Accept"
        ic = isa_counts()
        # ---- `roofline`: the kernel class that dominates, from the roofline leg (every launch of every proof timed, absolute
        # begin / end).  Per class: launches, sum of durations, UNION of the intervals (time with >= 1 launch of the class on
        # the GPU).  achieved = the class's work / its union: what the chip delivered for the class while it had any of it in
        # flight -- bounded by the wall clock, unlike work / sum-of-overlapping-durations (round 3's definition, kept as
        # frac_by_sum_of_durations; it divides by more time than passed).
        CLS = {0: "dense", 1: "dense", 2: "dense", 3: "tables", 4: "structure", 5: "level_hash", 6: "top", 7: "eval"}
        nproofs_all = nproofs
        cls_iv = {}
        acc_l, nproofs_l, wall_us, leg_value = acc, nproofs, dt * 1e6, None
        if roof_leg and roof_leg["recs"]:
            for c_, _p, a, b in roof_leg["recs"]:
                cls_iv.setdefault(CLS.get(c_, "other"), []).append((a, b))
            acc_l, nproofs_l = roof_leg["acc"], roof_leg["steps"] * B
            allr = [x for v in cls_iv.values() for x in v]
            wall_us = max(b for _, b in allr) - min(a for a, _ in allr)  # first launch begin .. last launch end, GPU clock
            leg_value = local_steps * roof_leg["steps"] / roof_leg["dt"]
        elif acc.get("_timed_proofs"):  # a context per lane with ZIGZ_BENCH_TIMING_EVERY: sums of durations only
            acc_l, nproofs_l = acc["_timed"], acc["_timed_proofs"]
        classes = {
            "level_hash": acc_l.get("list_hash_us", 0.0),       # k_level_hash: the list-driven levels 0 .. v - 8
            "structure": acc_l.get("structure_us", 0.0),        # k_runs_stage + k_cons_*: which nodes are hashed (no hashing)
            "top": acc_l.get("top_us", 0.0),                    # k_merkle_top: the last 8 levels
            "dense": acc_l.get("keccak_leaves_us", 0.0) + acc_l.get("keccak_level_wide_us", 0.0) + acc_l.get("keccak_level_small_us", 0.0),
            "tables": acc_l.get("small_domain_us", 0.0),        # k_keccak_small_l01
            "eval": acc_l.get("bind_vec_us", 0.0),              # k_radix_fold
        }
        unions = {k: union_us(v) for k, v in cls_iv.items()}
        kernel_us = max(sum(classes.values()), 1e-9)
        share = {k: v / kernel_us for k, v in classes.items()}
        kernel_us_per_proof = {k: v / max(nproofs_l, 1) for k, v in classes.items()}

        def valu_frac(perms, instr, us):
            return perms * instr / (us / 1e6) / 1e12 / VALU_PEAK_TOPS if us else None
        # ALGORITHMIC instructions per permutation: what the cheapest kernel here that performs one executes for it
        # (k_keccak_leaves, 3 976 VALU instructions: Keccak-f[1600] in the bit-interleaved 32-bit form + digest conversion).  The list
        # kernel's own code is longer (4 428 static, all three of its branches; 4 202 executed per hash by SQ_INSTS_VALU,
        # profiles/r04_one_proof_kernels.json): list entry, leader search and gathers are overhead, not work, so they do not
        # count as achieved (round 3 and the first half of round 4 multiplied by the kernel's static count: x1.114)
        IC_ALG = min(ic["leaves"], ic["level"])
        lh_perms = acc_l.get("list_hash_perms", 0)
        lh_launches = len(cls_iv.get("level_hash", [])) or (nv - 7) * nproofs_l  # one k_level_hash launch per level 0 .. v - 8
        dense_perms = acc_l.get("keccak_leaves_perms", 0) + acc_l.get("keccak_level_wide_perms", 0) + acc_l.get("keccak_level_small_perms", 0)
        traffic = traffic_alg = None
        for tname in ("r04_traffic.json", "r03_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))["k_level_hash"]
                    traffic = tj["hbm_bytes_per_hash"] * lh_perms / max(lh_launches, 1)      # per launch, like `achieved`
                    traffic_alg = tj["algorithmic_bytes_per_hash"] * lh_perms / max(lh_launches, 1)
                    break
                except Exception:
                    pass
        if share["level_hash"] >= share["dense"]:
            busy = unions.get("level_hash") or classes["level_hash"]   # (no intervals: the sum of durations, as round 3)
            roof = {"kernel": "k_level_hash", "bound": "valu", "unit": "T lane-instr/s", "peak": VALU_PEAK_TOPS,
                    "achieved": lh_perms * IC_ALG / (busy / 1e6) / 1e12 if busy else 0.0,
                    "traffic": traffic, "traffic_algorithmic": traffic_alg,
                    "avg_launch_us": classes["level_hash"] / max(lh_launches, 1), "launches": lh_launches,
                    "hashes_per_launch": lh_perms / max(lh_launches, 1), "valu_instr_per_hash": IC_ALG,
                    "valu_instr_per_hash_kernel_static": ic["level_hash"],
                    "frac_with_kernel_static_count": valu_frac(lh_perms, ic["level_hash"], busy),
                    "busy_us": busy, "sum_of_durations_us": classes["level_hash"],
                    "frac_by_sum_of_durations": valu_frac(lh_perms, IC_ALG, classes["level_hash"])}
        else:  # a dense build (--merkle dense / tables): the dense leaf + level kernels dominate
            busy = unions.get("dense") or classes["dense"]
            nl_d = len(cls_iv.get("dense", [])) or nproofs_l
            roof = {"kernel": "k_keccak_leaves+k_keccak_level", "bound": "valu", "unit": "T lane-instr/s", "peak": VALU_PEAK_TOPS,
                    "achieved": dense_perms * IC_ALG / (busy / 1e6) / 1e12 if busy else 0.0,
                    "traffic": None, "traffic_algorithmic": None,
                    "avg_launch_us": classes["dense"] / max(nl_d, 1), "launches": nl_d,
                    "hashes_per_launch": dense_perms / max(nl_d, 1), "valu_instr_per_hash": IC_ALG,
                    "busy_us": busy, "sum_of_durations_us": classes["dense"],
                    "frac_by_sum_of_durations": valu_frac(dense_perms, IC_ALG, classes["dense"])}
        roof["frac"] = roof["achieved"] / VALU_PEAK_TOPS
        roof = {k: roof[k] for k in ("kernel", "bound", "unit", "peak", "achieved", "frac", "traffic", "traffic_algorithmic",
                                     "avg_launch_us", "launches", "hashes_per_launch", "valu_instr_per_hash",
                                     "valu_instr_per_hash_kernel_static", "frac_with_kernel_static_count", "busy_us",
                                     "sum_of_durations_us", "frac_by_sum_of_durations") if k in roof}
        # what share of the leg's wall clock (first launch begin .. last launch end) had >= 1 launch of a class on the GPU
        roof["leg_wall_us"] = wall_us
        roof["busy_share_of_wall"] = roof["busy_us"] / wall_us if unions else None
        roof["leg_value"] = leg_value  # the rate of the roofline leg itself (events on every launch): what the timing costs
        detail_busy = {k: v / wall_us for k, v in unions.items()} if unions else None
        # the structure passes are HBM-side work: their algorithmic bytes (4 B per leaf of the 33 run-aware columns; the group's
        # ten columns twice + 12 B per node of its levels) over their time with ONE proof on the GPU at a time
        struct_bytes = {"cons": 224.0, "struct": 132.0, "regs": 124.0, "all": 140.0}.get(args.merkle, 0.0) * N
        roof["eval_hbm_frac"] = ((acc_l["bind_vec_bytes"] / 1e9) / (acc_l["bind_vec_us"] / 1e6) / HBM_PEAK_GBS) if acc_l.get("bind_vec_us") else None
        # the same class with ONE proof on the GPU at a time (no overlap between proofs), its launches back to back on a GPU that
        # stays busy (the commit path without the host transcript, one worker): the kernel's own quality inside a proof.  The
        # single-proof leg -- 27 ms of host sponge between two 1 ms bursts of GPU work -- measures the same launches on a chip
        # that has dropped its clocks in between (`in_proof_frac_idle_gaps`; the driver's round-3 line: 0.25 against 0.37).
        if gpu_one and gpu_one.get("acc"):
            a = gpu_one["acc"]
            roof["in_proof_frac"] = valu_frac(a.get("list_hash_perms", 0), IC_ALG, a.get("list_hash_us", 0.0))
            if a.get("structure_us") and struct_bytes:
                roof["structure_in_proof_hbm_frac"] = struct_bytes * gpu_one["steps"] / 1e9 / (a["structure_us"] / 1e6) / HBM_PEAK_GBS
        if solo and solo["acc"].get("list_hash_us"):
            a = solo["acc"]
            roof["in_proof_frac_idle_gaps"] = valu_frac(a.get("list_hash_perms", 0), IC_ALG, a.get("list_hash_us", 0.0))
        # all Keccak work of the TIMED REGION over its wall time: a lower bound on what the chip sustained while the bench ran
        roof["timed_region_aggregate_frac"] = acc["keccak_permutations"] * IC_ALG / dt / 1e12 / VALU_PEAK_TOPS
        kl = kern.get("k_keccak_leaves[43x2^%d]" % nv) if kern else None
        if kl:  # what the permutation code reaches back to back on 43 x 2^nv leaves in this process: the ceiling of its mix
            roof["permutation_ceiling_frac"] = kl["frac"]
        if kern:  # the north-star MLE kernels, cold-HBM launches in this run
            for key, name in (("bind", "k_bind_vec[43x2^%d]" % nv), ("half_sums", "k_half_sums[43x2^%d]" % nv),
                              ("radix_fold", "k_radix_fold[43x2^%d]" % nv)):
                if name in kern:
                    roof[key + "_hbm_frac"] = kern[name]["frac"]

        def rate(leg):
            return leg["trace_steps"] * leg["steps"] / leg["dt"]
        host_cpu_ms = acc.get("process_cpu_s", 0.0) / nproofs * 1e3  # CPU time of all threads of the process, per proof
        cfg = {
            "workload": "synthetic RV64I ADD/XOR loop, 2^%d trace, 43 columns in HBM; full Prover.prove hot path; %d proofs/GPU/step" % (nv, B),
            # which side bounded the line, and what each side could carry (whole job, all ranks): the host -- this rank's CPUs /
            # the CPU time one proof costs (sponge servers, lanes, serialiser, runtime threads) -- and the GPU -- the commit path
            # alone, back to back without the transcript (filled in below from the gpu_bound leg).  On a node whose cgroup gives
            # a rank few CPUs the line is host-bound at host_bound_ceiling_value whatever the GPUs could do.
            "bound": None, "host_bound_ceiling_value": (ncpu / host_cpu_ms * 1e3 * trace_steps * world) if host_cpu_ms else None,
            "gpu_bound_ceiling_value": None,
            "gpu_slots": slot_count(B, nv) if use_slots else 0,
            "traces_per_step_per_gpu": B, "sponge_servers": servers,
            "merkle_build": args.merkle, "keccak_permutations_per_proof": acc["keccak_permutations"] / nproofs,
            "ms_per_proof_per_gpu": dt / nproofs * 1e3, "cpus_pinned": pin.get("pinned", 0),
            "parallelism": ("1 proof/step, columns sharded over %d GPUs (strong)" % world) if shard else
                           ("independent traces: %d GPU(s) x %d proofs in flight, no data-path collective" % (world, B)),
        }
        detail = {"merkle_build": "--merkle %s: " % args.merkle + MERKLE_BUILDS[args.merkle], "lookup_steps": trace_lookups,
                  "blocking_sync": bool(blocking), "pin": pin, "hbm_free_at_start": hbm_free, "proof_bytes": len(proof),
                  "host_transcripts": ("%d sponge-server threads per GPU, each advancing up to 8 proofs' transcripts in lock "
                                       "step (8-way AVX-512 Keccak-f); the proofs' own threads sleep meanwhile" % servers)
                                      if servers else "every proof absorbs its transcript on its own host thread",
                  "trace_steps": trace_steps, "host_cpus_available": ncpu,
                  "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else "none",
                  "commit_begin_ms": phases.get("commit_begin", 0.0) / nproofs * 1e3,
                  "kernel_time_shares_roofline_leg": share, "kernel_us_per_proof_roofline_leg": kernel_us_per_proof,
                  "class_busy_share_of_wall_roofline_leg": detail_busy,
                  "roofline_leg": {"steps": roof_leg["steps"], "proofs": roof_leg["steps"] * B, "wall_us": wall_us,
                                   "launches_logged": len(roof_leg["recs"])} if roof_leg else None}
        if roof_leg and roof_leg.get("by_level"):
            # the big levels (0-4 hold 97 % of a proof's hashes and fill the chip alone) against the wall clock of the leg: how much
            # of the time is at least one of them running, how many at once, and how long a launch of each level takes in company
            bl = roof_leg["by_level"]
            big = [iv for lv, v in bl.items() if lv <= 4 for iv in v]
            ev = sorted([(a, 1) for a, _ in big] + [(b, -1) for _, b in big])
            hist, cnt, last = {}, 0, ev[0][0] if ev else 0.0
            for t_, d_ in ev:
                hist[cnt] = hist.get(cnt, 0.0) + (t_ - last)
                last, cnt = t_, cnt + d_
            detail["level_hash_levels_roofline_leg"] = {
                "avg_us_by_level": [round(sum(b - a for a, b in bl[lv]) / len(bl[lv]), 1) for lv in sorted(bl)],
                "big_levels_0_4_union_share_of_wall": union_us(big) / wall_us if big else None,
                "big_levels_at_once_share_of_wall": {str(k_): round(v_ / wall_us, 4) for k_, v_ in sorted(hist.items()) if k_ > 0}}
        for k_ in ("leg_wall_us", "sum_of_durations_us", "eval_hbm_frac"):
            detail["roofline_" + k_] = roof.pop(k_, None)
        if solo:
            cfg["single_proof_ms"] = solo["dt"] / solo["n"] * 1e3
            a = solo["acc"]
            detail["single_proof"] = {
                "steps_per_s": trace_steps * solo["n"] / solo["dt"], "merkle_build_ms": a["merkle_build_us"] / solo["n"] / 1e3,
                "structure_us": a.get("structure_us", 0.0) / solo["n"], "level_hash_us": a.get("list_hash_us", 0.0) / solo["n"],
                "top_us": a.get("top_us", 0.0) / solo["n"], "eval_us": a["eval_us"] / solo["n"],
                "level_hash_gperm_per_s": (a.get("list_hash_perms", 0) / 1e9 / (a["list_hash_us"] / 1e6)) if a.get("list_hash_us") else None,
                "eval_fold_avg_launch_us": a["bind_vec_us"] / max(a["bind_vec_launches"], 1)}
            if "from_program_ms" in solo:
                cfg["from_program_bytes_ms"] = solo["from_program_ms"]  # one Prover.prove incl. VM execution
                detail["single_proof"]["vm_ms"] = solo["vm_ms"]
        if legs.get("pcie"):  # trace upload + witness kernels inside the loop
            cfg["pcie_inclusive_value"] = rate(legs["pcie"])
            detail["pcie_upload"] = {"bytes_per_step_record": legs["pcie"].get("upload_form"), "bytes_per_proof": legs["pcie"].get("upload_bytes")}
        if legs.get("gpu_all"):  # the commit path alone: what the GPU needs per proof (ms), and the same without its hashing
            def gpu_ms(leg):
                return leg["dt"] / (leg["steps"] * leg.get("lanes", B)) * 1e3
            cfg["gpu_bound_ms_per_proof"] = gpu_ms(legs["gpu_all"])
            cfg["gpu_bound_ceiling_value"] = trace_steps / cfg["gpu_bound_ms_per_proof"] * 1e3 * world
            if cfg["host_bound_ceiling_value"]:
                cfg["bound"] = "host" if cfg["host_bound_ceiling_value"] < cfg["gpu_bound_ceiling_value"] else "gpu"
            dec = {"complete": gpu_ms(legs["gpu_all"])}
            if legs.get("gpu_nohash"):
                dec["without_hash_launches"] = gpu_ms(legs["gpu_nohash"])
            if legs.get("gpu_nohash_nostruct"):
                dec["without_hash_and_structure_launches"] = gpu_ms(legs["gpu_nohash_nostruct"])
            if "without_hash_launches" in dec:
                dec["hashing_share_of_gpu_time"] = 1.0 - dec["without_hash_launches"] / dec["complete"]
                roof["hash_share_of_gpu_time"] = dec["hashing_share_of_gpu_time"]  # (takes the place of structure_in_proof_hbm_frac)
                detail["structure_in_proof_hbm_frac"] = roof.pop("structure_in_proof_hbm_frac", None)
            detail["gpu_ms_per_proof_commit_path_only"] = dec
        for key, name in (("worst", "register_worst_case_value"), ("mixed", "config4_mixed_value"), ("straight", "straight_line_value")):
            if legs.get(key):
                cfg[name] = rate(legs[key])
                detail.setdefault("other_traces_keccak_permutations_per_proof", {})[key] = legs[key]["perms"]
                detail.setdefault("other_traces_rebuilds_per_proof", {})[key] = [legs[key]["rebuilds_while_learning"], legs[key]["rebuilds"]]
        variants = {k.split(":", 1)[1]: v for k, v in legs.items() if k.startswith("variant:") and v}
        if variants:
            if "dense" in variants:
                cfg["dense_merkle_value"] = rate(variants["dense"])
            detail["merkle_variants"] = {
                m: {"value": rate(vv), "keccak_permutations_per_proof": vv["perms"], "proofs_in_flight": vv["lanes"]}
                for m, vv in variants.items()}
        if self_check:
            cfg["self_check_ok"] = bool(all(self_check.values()))
            detail["self_check"] = dict(self_check, note="SHA-256 of every lane's 2^%d proof: identical under every "
                                        "Merkle build, and lane 0 alone (single-state transcript code) vs in the batch "
                                        "(sponge service)" % nv)
        for nv_s in (16, 22, 24):
            sv = legs.get("nv%d" % nv_s)
            if sv:
                cfg["value_nv%d" % nv_s] = rate(sv)
                detail["value_nv%d_lanes" % nv_s] = sv["lanes"]
        if leg_errors:
            detail["leg_errors"] = leg_errors
        if shard_leg:
            for k, v in shard_leg.items():
                detail["one_proof_over_all_gpus_" + k] = v
            if "ms_per_proof" in shard_leg:
                cfg["one_proof_over_all_gpus_ms"] = shard_leg["ms_per_proof"]
        if kern:
            detail["kernel_leg"] = {k: {"frac": v["frac"], "avg_us": v["avg_us"]} for k, v in kern.items()}
        detail["host_phase_ms_per_proof"] = {k: v / nproofs * 1e3 for k, v in phases.items()}
        detail["host_cpu_ms_per_proof"] = host_cpu_ms
        detail["host_cpus_busy"] = acc.get("process_cpu_s", 0.0) / dt
        detail["host_keccak"] = zigz_amd._ffi.lib.zigz_host_keccak_impl().decode()
        # (`roofline` stays at 24 flat keys -- what the driver's record keeps per dict --, the figures a reader wants first: the
        # rest moves to detail)
        for k_ in ("valu_instr_per_hash_kernel_static", "leg_value", "busy_share_of_wall"):
            if k_ in roof and len(roof) > 24:
                detail.setdefault("roofline_more", {})[k_] = roof.pop(k_)
        out = {
            "metric": "trace steps/sec proved (BabyBear, 2^%d RV64I trace)" % nv,
            "value": total_steps * args.steps / dt,
            "unit": "trace steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": cfg,
            "roofline": roof,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        out["detail"] = detail
        emit(json.dumps(out))
    pool.shutdown()
    for l in lanes_main:
        l.close()
    if slots is not None:
        slots.close()
    setup_ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
