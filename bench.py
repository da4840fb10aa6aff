#!/usr/bin/env python3
"""bench.py -- trace steps/sec proved on the synthetic RV64I ADD/XOR loop at a 2^20 trace (BASELINE config 3).

A "step" is ONE full pass of the prover hot path over one batch of `--batch` independent 2^20-step traces
(default 4 per GPU; every trace gets its own complete proof) whose 43 witness columns each are already
resident in HBM when the timed region starts.  Per trace: the exact Fiat-Shamir schedule of Prover.prove
(public inputs, SUMCHECK_BEGIN, one LASSO_TABLE absorption per lookup step, POLY_COMMITMENTS, 43*v
challenges, OPENING_CLAIMS), 43 SHA3 Merkle commits, 43 MLE evaluations, 43 openings, packagePublicIO and
the ZIGZ v1 serialisation of the proof.  Nothing is skipped or cached between steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--nv 20] [--no-cpu-baseline]

Why a batch: one proof is bounded by its own sequential SHA3 transcript on ONE host core (19 bytes absorbed per
lookup step, ~30 ms at 2^20) while its GPU work takes ~10 ms, so a proving service keeps the GPU busy by
running several proofs per GPU concurrently (one host thread + one HIP stream each).  `--batch 1` measures
single-proof latency.

N > 1 is launched by torch.distributed.run (one rank per GPU, backend nccl = RCCL).  The path shards by
independent traces (one proof per rank, no data-path collective): scaling = "weak"; value = all ranks'
trace steps / max-over-ranks time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
# 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz (max clock) int32 VALU lane-ops/s
VALU_PEAK_OPS = 256 * 4 * 32 * 2.4e9
KECCAK_OPS = 4020.0  # VALU instructions per permutation in k_keccak_* (2756 v_bitop3 + 1204 v_alignbit + misc)


def cpu_baseline(nv, program, num_lookups, sample_cols):
    """Reference algorithm on the host CPU (oracle = literal C port, 1 thread) on a bounded sample of the same
    workload: `sample_cols` of the 43 columns through commit + eval + open exactly as prover.zig:405-431 does
    (scaled x43/sample_cols), plus the sequential transcript at the measured single-thread SHA3 rate."""
    import oracle_lib as O
    P = O.P_BB
    cols, nv_o, ns = O.witness_from_program(P, program, 0x1000, None, 1 << (nv + 1))
    assert nv_o == nv
    pts = O.splitmix64_field(99, sample_cols * nv).reshape(sample_cols, nv)
    t0 = time.perf_counter()
    for c in range(sample_cols):
        O.commit_column_literal(P, cols[(c * 43) // sample_cols], pts[c])
    t_cols = time.perf_counter() - t0
    nbytes = 19 * num_lookups + 40 * nv + 43 * 32 * (1 + nv)  # LASSO_TABLE absorptions dominate the sponge input
    buf = bytes(min(nbytes, 1 << 22))
    t0 = time.perf_counter()
    O.sha3_256(buf)
    t_transcript = (time.perf_counter() - t0) * (nbytes / len(buf))
    total = t_cols * (43.0 / sample_cols) + t_transcript
    return {"value": ns / total, "unit": "trace steps/s", "cores": 1, "kind": "port",
            "sample": "%d of 43 columns at 2^%d through commit + 2 naive evals + recompute-on-open (%.1f s), scaled "
                      "x43/%d; + %d B of sequential transcript at the measured 1-thread SHA3 rate (%.0f ms); "
                      "oracle/zigz_oracle.c, gcc -O3" % (sample_cols, nv, t_cols, sample_cols, nbytes, t_transcript * 1e3),
            "seconds_per_proof_est": total}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4, help="independent traces proven concurrently per GPU per step "
                    "(one proof alone is bound by its sequential host transcript; 4 lanes saturate the GPU; 6 give +2 %% "
                    "but the MLE kernel then waits for CU slots behind other proofs' Keccak workgroups, which distorts "
                    "the per-launch roofline figure)")
    ap.add_argument("--mode", choices=["traces", "shard"], default="traces",
                    help="traces (default, the headline): every GPU proves its own traces, no data-path collective, weak "
                    "scaling.  shard: ONE proof per step, its 43 columns sharded over the GPUs (two all-gathers of 43 x "
                    "32 B and 43 x (24 + 33 v) B per proof), strong scaling; bounded by the sequential host transcript "
                    "that every rank replays (DESIGN.md s7)")
    ap.add_argument("--nv", type=int, default=20, help="log2 of the padded trace length (BASELINE config 3: 20)")
    ap.add_argument("--dedup", action="store_true", help="run-aware Merkle build (option merkle_dedup); default off: "
                    "the headline is measured with the dense, data-independent build")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cols", type=int, default=12, help="columns of the CPU baseline sample (~1.1 s each)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    backend = os.environ.get("ZIGZ_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N > 1 path on a 1-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)  # tolerates a launcher that masks devices per rank
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
    tdev = "cuda" if backend == "nccl" else "cpu"

    import zigz_amd
    from zigz_amd import host
    import programs

    from concurrent.futures import ThreadPoolExecutor

    nv = args.nv
    N = 1 << nv
    shard = args.mode == "shard"
    B = 1 if shard else max(1, args.batch)

    class Lane:  # one trace + its own context (HIP stream, workspaces) + resident witness
        def __init__(self, k):
            self.ctx = zigz_amd.Context(local_rank)  # raises NoDevice: the product has no CPU path
            if args.dedup:
                self.ctx.set_option("merkle_dedup", 1)
            # synthetic RV64I ADD/XOR loop; every lane / rank proves a different trace
            self.prog = programs.add_xor_loop((N - 3) // 4 - (0 if shard else rank * B + k))  # shard: the same trace everywhere
            self.trace = host.Trace(self.prog, 0x1000, None, 2 * N)  # [1/6] VM execution: outside the timed region
            assert self.trace.num_vars == nv, (self.trace.num_vars, nv)
            self.d_cols = self.ctx.dev_alloc(43 * N * 4)
            self.trace.witness_to_device(self.ctx, self.d_cols, N)   # [2/6] witness resident in HBM before timing
            self.ctx.synchronize()
            self.proof = None

        def prove(self):
            if shard and dist is not None:
                self.proof = self.trace.prove_sharded(self.ctx, self.d_cols, N, dist, allgather_hook)
            else:
                self.proof = self.trace.prove(self.ctx, self.d_cols, N, want_bytes="borrow")
            return self.ctx.stats(), host.last_timings()

    allgather_hook = host.make_allgather(dist) if (shard and dist is not None) else None
    lanes = [Lane(k) for k in range(B)]
    pool = ThreadPoolExecutor(max_workers=B)

    def sync_all():
        for l in lanes:
            l.ctx.synchronize()
        if torch is not None:
            if backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    def run_step():
        return [f.result() for f in [pool.submit(l.prove) for l in lanes]]

    run_step()  # set-up, not a step: first-use allocation of every lane's workspaces (2.7 GiB of tree each), thread start-up
    for _ in range(args.warmup):
        run_step()
    for l in lanes:
        l.ctx.enable_timing(True)
    bind_us = bind_bytes = bind_launches = 0
    merkle_us = eval_us = 0.0
    perms = 0
    phases = {}
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for st, ph in run_step():
            bind_us += st["bind_vec_us"]; bind_bytes += st["bind_vec_bytes"]; bind_launches += st["bind_vec_launches"]
            merkle_us += st["merkle_build_us"]; eval_us += st["eval_us"]; perms += st["keccak_permutations"]
            for k, v in ph.items():
                phases[k] = phases.get(k, 0.0) + v
    sync_all()
    dt = time.perf_counter() - t0
    # the same kernel with the GPU otherwise idle (one more proof on lane 0 alone), outside the timed region:
    # under --batch > 1 the timed-region launches share the chip with other proofs' Keccak kernels
    solo_runs = [lanes[0].prove()[0] for _ in range(3)]
    solo_st = {k: sum(r[k] for r in solo_runs) / len(solo_runs) for k in solo_runs[0]}
    local_steps = float(sum(l.trace.num_steps for l in lanes))
    trace = lanes[0].trace
    prog = lanes[0].prog
    proof = lanes[0].proof
    nproofs = args.steps * B
    if torch is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([local_steps], dtype=torch.float64, device=tdev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_steps = local_steps if shard else float(s.item())  # shard: all ranks worked on the same trace
    else:
        total_steps = local_steps

    if rank == 0:
        proof = proof.tobytes()
        assert host.verify(proof, prog) == "Accept"
        ach = (bind_bytes / 1e9) / (bind_us / 1e6) if bind_us > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "bind_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "trace steps/sec proved (BabyBear, 2^%d RV64I trace)" % nv,
            "value": total_steps * args.steps / dt,
            "unit": "trace steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "synthetic RV64I ADD/XOR loop, 2^%d trace, 43 witness columns resident in HBM; full "
                                   "Prover.prove hot path incl. Fiat-Shamir transcript and ZIGZ v1 serialisation; "
                                   "%d independent traces (proofs) per GPU per step" % (nv, B),
                       "trace_steps": trace.num_steps, "lookup_steps": trace.num_lookups, "proof_bytes": len(proof),
                       "merkle_build": "run-aware (merkle_dedup)" if args.dedup else "dense",
                       "traces_per_step_per_gpu": B, "ms_per_proof_per_gpu": dt / nproofs * 1e3,
                       "parallelism": ("one proof per step, 43 columns sharded over %d GPU(s)" % world) if shard else
                                      "independent traces: %d GPU x %d concurrent proofs" % (world, B)},
            "roofline": {"kernel": "k_radix_fold (MLE bind of the top v-10 variables of all 43 columns in one pass: the bulk "
                                   "of the 43 evals inside the timed region; 4 B read per element + partial sums)",
                         "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "uncontended": {"achieved": (solo_st["bind_vec_bytes"] / 1e9) / (solo_st["bind_vec_us"] / 1e6),
                                         "frac": (solo_st["bind_vec_bytes"] / 1e9) / (solo_st["bind_vec_us"] / 1e6) / HBM_PEAK_GBS,
                                         "note": "same launches, one proof at a time on the GPU right after the timed region (mean of 3)"},
                         "launches_per_proof": bind_launches / nproofs,
                         "avg_launch_us": bind_us / max(bind_launches, 1),
                         "algorithmic_bytes_per_launch": bind_bytes / max(bind_launches, 1)},
            "kernels": {"merkle_build_ms_per_proof": merkle_us / nproofs / 1e3,
                        "eval_ms_per_proof": eval_us / nproofs / 1e3,
                        "host_phase_ms_per_proof": {k: v / nproofs * 1e3 for k, v in phases.items()},
                        "host_keccak": zigz_amd._ffi.lib.zigz_host_keccak_impl().decode(),
                        # Keccak Merkle build of ONE proof alone on the GPU (the per-proof figures above are wall
                        # times of concurrent lanes): permutations/s and the share of the int-VALU issue peak
                        "uncontended": {
                            "merkle_build_ms": solo_st["merkle_build_us"] / 1e3,
                            "keccak_gperm_per_s": (solo_st["keccak_permutations"] / 1e9) / (solo_st["merkle_build_us"] / 1e6),
                            "keccak_frac_of_int_valu_peak": (solo_st["keccak_permutations"] * KECCAK_OPS /
                                                             (solo_st["merkle_build_us"] / 1e6)) / VALU_PEAK_OPS,
                            "eval_ms": solo_st["eval_us"] / 1e3},
                        "gpu_busy_keccak_gperm_per_s": (perms / 1e9) / dt},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nv, prog, trace.num_lookups, args.cpu_sample_cols)
        print(json.dumps(out), flush=True)
    pool.shutdown()
    for l in lanes:
        l.ctx.dev_free(l.d_cols)
        l.ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
