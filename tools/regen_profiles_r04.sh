# Round-4 profile run: everything profiles/r04_* comes from.  Two gpurun calls (each within the 20-minute limit):
#     gpurun --timeout 1200 -- 'bash tools/regen_profiles_r04.sh A'      rocprofv3 passes
#     gpurun --timeout 1200 -- 'bash tools/regen_profiles_r04.sh B'      bench lines, configs, soak
# then `python tools/collect_profiles_r04.py` here.  Under rocprofv3 the program itself follows `--` and bench.py gets
# --no-cpu-baseline (its CPU baseline is a child process, and a process the profiler has attached to must not start other
# programs).  PMC counters in passes of their own, with --kernel-trace only.
set -e
PART=${1:-A}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K="python3 bench.py --kernels --kernel-iters 10"
ONE="python3 tools/trace_one_proof.py"
if [ "$PART" = "A" ]; then
rm -rf gpurun_out/p4_*
# (1) the per-kernel leg: cold-HBM launches of the MLE and Keccak kernels, with the profiler, without it, and under the two
# HBM-traffic counters (separate passes)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_kernels -- $K > gpurun_out/p4_kernels.json 2> gpurun_out/p4_kernels.err
$K > gpurun_out/p4_kernels_plain.json 2>> gpurun_out/p4_kernels.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/p4_kernels_FETCH -o t -- python3 bench.py --kernels --kernel-iters 4 > /dev/null 2>> gpurun_out/p4_kernels.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/p4_kernels_WRITE -o t -- python3 bench.py --kernels --kernel-iters 4 > /dev/null 2>> gpurun_out/p4_kernels.err
echo "[1/4] kernel leg done"
# (2) ONE lone 2^20 proof: every launch in order, and HBM traffic / VALU counters per kernel (separate passes)
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p4_one -o t -- $ONE 2> gpurun_out/p4_one.err | tail -1 > gpurun_out/p4_one_stats.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/p4_one_FETCH -o t -- $ONE > /dev/null 2>> gpurun_out/p4_one.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/p4_one_WRITE -o t -- $ONE > /dev/null 2>> gpurun_out/p4_one.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 --kernel-trace --output-format csv -d gpurun_out/p4_one_VALU -o t -- $ONE > /dev/null 2>> gpurun_out/p4_one.err
echo "[2/4] one-proof passes done"
# (3) whole proofs under the profiler: one at a time, and the default bench workload -- whose own line (events on every launch
# of its roofline leg) is kept beside the profiler's trace of the SAME run: tools/trace_union.py re-derives roofline.frac from it
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_b1 -- python3 bench.py --steps 5 --warmup 1 --batch 1 --sponge-servers 0 --no-cpu-baseline --no-extras > gpurun_out/p4_b1.json 2> gpurun_out/p4_b1.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_bdef -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/p4_bdef.json 2> gpurun_out/p4_bdef.err
T=$(ls gpurun_out/p4_bdef/*/*kernel_trace.csv | head -1)
PPP=$(python3 -c "import json; d=json.load(open('gpurun_out/p4_bdef.json')); print(d['config']['keccak_permutations_per_proof'] - 255*43)")
LAST=$(python3 -c "import json; print(json.load(open('gpurun_out/p4_bdef.json'))['roofline']['launches'])")
python3 tools/trace_union.py $T --perms-per-proof $PPP --json gpurun_out/p4_bdef_union_whole_run.json --intervals gpurun_out/p4_bdef_intervals.csv.gz > /dev/null
python3 tools/trace_concurrency.py $T --json gpurun_out/p4_bdef_concurrency.json > /dev/null  # how many launches at once, and which
python3 tools/trace_union.py gpurun_out/p4_bdef_intervals.csv.gz --perms-per-proof $PPP --last $LAST --json gpurun_out/p4_bdef_union.json > /dev/null
rm -f $T gpurun_out/p4_b1/*/*kernel_trace.csv gpurun_out/p4_kernels/*/*kernel_trace.csv
echo "[3/4] profiled bench runs done"
# (4) Lasso and the real sumcheck
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_lasso -- python3 bench.py --lasso > gpurun_out/p4_lasso.json 2> gpurun_out/p4_lasso.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_sumcheck -- python3 tools/measure_extra.py --sumcheck-only > gpurun_out/p4_sumcheck.json 2> gpurun_out/p4_sumcheck.err
rm -f gpurun_out/p4_lasso/*/*kernel_trace.csv gpurun_out/p4_sumcheck/*/*kernel_trace.csv
./tools/bin/event_semantics > gpurun_out/p4_event_semantics.txt 2>&1 || true
# how the chip schedules launches: chains side by side (4), a hashing wave's priority on full SIMDs, co-running Keccak variants
(timeout -k 10 120 ./tools/bin/stream_concurrency; for q in 2 8 16; do GPU_MAX_HW_QUEUES=$q timeout -k 10 120 ./tools/bin/stream_concurrency; done) > gpurun_out/p4_stream_concurrency.txt 2>&1 || true
timeout -k 10 120 ./tools/bin/prio_probe > gpurun_out/p4_prio_probe.txt 2>&1 || true
(timeout -k 10 120 ./tools/bin/icache_corun; timeout -k 10 120 ./tools/bin/icache_corun 4000 2) > gpurun_out/p4_icache_corun.txt 2>&1 || true
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p4_evsem -o t -- $GRAFT_REPO_ROOT/tools/bin/event_semantics > gpurun_out/p4_event_semantics_profiled.txt 2>/dev/null || true
echo "[4/4] lasso / sumcheck / event semantics done"
python3 -c "import json; print(json.dumps(json.load(open('gpurun_out/p4_bdef_union.json'))['classes']['level_hash']))"
else
# (5) bench lines
python3 bench.py > gpurun_out/p4_bench.json 2> gpurun_out/p4_bench.err
python3 bench.py --slots 0 --no-cpu-baseline --no-extras > gpurun_out/p4_bench_ctx_per_lane.json 2>> gpurun_out/p4_bench.err
python3 bench.py --batch 1 --sponge-servers 0 --no-cpu-baseline --no-extras > gpurun_out/p4_bench_b1.json 2>> gpurun_out/p4_bench.err
python3 bench.py --sponge-servers 0 --no-cpu-baseline --no-extras > gpurun_out/p4_bench_s0.json 2>> gpurun_out/p4_bench.err
echo "[5/7] bench lines done"
for t in add_xor mixed round_robin straight; do python3 tools/gpu_bound_rate.py --lanes 14 --trace $t --phases; done > gpurun_out/p4_gpu_bound.txt 2>> gpurun_out/p4_bench.err
for cfg in "8 2" "8 3" "8 4" "4 8"; do set -- $cfg; python3 tools/gpu_bound_rate.py --lanes $1 --batch $2 --iters 20; done >> gpurun_out/p4_gpu_bound.txt 2>> gpurun_out/p4_bench.err
# the PCIe-inclusive path with the 16-byte and the 32-byte record
python3 bench.py --upload --slots 16 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/p4_bench_upload16.json 2>> gpurun_out/p4_bench.err
ZIGZ_TRACE32=1 python3 bench.py --upload --slots 16 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/p4_bench_upload32.json 2>> gpurun_out/p4_bench.err
# small proofs of 2^20 traces in shared commit jobs (arena form, groups of up to 4 that form while their driver waits for a slot)
ZIGZ_BENCH_BATCH_NV=20 ZIGZ_BENCH_BATCH_LINGER_US=0 ZIGZ_BENCH_BATCH_MAX=4 python3 bench.py --slots 8 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/p4_bench_batch4.json 2>> gpurun_out/p4_bench.err
python3 tools/measure_extra.py > gpurun_out/p4_extra.json 2>> gpurun_out/p4_bench.err
# (6) BASELINE configs 2-5 at full size on one GPU
rm -f gpurun_out/p4_configs.jsonl
for c in 2 3 4 5; do python3 tests/run_config.py --config $c --check-cols 1 >> gpurun_out/p4_configs.jsonl 2>> gpurun_out/p4_bench.err; done
echo "[6/7] configs done"
# (7) soak: random looping RV64IM programs through every build variant, proofs byte-identical to the oracle's
python3 tests/stress_gpu.py --cases 30 --seed 11 --max-log 18 > gpurun_out/p4_stress_soak.log 2>&1 || echo "soak failed"
echo "[7/7] done"
tail -3 gpurun_out/p4_stress_soak.log
tail -c 300 gpurun_out/p4_bench.json
fi
