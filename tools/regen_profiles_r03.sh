# Round-3 profile run: everything profiles/r03_* comes from (gpurun --timeout 1200 -- 'bash tools/regen_profiles_r03.sh').
# Under rocprofv3 the program itself follows `--` and bench.py gets --no-cpu-baseline (its CPU baseline is a child process,
# and a process the profiler has attached to must not start other programs).  PMC counters in passes of their own.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p3_*
mkdir -p gpurun_out
K="python3 bench.py --kernels --kernel-iters 10"
ONE="python3 tools/trace_one_proof.py"
# (1) the per-kernel leg: cold-HBM launches of the MLE and Keccak kernels
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3_kernels -- $K > gpurun_out/p3_kernels.json 2> gpurun_out/p3_kernels.err
$K > gpurun_out/p3_kernels_plain.json 2>> gpurun_out/p3_kernels.err
echo "[1/6] kernel leg done"
# (2) ONE lone 2^20 proof: every launch in order, and HBM traffic / VALU counters per kernel (separate passes)
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p3_one -o t -- $ONE 2> gpurun_out/p3_one.err | tail -1 > gpurun_out/p3_one_stats.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/p3_one_FETCH -o t -- $ONE > /dev/null 2>> gpurun_out/p3_one.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/p3_one_WRITE -o t -- $ONE > /dev/null 2>> gpurun_out/p3_one.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 --kernel-trace --output-format csv -d gpurun_out/p3_one_VALU -o t -- $ONE > /dev/null 2>> gpurun_out/p3_one.err
echo "[2/6] one-proof passes done"
# (3) whole proofs under the profiler: one at a time, and the default bench workload
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3_b1 -- python3 bench.py --steps 5 --warmup 1 --batch 1 --sponge-servers 0 --no-cpu-baseline --no-extras > gpurun_out/p3_b1.json 2> gpurun_out/p3_b1.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3_bdef -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/p3_bdef.json 2> gpurun_out/p3_bdef.err
echo "[3/6] profiled bench runs done"
# (4) Lasso and the real sumcheck
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3_lasso -- python3 bench.py --lasso > gpurun_out/p3_lasso.json 2> gpurun_out/p3_lasso.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3_sumcheck -- python3 tools/measure_extra.py --sumcheck-only > gpurun_out/p3_sumcheck.json 2> gpurun_out/p3_sumcheck.err
echo "[4/6] lasso / sumcheck done"
# (5) bench lines
python3 bench.py > gpurun_out/p3_bench.json 2> gpurun_out/p3_bench.err
python3 bench.py --batch 1 --sponge-servers 0 --no-cpu-baseline --no-extras > gpurun_out/p3_bench_b1.json 2>> gpurun_out/p3_bench.err
python3 bench.py --sponge-servers 8 --batch 64 --no-cpu-baseline --no-extras > gpurun_out/p3_bench_8x8.json 2>> gpurun_out/p3_bench.err
python3 bench.py --sponge-servers 11 --batch 88 --no-cpu-baseline --no-extras > gpurun_out/p3_bench_11x8.json 2>> gpurun_out/p3_bench.err
python3 bench.py --sponge-servers 0 --no-cpu-baseline --no-extras > gpurun_out/p3_bench_s0.json 2>> gpurun_out/p3_bench.err
echo "[5/6] bench lines done"
for t in add_xor mixed round_robin straight; do python3 tools/gpu_bound_rate.py --lanes 14 --trace $t --phases; done > gpurun_out/p3_gpu_bound.txt 2>> gpurun_out/p3_bench.err
python3 tools/measure_extra.py > gpurun_out/p3_extra.json 2>> gpurun_out/p3_bench.err
# (6) BASELINE configs 2-5 at full size on one GPU
rm -f gpurun_out/p3_configs.jsonl
for c in 2 3 4 5; do python3 tests/run_config.py --config $c --check-cols 1 >> gpurun_out/p3_configs.jsonl 2>> gpurun_out/p3_bench.err; done
echo "[6/6] done"
tail -c 300 gpurun_out/p3_bench.json
