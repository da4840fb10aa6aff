# Round-2 profile run: everything profiles/r02_* comes from (one gpurun call:  gpurun --timeout 1200 -- 'bash tools/regen_profiles.sh').
# Under rocprofv3 the program itself follows `--` and bench.py gets --no-cpu-baseline (its CPU baseline is a child process,
# and a process the profiler has attached to must not start other programs).  PMC counters in passes of their own.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p2_*
K="python3 bench.py --kernels --kernel-iters 10"
# (1) the per-kernel leg: cold-HBM launches of the MLE and Keccak kernels (K1/K2/K3/K4, K5/K6)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2_kernels -- $K > gpurun_out/p2_kernels.json 2> gpurun_out/p2_kernels.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/p2_pmc_FETCH_SIZE -- $K > /dev/null 2> gpurun_out/p2_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/p2_pmc_WRITE_SIZE -- $K > /dev/null 2> gpurun_out/p2_pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 --kernel-trace --output-format csv -d gpurun_out/p2_pmc_VALU -- $K > /dev/null 2> gpurun_out/p2_pmc_valu.err
python3 bench.py --kernels --kernel-iters 10 > gpurun_out/p2_kernels_plain.json 2>> gpurun_out/p2_kernels.err
# (2) whole proofs: one at a time (kernel quality inside a proof) and the default bench workload
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2_b1 -- python3 bench.py --steps 5 --warmup 1 --batch 1 --no-cpu-baseline --no-extras > gpurun_out/p2_b1.json 2> gpurun_out/p2_b1.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2_bdef -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/p2_bdef.json 2> gpurun_out/p2_bdef.err
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p2_one -o t -- python3 tools/trace_one_proof.py > /dev/null 2> gpurun_out/p2_one.err
# (3) Lasso and the real sumcheck
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2_lasso -- python3 bench.py --lasso > gpurun_out/p2_lasso.json 2> gpurun_out/p2_lasso.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2_sumcheck -- python3 tools/measure_extra.py --sumcheck-only > gpurun_out/p2_sumcheck.json 2> gpurun_out/p2_sumcheck.err
# (4) bench lines
python3 bench.py > gpurun_out/p2_bench.json 2> gpurun_out/p2_bench.err
python3 bench.py --batch 1 --sponge-servers 0 --no-cpu-baseline > gpurun_out/p2_bench_b1.json 2>> gpurun_out/p2_bench.err
python3 bench.py --sponge-servers 0 --no-cpu-baseline > gpurun_out/p2_bench_s0.json 2>> gpurun_out/p2_bench.err
python3 bench.py --sponge-servers 0 --batch 8 --no-cpu-baseline > gpurun_out/p2_bench_s0_b8.json 2>> gpurun_out/p2_bench.err
python3 bench.py --sponge-servers 5 --batch 40 --no-cpu-baseline --no-extras > gpurun_out/p2_bench_s4.json 2>> gpurun_out/p2_bench.err
python3 bench.py --sponge-servers 7 --batch 56 --no-cpu-baseline --no-extras > gpurun_out/p2_bench_s6.json 2>> gpurun_out/p2_bench.err
python3 bench.py --merkle tables --sponge-servers 0 --batch 8 --no-cpu-baseline > gpurun_out/p2_bench_tables_b8.json 2>> gpurun_out/p2_bench.err
python3 bench.py --merkle dense --sponge-servers 0 --batch 8 --no-cpu-baseline > gpurun_out/p2_bench_dense_b8.json 2>> gpurun_out/p2_bench.err
python3 bench.py --merkle all --no-cpu-baseline --no-extras > gpurun_out/p2_bench_all.json 2>> gpurun_out/p2_bench.err
ZIGZ_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 --warmup 1 --sponge-servers 2 --batch 12 > gpurun_out/p2_bench_gpus2_rehearsal.json 2>> gpurun_out/p2_bench.err
for t in add_xor mixed round_robin straight; do for h in regs regs+mem all cons; do python3 tools/gpu_bound_rate.py --lanes 14 --trace $t --hint $h; done; done > gpurun_out/p2_gpu_bound.txt 2>> gpurun_out/p2_bench.err
python3 tools/measure_extra.py > gpurun_out/p2_extra.json 2>> gpurun_out/p2_bench.err
# (5) BASELINE configs 2-5 at full size on one GPU
rm -f gpurun_out/p2_configs.jsonl
for c in 2 3 4 5; do python3 tests/run_config.py --config $c --check-cols 1 >> gpurun_out/p2_configs.jsonl 2>> gpurun_out/p2_bench.err; done
tail -c 400 gpurun_out/p2_bench.json
