set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fin_* 
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_b1 -- python3 bench.py --steps 5 --warmup 1 --batch 1 --no-cpu-baseline > gpurun_out/fin_b1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_bdef -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/fin_bdef.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/fin_pmc_FETCH_SIZE -- python3 bench.py --steps 2 --warmup 1 --batch 1 --no-cpu-baseline > gpurun_out/fin_pmc_FETCH_SIZE.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/fin_pmc_WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --batch 1 --no-cpu-baseline > gpurun_out/fin_pmc_WRITE_SIZE.log 2>&1
python bench.py > gpurun_out/fin_bench.json 2> gpurun_out/fin_bench.err
python bench.py --batch 1 --no-cpu-baseline > gpurun_out/fin_bench_b1.json 2>> gpurun_out/fin_bench.err
python bench.py --batch 6 --no-cpu-baseline > gpurun_out/fin_bench_b6.json 2>> gpurun_out/fin_bench.err
python bench.py --dedup --batch 8 --no-cpu-baseline > gpurun_out/fin_bench_dedup8.json 2>> gpurun_out/fin_bench.err
python tools/measure_extra.py > gpurun_out/fin_extra.json 2>> gpurun_out/fin_bench.err
hipcc --offload-arch=gfx950 -O3 -I zigz_amd/csrc -I include tools/merkle_rate.hip -o /tmp/merkle_rate 2>/dev/null && /tmp/merkle_rate > gpurun_out/fin_merkle_rate.txt
hipcc --offload-arch=gfx950 -O3 -I zigz_amd/csrc tools/fold_rate.hip -o /tmp/fold_rate 2>/dev/null && /tmp/fold_rate > gpurun_out/fin_fold_rate.txt
hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate 2>/dev/null && /tmp/valu_rate > gpurun_out/fin_valu_rate.txt
hipcc --offload-arch=gfx950 -O3 tools/bank_rate.hip -o /tmp/bank_rate 2>/dev/null && /tmp/bank_rate > gpurun_out/fin_bank_rate.txt
python3 tools/gen_valu2_rate.py && hipcc --offload-arch=gfx950 -O3 -I tools tools/valu2_rate.hip -o /tmp/valu2_rate 2>/dev/null && /tmp/valu2_rate > gpurun_out/fin_valu2_rate.txt
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/fin_bench_b8.json 2>> gpurun_out/fin_bench.err
rm -f gpurun_out/fin_configs.jsonl
for c in 2 3 4 5; do python tests/run_config.py --config $c --check-cols 1 >> gpurun_out/fin_configs.jsonl 2>> gpurun_out/fin_bench.err; done
tail -c 600 gpurun_out/fin_bench.json
