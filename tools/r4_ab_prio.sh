#!/bin/bash
# A/B of s_setprio in the short kernels on ONE box: GPU-only rate of the commit path, interleaved, twice; then the bench line
out=gpurun_out/r4u; mkdir -p $out
cp zigz_amd/lib/libzigz_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in P0 P1 P3; do
    cp tools/bin/ab/hip_$v.so zigz_amd/lib/libzigz_hip.so
    echo "== $v rep $rep: $(python3 tools/gpu_bound_rate.py --lanes 14 --iters 30 --blocking-sync 2>&1 | tail -1)"
  done
done | tee $out/ab_prio.txt
for v in P0 P3 P0 P3; do
  cp tools/bin/ab/hip_$v.so zigz_amd/lib/libzigz_hip.so
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > $out/bench_$v.json 2> $out/bench_$v.err
  echo "== bench $v: $(python3 -c "import json; d=json.load(open('$out/bench_$v.json')); print(d['value']/1e6)")"
done | tee -a $out/ab_prio.txt
cp /tmp/keep.so zigz_amd/lib/libzigz_hip.so
