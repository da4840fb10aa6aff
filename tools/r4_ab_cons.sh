#!/bin/bash
# A/B of the table passes' workgroup size on ONE box: GPU-only rate of the commit path, interleaved, twice; then the bench line
out=gpurun_out/r4t; mkdir -p $out
cp zigz_amd/lib/libzigz_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in T1024 T512 T256 T128; do
    cp tools/bin/ab/hip_$v.so zigz_amd/lib/libzigz_hip.so
    echo "== $v rep $rep: $(python3 tools/gpu_bound_rate.py --lanes 14 --iters 30 --blocking-sync 2>&1 | tail -1)"
  done
done | tee $out/ab_cons.txt
for v in T1024 T256 T1024 T256; do
  cp tools/bin/ab/hip_$v.so zigz_amd/lib/libzigz_hip.so
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > $out/bench_$v.json 2> $out/bench_$v.err
  echo "== bench $v: $(python3 -c "import json; d=json.load(open('$out/bench_$v.json')); print(d['value']/1e6)")"
done | tee -a $out/ab_cons.txt
cp /tmp/keep.so zigz_amd/lib/libzigz_hip.so
