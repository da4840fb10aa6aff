#!/bin/bash
out=gpurun_out/r4ag; mkdir -p $out
cp zigz_amd/lib/libzigz_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in lds6 lds5 lds4; do
    cp tools/bin/ab/hip_$v.so zigz_amd/lib/libzigz_hip.so
    echo "== $v rep $rep: $(python3 tools/gpu_bound_rate.py --lanes 14 --iters 30 --blocking-sync 2>&1 | tail -1)"
  done
done | tee $out/ab_lds.txt
for v in lds6 lds5 lds4; do
  cp tools/bin/ab/hip_$v.so zigz_amd/lib/libzigz_hip.so
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > $out/bench_$v.json 2> $out/bench_$v.err
  echo "== bench $v: $(python3 -c "import json; d=json.load(open('$out/bench_$v.json')); print(d['value']/1e6, d['detail']['level_hash_levels_roofline_leg'])")"
done | tee -a $out/ab_lds.txt
cp /tmp/keep.so zigz_amd/lib/libzigz_hip.so
