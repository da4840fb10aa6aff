#!/bin/bash
# A/B of k_level_hash builds on ONE box: the GPU-only rate of the commit path (tools/gpu_bound_rate.py), interleaved, twice
out=gpurun_out/r4i; mkdir -p $out
cp zigz_amd/lib/libzigz_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in O A B C D; do
    cp tools/bin/ab/hip_$v.so zigz_amd/lib/libzigz_hip.so
    echo "== $v rep $rep: $(python3 tools/gpu_bound_rate.py --lanes 14 --iters 30 --blocking-sync 2>&1 | tail -1)"
  done
done | tee $out/ab_levelhash.txt
cp /tmp/keep.so zigz_amd/lib/libzigz_hip.so
