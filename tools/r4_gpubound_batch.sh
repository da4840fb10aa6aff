#!/bin/bash
out=gpurun_out/r4w; mkdir -p $out
for rep in 1 2; do
for cfg in "14 1" "8 2" "8 3" "6 4" "4 4" "8 4" "4 8"; do
  set -- $cfg
  echo "== lanes $1 batch $2: $(timeout -k 10 200 python3 tools/gpu_bound_rate.py --lanes $1 --batch $2 --iters 20 --blocking-sync 2>&1 | tail -1)"
done; done | tee $out/gpubound_batch.txt
