#!/bin/bash
# A/B of the read-only MLE kernels (k_block_sums, k_radix_fold) on ONE box: bench.py --kernels, interleaved, twice
out=gpurun_out/r4j; mkdir -p $out
cp zigz_amd/lib/libzigz_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in ${ZIGZ_AB_VARIANTS:-plain nt8 nt16 nt4}; do
    cp tools/bin/ab/hip_$v.so zigz_amd/lib/libzigz_hip.so
    python3 bench.py --kernels --kernel-iters 10 > $out/k_$v$rep.json 2>/dev/null
    python3 - <<PY
import json
d=json.load(open("$out/k_$v$rep.json"))["kernels"]
print("$v $rep", {k.split("[")[0]+"["+k.split("[")[1][:5]: (round(x["avg_us"],1), round(x["frac"],3)) for k,x in d.items() if "keccak" not in k and "bind" not in k})
PY
  done
done | tee $out/ab_mle.txt
cp /tmp/keep.so zigz_amd/lib/libzigz_hip.so
