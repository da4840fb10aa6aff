# A/B of a kernels.hip build flag on ONE box, on whole proofs: bash tools/ab_runs.sh -DFLAG
set -e
FLAG="$1"
python3 -m zigz_amd.build --force > gpurun_out/ab_build0.log 2>&1
cp zigz_amd/lib/libzigz_hip.so /tmp/base.so
ZIGZ_EXTRA_HIPCC_FLAGS="$FLAG" python3 -m zigz_amd.build --force > gpurun_out/ab_build1.log 2>&1
cp zigz_amd/lib/libzigz_hip.so /tmp/flag.so
for rep in 1 2 3; do
  for v in base flag; do
    cp /tmp/$v.so zigz_amd/lib/libzigz_hip.so
    echo "$v $rep: $(python3 tools/gpu_bound_rate.py --lanes 14 --iters 30)"
  done
done
cp /tmp/base.so zigz_amd/lib/libzigz_hip.so
python3 -m pytest tests/test_gpu_parity.py -x -q -k "run_aware" 2>&1 | tail -2
