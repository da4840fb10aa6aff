# A/B of a kernel build flag on ONE box: bash tools/ab_build.sh -DFLAG   (interleaved runs of bench.py --kernels)
set -e
FLAG="$1"
python3 -m zigz_amd.build --force > gpurun_out/ab_build0.log 2>&1
cp zigz_amd/lib/libzigz_hip.so /tmp/base.so
ZIGZ_EXTRA_HIPCC_FLAGS="$FLAG" python3 -m zigz_amd.build --force > gpurun_out/ab_build1.log 2>&1
cp zigz_amd/lib/libzigz_hip.so /tmp/flag.so
for rep in 1 2 3; do
  for v in base flag; do
    cp /tmp/$v.so zigz_amd/lib/libzigz_hip.so
    python3 bench.py --kernels --kernel-iters 6 > gpurun_out/ab_$v$rep.json 2>/dev/null
  done
done
cp /tmp/base.so zigz_amd/lib/libzigz_hip.so
python3 - <<'PY'
import json
for v in ("base","flag"):
    for rep in (1,2,3):
        d=json.load(open("gpurun_out/ab_%s%d.json"%(v,rep)))["kernels"]
        print(v, rep, {k.split("[")[0]:(round(x["avg_us"]),round(x["min_us"])) for k,x in d.items() if "keccak" in k})
PY
python3 -m pytest tests/test_gpu_parity.py -x -q -k "merkle or commit_job" 2>&1 | tail -2
