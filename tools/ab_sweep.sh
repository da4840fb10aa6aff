# Sweep of kernel build flags on ONE box:  bash tools/ab_sweep.sh "<flags A>" "<flags B>" ...   ("" = baseline)
# Two interleaved passes over all variants; prints avg / min microseconds of the two Keccak kernels (43 x 2^20, cold).
set -e
i=0
for F in "$@"; do
  ZIGZ_EXTRA_HIPCC_FLAGS="$F" python3 -m zigz_amd.build --force > gpurun_out/sweep_build_$i.log 2>&1
  cp zigz_amd/lib/libzigz_hip.so /tmp/sweep_$i.so
  i=$((i+1))
done
n=$i
for rep in 1 2; do
  i=0
  for F in "$@"; do
    cp /tmp/sweep_$i.so zigz_amd/lib/libzigz_hip.so
    python3 bench.py --kernels --kernel-iters 5 > gpurun_out/sweep_${i}_$rep.json 2>/dev/null
    i=$((i+1))
  done
done
cp /tmp/sweep_0.so zigz_amd/lib/libzigz_hip.so
python3 - "$@" <<'PY'
import json, sys
flags = sys.argv[1:]
for i, f in enumerate(flags):
    row = []
    for rep in (1, 2):
        d = json.load(open("gpurun_out/sweep_%d_%d.json" % (i, rep)))["kernels"]
        row.append({k.split("[")[0].replace("k_keccak_", ""): (round(x["avg_us"]), round(x["min_us"])) for k, x in d.items() if "keccak" in k})
    print("%-60s %s" % (f or "(baseline)", row))
PY
