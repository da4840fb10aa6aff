#!/bin/bash
out=gpurun_out/r4ak; mkdir -p $out
for cfg in "14 1" "8 3" "8 4" "10 4"; do
  set -- $cfg
  echo "== lanes $1 batch $2: $(timeout -k 10 200 python3 tools/gpu_bound_rate.py --lanes $1 --batch $2 --iters 20 --blocking-sync 2>&1 | tail -1)"
done
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json"))
    ph=d["detail"]["host_phase_ms_per_proof"]
    print("$name", "value %.0f M" % (d["value"]/1e6), "lanes", d["config"]["traces_per_step_per_gpu"], "slots", d["config"]["gpu_slots"], "servers", d["config"]["sponge_servers"], "cpu/proof %.2f" % d["detail"]["host_cpu_ms_per_proof"], "busy %.1f" % d["detail"]["host_cpus_busy"], {k: round(v,2) for k,v in ph.items() if v>0.05})
except Exception as e:
    print("$name", "no line", e)
PY
}
for rep in 1 2; do
run base_$rep --steps 20 --warmup 3
ZIGZ_BENCH_BATCH_NV=20 ZIGZ_BENCH_BATCH_LINGER_US=0 ZIGZ_BENCH_BATCH_MAX=4 run b4s8_$rep --steps 20 --warmup 3 --slots 8
ZIGZ_BENCH_BATCH_NV=20 ZIGZ_BENCH_BATCH_LINGER_US=0 ZIGZ_BENCH_BATCH_MAX=4 run b4s8_12_$rep --steps 20 --warmup 3 --slots 8 --sponge-servers 12
ZIGZ_BENCH_BATCH_NV=20 ZIGZ_BENCH_BATCH_LINGER_US=0 ZIGZ_BENCH_BATCH_MAX=3 run b3s10_12_$rep --steps 20 --warmup 3 --slots 10 --sponge-servers 12
done
