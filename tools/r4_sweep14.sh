#!/bin/bash
out=gpurun_out/r4ac; mkdir -p $out
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
run s11_$rep --steps 20 --warmup 3 --sponge-servers 11
ZIGZ_BENCH_BATCH_NV=20 ZIGZ_BENCH_BATCH_LINGER_US=0 ZIGZ_BENCH_BATCH_MAX=4 run s11b4_$rep --steps 20 --warmup 3 --sponge-servers 11 --slots 8
ZIGZ_BENCH_BATCH_NV=20 ZIGZ_BENCH_BATCH_LINGER_US=0 ZIGZ_BENCH_BATCH_MAX=4 run s12b4_$rep --steps 20 --warmup 3 --sponge-servers 12 --slots 8
ZIGZ_BENCH_BATCH_NV=20 ZIGZ_BENCH_BATCH_LINGER_US=0 ZIGZ_BENCH_BATCH_MAX=3 run s12b3_$rep --steps 20 --warmup 3 --sponge-servers 12 --slots 10
done
