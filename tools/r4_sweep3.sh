#!/bin/bash
out=gpurun_out/r4e; mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json"))
    ph=d["detail"]["host_phase_ms_per_proof"]
    print("$name", "value %.0f M" % (d["value"]/1e6), "lanes", d["config"]["traces_per_step_per_gpu"], "slots", d["config"]["gpu_slots"], "cpu/proof %.2f" % d["detail"]["host_cpu_ms_per_proof"], "busy %.1f" % d["detail"]["host_cpus_busy"], {k: round(v,2) for k,v in ph.items() if v>0.05})
except Exception as e:
    print("$name", "no line", e)
PY
}
run nv16_b16 --nv 16 --slots 6 --steps 40 --warmup 6
ZIGZ_BENCH_BATCH_MAX=32 ZIGZ_BENCH_BATCH_LINGER_US=400 run nv16_b32 --nv 16 --slots 4 --steps 40 --warmup 6
ZIGZ_BENCH_BATCH_MAX=16 run nv16_b16_l160 --nv 16 --slots 6 --batch 160 --steps 40 --warmup 6
ZIGZ_BENCH_BATCH_MAX=16 run nv16_b16_l128_s8 --nv 16 --slots 8 --batch 128 --steps 40 --warmup 6
ZIGZ_BENCH_BATCH_MAX=8 run nv16_b8_l128_s12 --nv 16 --slots 12 --batch 128 --steps 40 --warmup 6
run nv12_b16 --nv 12 --slots 6 --steps 40 --warmup 6
run nv17_b16 --nv 17 --slots 6 --steps 20 --warmup 4
run nv14_b16 --nv 14 --slots 6 --steps 40 --warmup 6
