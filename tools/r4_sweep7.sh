#!/bin/bash
out=gpurun_out/r4o; mkdir -p $out
run() { name=$1; shift; timeout -k 10 500 python bench.py --no-extras --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json"))
    ph=d["detail"]["host_phase_ms_per_proof"]
    print("$name", "value %.0f M" % (d["value"]/1e6), "lanes", d["config"]["traces_per_step_per_gpu"], "slots", d["config"]["gpu_slots"], "cpu/proof %.2f" % d["detail"]["host_cpu_ms_per_proof"], "busy %.1f" % d["detail"]["host_cpus_busy"], {k: round(v,2) for k,v in ph.items() if v>0.05})
except Exception as e:
    print("$name", "no line", e)
PY
}
run q4 --steps 20 --warmup 3
GPU_MAX_HW_QUEUES=8 run q8 --steps 20 --warmup 3
GPU_MAX_HW_QUEUES=12 run q12 --steps 20 --warmup 3
GPU_MAX_HW_QUEUES=16 run q16 --steps 20 --warmup 3
GPU_MAX_HW_QUEUES=2 run q2 --steps 20 --warmup 3
run q4b --steps 20 --warmup 3
