#!/bin/bash
out=gpurun_out/r4l; mkdir -p $out
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
run resident
run up_s12 --upload
run up_s16 --upload --slots 16
run up_s24 --upload --slots 24
run up_s16_b104 --upload --slots 16 --batch 104
ZIGZ_TRACE48=1 run up48_s12 --upload
run nv24_b76 --nv 24 --steps 4 --warmup 1
run nv24_s10 --nv 24 --steps 4 --warmup 1 --slots 10
