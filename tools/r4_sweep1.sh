#!/bin/bash
# round-4 A/B: GPU slots per lane count, legacy form, the ends of the size sweep (phases)
out=gpurun_out/r4c; mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json"))
    ph=d["detail"]["host_phase_ms_per_proof"]
    print("$name", "value %.0f M" % (d["value"]/1e6), "lanes", d["config"]["traces_per_step_per_gpu"], "slots", d["config"]["gpu_slots"], "cpu/proof %.2f" % d["detail"]["host_cpu_ms_per_proof"], "busy %.1f" % d["detail"]["host_cpus_busy"], {k: round(v,2) for k,v in ph.items() if v>0.05}, "frac %.3f busy %.2f" % (d["roofline"]["frac"], d["roofline"].get("busy_share_of_wall") or 0))
except Exception as e:
    print("$name", "no line", e)
PY
}
run s8 --slots 8
run s12 --slots 12
run s24 --slots 24
run s32 --slots 32
run s16_b104 --slots 16 --batch 104
run legacy --slots 0
run nv24_s8 --nv 24 --steps 3 --warmup 1 --slots 8
run nv24_s6_b64 --nv 24 --steps 3 --warmup 1 --slots 6 --batch 64
run nv16_s16 --nv 16 --slots 16
run nv16_s48 --nv 16 --slots 48
run nv16_legacy --nv 16 --slots 0
