#!/bin/bash
out=gpurun_out/r4h; mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json"))
    ph=d["detail"]["host_phase_ms_per_proof"]
    print("$name", "value %.0f M" % (d["value"]/1e6), "lanes", d["config"]["traces_per_step_per_gpu"], "slots", d["config"]["gpu_slots"], "cpu/proof %.2f" % d["detail"]["host_cpu_ms_per_proof"], "busy %.1f" % d["detail"]["host_cpus_busy"], {k: round(v,2) for k,v in ph.items() if v>0.05}, "perms %.1f M" % (d["config"]["keccak_permutations_per_proof"]/1e6))
except Exception as e:
    print("$name", "no line", e)
PY
}
run straight_s12 --trace straight --steps 8 --warmup 3
run straight_s0 --trace straight --steps 8 --warmup 3 --slots 0
run straight_s24 --trace straight --steps 8 --warmup 3 --slots 24
run straight_s6 --trace straight --steps 8 --warmup 3 --slots 6
run worst_s12 --trace worst --steps 8 --warmup 3
run worst_s0 --trace worst --steps 8 --warmup 3 --slots 0
