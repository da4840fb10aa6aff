#!/bin/bash
out=gpurun_out/r4p; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_commit_batch_equals_single_jobs" 2>&1 | tail -3
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err || echo "$name failed"; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json"))
    ph=d["detail"]["host_phase_ms_per_proof"]
    print("$name", "value %.0f M" % (d["value"]/1e6), "ok", d["config"].get("self_check_ok"), "lanes", d["config"]["traces_per_step_per_gpu"], "slots", d["config"]["gpu_slots"], "cpu/proof %.2f" % d["detail"]["host_cpu_ms_per_proof"], "busy %.1f" % d["detail"]["host_cpus_busy"], {k: round(v,2) for k,v in ph.items() if v>0.05})
except Exception as e:
    print("$name", "no line", e)
PY
}
export ZIGZ_BENCH_BATCH_NV=20
ZIGZ_BENCH_BATCH_MAX=4 run b4s6 --steps 20 --warmup 3 --slots 6
ZIGZ_BENCH_BATCH_MAX=4 run b4s4 --steps 20 --warmup 3 --slots 4
ZIGZ_BENCH_BATCH_MAX=2 run b2s8 --steps 20 --warmup 3 --slots 8
ZIGZ_BENCH_BATCH_MAX=8 run b8s4 --steps 20 --warmup 3 --slots 4
ZIGZ_BENCH_BATCH_MAX=4 ZIGZ_BENCH_BATCH_LINGER_US=1000 run b4s6l --steps 20 --warmup 3 --slots 6
