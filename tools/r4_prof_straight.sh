#!/bin/bash
# rocprofv3 kernel stats of the straight-line trace: GPU slots against a context per lane
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in 12 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_s$m -- python3 $R/bench.py --no-extras --no-cpu-baseline --trace straight --steps 6 --warmup 3 --slots $m > $out/prof_s$m.json 2> $out/prof_s$m.err || echo "fail $m"
done
cd $R
for m in 12 0; do
  f=$(ls $out/prof_s$m/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== slots $m ($f)"
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:16]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%10.1f ms" % (float(r["TotalDurationNs"])/1e6), "%8.1f us avg" % (float(r["AverageNs"])/1e3))
PY
done
find $out -name "*kernel_trace.csv" -delete
