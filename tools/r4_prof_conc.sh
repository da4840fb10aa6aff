#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r4s; mkdir -p $out
prof() { name=$1; shift
  echo "start $name"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras "$@" > $out/$name.json 2> $out/$name.err || { echo "$name: profiler run failed"; tail -3 $out/$name.err | cut -c1-200; return 1; }
  T=$(ls $out/$name/*/*kernel_trace.csv | head -1)
  echo "trace $T"
  python3 tools/trace_concurrency.py $T --json $out/${name}_conc.json > /dev/null
  rm -rf $out/$name
  python3 -c "import json; d=json.load(open('$out/$name.json')); print('$name', d['value']/1e6)"
}
prof base && ZIGZ_BENCH_BATCH_NV=20 ZIGZ_BENCH_BATCH_MAX=3 prof b3s8 --slots 8
