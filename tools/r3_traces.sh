#!/bin/bash
# per-kernel summary of the batched commit path on the other traces (gpurun -- 'bash tools/r3_traces.sh TAG [traces...]')
TAG=${1:-traces}; shift
TRACES=${@:-straight mixed round_robin}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in $TRACES; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$t -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_bound_rate.py --lanes 14 --iters 10 --trace $t > $OUT/$t.log 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/tools/summarize_trace.py $(find $OUT/$t -name '*kernel_trace.csv' | head -1) --stats --out $OUT/${t}_stats.csv
  rm -rf $OUT/$t
  tail -1 $OUT/$t.log
done
