#!/bin/bash
# one build -> measure iteration on the GPU box (gpurun -- 'bash tools/r3_iter.sh TAG'): Merkle parity subset, GPU-bound rates,
# a per-kernel summary of the batched commit path and the launches of one lone build.  Results under gpurun_out/TAG/.
TAG=${1:-iter}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prove.py -x -q -k "run_aware or content_addressed or small_domain or merkle or commit or prove" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for t in add_xor mixed round_robin straight; do
  timeout -k 10 200 python3 tools/gpu_bound_rate.py --lanes 14 --iters 20 --trace $t 2>&1 | tee -a $OUT/rate.txt || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/lone -o t -- python3 $GRAFT_REPO_ROOT/tools/trace_one_proof.py > $OUT/lone.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/batch -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_bound_rate.py --lanes 14 --iters 10 > $OUT/batch.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_trace.py $(find $OUT/lone -name '*kernel_trace.csv' | head -1) --last-build --out $OUT/lone_build.csv
python3 tools/summarize_trace.py $(find $OUT/batch -name '*kernel_trace.csv' | head -1) --stats --out $OUT/batch_stats.csv
rm -rf $OUT/lone $OUT/batch
tail -1 $OUT/batch.log
