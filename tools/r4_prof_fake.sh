#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r4x; mkdir -p $out
cp zigz_amd/lib/libzigz_hip.so /tmp/keep.so
for v in real fake; do
  cp tools/bin/ab/hip_$v.so zigz_amd/lib/libzigz_hip.so
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/tr_$v -o t -- python3 tools/trace_one_batch.py 4 2> $out/prof_$v.txt
  python3 - <<PY
import csv,glob
f=glob.glob("$out/tr_$v/**/t_kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
out=open("$out/launches_$v.txt","w")
t0=int(rows[0]["Start_Timestamp"])
for r in rows[-60:]:
    out.write("%10.1f %8.1f  %-40s grid %s\n" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:40], r.get("Grid_Size_X","?")+"x"+r.get("Grid_Size_Z","?")))
PY
  rm -rf $out/tr_$v
  echo "== $v: $(timeout -k 10 200 python3 tools/gpu_bound_rate.py --lanes 14 --iters 20 --blocking-sync 2>&1 | tail -1)"
done
cp /tmp/keep.so zigz_amd/lib/libzigz_hip.so
