#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
python3 tools/trace_one_batch.py 4 2> gpurun_out/r4q/plain.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4q/tr -o t -- python3 tools/trace_one_batch.py 4 2> gpurun_out/r4q/prof.txt
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r4q/tr/**/t_kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
out=open("gpurun_out/r4q/launches.txt","w")
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    out.write("%10.1f %8.1f  %-40s grid %s wg %s\n" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:40], r.get("Grid_Size_X","?")+"x"+r.get("Grid_Size_Z","?"), r.get("Workgroup_Size_X","?")))
PY
rm -rf gpurun_out/r4q/tr
cat gpurun_out/r4q/plain.txt
