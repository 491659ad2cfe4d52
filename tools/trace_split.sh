#!/bin/bash
# tools/trace_split.sh "<env>" : kernel start/end timeline (us) of the last steps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt; env $1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/kt.log 2>&1
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/kt/kt_kernel_trace.csv"))]
rows=[r for r in rows if "rbs_" in r["Kernel_Name"] and "fill" not in r["Kernel_Name"] and "aux" not in r["Kernel_Name"] and "render" not in r["Kernel_Name"]]
t0=min(int(r["Start_Timestamp"]) for r in rows)
for r in rows[-6:]:
    print(r["Kernel_Name"][10:40], "start %.1f end %.1f dur %.1f us"%((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3), "vgpr", r.get("VGPR_Count"), "lds", r.get("LDS_Block_Size"), "scratch", r.get("Scratch_Size"))
PY
