#!/bin/bash
# tools/trace_split.sh "<env>" : kernel start/end timeline (us) of the last two calls
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt; env $1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --quick --steps 6 --warmup 2 > gpurun_out/kt.log 2>&1
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/kt/kt_kernel_trace.csv"))]
rows=[r for r in rows if "rbs" in r["Kernel_Name"] and "fill" not in r["Kernel_Name"] and "aux" not in r["Kernel_Name"] and "render" not in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-10:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    n=r["Kernel_Name"]; n=n[n.index("rbs_"):][:24]
    print("%-26s start %8.1f end %8.1f dur %7.1f us"%(n,(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
