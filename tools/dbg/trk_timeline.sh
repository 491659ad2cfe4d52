#!/bin/bash
# kernel + copy timeline of two synchronous and two pipelined tracker frames (GPU box): tools/dbg/trk_timeline.sh [particles]
n=${1:-2000}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tt; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tt -o tt -- python tools/dbg/trk_trace.py $n > gpurun_out/tt.log 2>&1; tail -1 gpurun_out/tt.log
python - <<PY
import csv
ev=[]
for r in csv.DictReader(open("gpurun_out/tt/tt_kernel_trace.csv")):
    n=r["Kernel_Name"]
    for key in ("rbs_","rbt"):
        if key in n: n=n[n.index(key):]; break
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n[:40]))
for r in csv.DictReader(open("gpurun_out/tt/tt_memory_copy_trace.csv")):
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"copy "+r["Direction"][12:]))
ev.sort()
idx=[i for i,e in enumerate(ev) if "propagate" in e[2]]
for a,b,lab in ((idx[6],idx[8],"synchronous"),(idx[-4],idx[-2],"pipelined")):
    print(lab); t0=ev[a][0]
    for s,e,n in ev[a:b]:
        print("  %8.1f -> %8.1f (%6.1f)  %s"%((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,n))
PY
