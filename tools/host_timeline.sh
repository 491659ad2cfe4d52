#!/bin/bash
# kernel + memory-copy timeline of the host-pointer step (run on the GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ht; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ht -o ht -- python tools/dbg/host_leg_times.py > gpurun_out/ht.log 2>&1
python - <<'PY'
import csv, glob
ev = []
for r in csv.DictReader(open("gpurun_out/ht/ht_kernel_trace.csv")):
    n = r["Kernel_Name"]; n = n[n.index("rbs_"):][:22] if "rbs_" in n else n[:22]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
for r in csv.DictReader(open("gpurun_out/ht/ht_memory_copy_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r["Direction"] + " " + r.get("Bytes", r.get("Size", "?"))))
ev.sort()
# the "copy" leg is the 3rd of 6 legs: take a window two thirds through the run
k = len(ev) * 3 // 6 - 40
t0 = ev[k][0]
for s, e, n in ev[k:k + 28]:
    print("%9.1f -> %9.1f  (%6.1f us)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
