#!/bin/bash
# kernel + copy + HIP-call timeline of one plugin-surface step (host_bench --plugin), run on the GPU box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MODE=${1:---plugin}
python tools/dbg/write_workload.py /tmp/wl.bin > gpurun_out/pt_wl.log 2>&1
tests/cpp/host_bench $MODE /tmp/wl.bin 600 20
tests/cpp/host_bench $MODE /tmp/wl.bin 600 20
rm -rf gpurun_out/pt; rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d gpurun_out/pt -o pt -- tests/cpp/host_bench $MODE /tmp/wl.bin 200 20 > gpurun_out/pt.log 2>&1
tail -2 gpurun_out/pt.log
python - <<'PY'
import csv, glob
ev = []
for r in csv.DictReader(open("gpurun_out/pt/pt_kernel_trace.csv")):
    n = r["Kernel_Name"]; n = n[n.index("rbs_"):][:28] if "rbs_" in n else n[:28]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "  GPU  " + n))
try:
    for r in csv.DictReader(open("gpurun_out/pt/pt_memory_copy_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "  COPY " + r["Direction"] + " " + r.get("Bytes", r.get("Size", "?"))))
except FileNotFoundError:
    pass
for f in glob.glob("gpurun_out/pt/pt_hip_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host " + r["Function"]))
ev.sort()
# a window of two steps, three quarters through the run
ks = [i for i, e in enumerate(ev) if "rbs_eval" in e[2] or "rbs_raster" in e[2]]
k = ks[len(ks) * 3 // 4]
while k > 0 and "hipEventSynchronize" not in ev[k][2]: k -= 1
t0 = ev[k][1]
n = 0
for s, e, name in ev[k:]:
    print("%9.1f -> %9.1f  (%6.1f us)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name))
    n += 1
    if n > 90: break
PY
