#!/bin/bash
# tools/tracker_timeline.sh [particles] -- kernel + copy timeline of the device tracker's last frames
# (run on the GPU box): where a tracker frame's time goes between the sensor's kernels.
n=${1:-2000}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/tl_$n; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o tl -- python tools/tracker_fps.py $n > $out/run.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/**/tl_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
for f in glob.glob(out + "/**/tl_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
# the device tracker runs first (30 frames): take frames 20..23 of it by locating propagate launches
prop = [i for i, r in enumerate(rows) if "propagate_kernel" in r[2]]
lo, hi = prop[20], prop[24]
t0 = rows[lo][0]
prev_end = None
for s, e, name in rows[lo:hi]:
    print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3, name))
    prev_end = max(e, prev_end or 0)
print("frame period: %.1f us" % ((rows[hi][0] - rows[lo][0]) / 4e3))
with open(out + "/all.txt", "w") as f:   # the whole run, for a look at the other legs (pipelined, host filter)
    for s, e, name in rows:
        f.write("%12.1f %8.1f %s\n" % ((s - rows[0][0]) / 1e3, (e - s) / 1e3, name))
PY
