#!/usr/bin/env python3
"""Print the kernel timeline (start offset, duration, gap to the previous kernel) of the last
few bench steps from a rocprofv3 --kernel-trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [r for r in rows if "rbs_" in r["Kernel_Name"] or "frame_aux" in r["Kernel_Name"]][-int(sys.argv[2]) if len(sys.argv) > 2 else -16:]
t0 = int(last[0]["Start_Timestamp"]); prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %7.1f us  gap %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3, r["Kernel_Name"][:50]))
    prev_end = e
