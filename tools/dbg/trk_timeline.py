import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last two frames: find last 2 occurrences of propagate_kernel
idx = [i for i, r in enumerate(rows) if "propagate_kernel" in r["Kernel_Name"]]
a = idx[-2]
t0 = int(rows[a]["Start_Timestamp"]); prev = None
for r in rows[a:idx[-1] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us dur %7.1f gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3, r["Kernel_Name"][:60]))
    prev = e
