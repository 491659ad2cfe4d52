"""Summarise one rocprofv3 --pmc pass over tools/valu_bench (one instruction class, waves/SIMD 1..8):
per dispatch (warm-ups skipped) the counters, summed over the chip as rocprofv3 reports them."""
import csv
import sys
from collections import defaultdict

cls, path = sys.argv[1], sys.argv[2]
rows = defaultdict(dict)
for r in csv.DictReader(open(path)):
    rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    rows[int(r["Dispatch_Id"])]["grid"] = int(r["Grid_Size"])
for k, (d, c) in enumerate(sorted(rows.items())):
    if k % 2 == 0:
        continue          # the warm-up launch of each point
    wps = c["grid"] // 256 // 256
    names = [n for n in c if n != "grid"]
    print(f"{cls:14s} waves/SIMD={wps} " + " ".join(f"{n}={c[n]:.4g}" for n in sorted(names)))
