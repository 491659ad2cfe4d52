#!/usr/bin/env python3
"""One line per kernel from the compiler's -Rpass-analysis=kernel-resource-usage report
(dbot_ros_amd/lib/resource_usage.txt, written by csrc/Makefile)."""
import re, subprocess, sys
path = sys.argv[1] if len(sys.argv) > 1 else "dbot_ros_amd/lib/resource_usage.txt"
txt = open(path).read()
keys = [("V", r"VGPRs"), ("S", r"TotalSGPRs"), ("scr", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("sspill", r"SGPRs Spill"), ("vspill", r"VGPRs Spill"), ("lds", r"LDS Size \[bytes/block\]")]
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn)
    vals = []
    for k, pat in keys:
        m = re.search(r" " + pat + r": (\S+)", b)
        vals.append(f"{k}={m.group(1) if m else '?'}")
    print(f"{dn:70s} " + " ".join(vals))
