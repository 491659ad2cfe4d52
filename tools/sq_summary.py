#!/usr/bin/env python3
"""Summarise tools/sq_profile.sh output: mean per-dispatch counter values of the raster kernel."""
import csv, glob, json, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "rbs_raster_kernel" not in row["Kernel_Name"]:
            continue
        acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
m = {k: sum(sorted(v.values())[len(v) // 4:]) / max(1, len(v) - len(v) // 4) for k, v in acc.items()}
d = dict(m)
g = m.get
if g("SQ_WAVE_CYCLES"):
    d["valu_active_fraction_of_wave_cycles"] = g("SQ_ACTIVE_INST_VALU", 0) / g("SQ_WAVE_CYCLES")
    d["lds_active_fraction_of_wave_cycles"] = g("SQ_ACTIVE_INST_LDS", 0) / g("SQ_WAVE_CYCLES")
    d["vmem_active_fraction_of_wave_cycles"] = g("SQ_ACTIVE_INST_VMEM", 0) / g("SQ_WAVE_CYCLES")
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    d["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
if g("SQ_INSTS_VALU"):
    d["f64_share_of_valu"] = sum(g(k, 0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")) / g("SQ_INSTS_VALU")
json.dump(d, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k in sorted(d):
    print("%-44s %.6g" % (k, d[k]))
