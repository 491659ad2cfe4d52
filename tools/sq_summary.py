#!/usr/bin/env python3
"""Summarise tools/sq_profile.sh output: mean per-dispatch counter values of one kernel (default: the raster kernel;
argv[2:] = kernel-name substrings, one summary_<substring>.json each).  Counters a pass did not collect are OMITTED,
never printed as zero."""
import csv, glob, json, os, sys
out = sys.argv[1]
names = sys.argv[2:] or ["rbs_raster_kernel"]
rows = []
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
dur = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for name in names:
    acc = {}
    for row in rows:
        if name not in row["Kernel_Name"]:
            continue
        acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    m = {k: sum(sorted(v.values())[len(v) // 4:]) / max(1, len(v) - len(v) // 4) for k, v in acc.items()}
    d = dict(m)
    g = m.get
    if g("SQ_WAVE_CYCLES"):
        for key, c in (("valu_active_fraction_of_wave_cycles", "SQ_ACTIVE_INST_VALU"), ("lds_active_fraction_of_wave_cycles", "SQ_ACTIVE_INST_LDS"),
                       ("vmem_active_fraction_of_wave_cycles", "SQ_ACTIVE_INST_VMEM")):
            if c in m:
                d[key] = m[c] / g("SQ_WAVE_CYCLES")
    if g("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in m:
        # SQ_BUSY_CYCLES counts per SE-quadrant (x 32 of them on this chip); SQ_ACTIVE_INST_VALU in quad-cycles summed over SIMDs:
        # a VALU instruction in flight on the SIMDs, as a fraction of the kernel's busy time x 1 024 SIMDs
        d["valu_busy_frac"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (g("SQ_BUSY_CYCLES") / 32.0 * 1024.0)
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        d["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
    if g("SQ_INSTS_VALU"):
        f64 = [k for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64") if k in m]
        if len(f64) == 4:
            d["f64_share_of_valu"] = sum(m[k] for k in f64) / g("SQ_INSTS_VALU")
    if g("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in m:
        d["fraction_waiting"] = m["SQ_WAIT_ANY"] / g("SQ_WAVE_CYCLES")
    ds = [x for k, v in dur.items() if name in k for x in v]
    if ds:
        ds.sort()
        d["kernel_us_median_under_counters"] = ds[len(ds) // 2]
    tag = "" if names == ["rbs_raster_kernel"] else "_" + name
    json.dump(d, open(os.path.join(out, f"summary{tag}.json"), "w"), indent=1)
    print("==", name)
    for k in sorted(d):
        print("%-44s %.6g" % (k, d[k]))
