#!/usr/bin/env python3
"""tools/summarize_profile.py rNN -- condense gpurun_out/prof_rNN (written on the GPU box by
tools/profile_round.sh) into the tracked files under profiles/:

  profiles/rNN_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command
  profiles/rNN_pmc_hbm.json       FETCH_SIZE / WRITE_SIZE per dispatch (separate --pmc passes),
                                  gfx950 correction and the calibration run it is based on
  profiles/pmc_traffic.json       {"hbm_bytes_per_launch": ...} read by bench.py for roofline.traffic

Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced streaming
read, so it is doubled; both are cross-checked here against a plain float4 stream copy of a
known byte count (tools/copybench.hip) profiled in the same session.
"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
layout = "dense" if tag.endswith("_dense") else "window"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def per_dispatch(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"], r["Counter_Name"])
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in d.items()}, {k: v[0] for k, v in d.items()}


shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
fetch, nf = per_dispatch(os.path.join(src, "fetch", "fetch_counter_collection.csv"))
write, _ = per_dispatch(os.path.join(src, "write", "write_counter_collection.csv"))
l2, _ = per_dispatch(os.path.join(src, "l2", "l2_counter_collection.csv"))

out = {"tag": tag, "units": "FETCH_SIZE/WRITE_SIZE in KiB per dispatch (average over dispatches)",
       "kernels": {}, "calibration": {}}
KNOWN = 2000 * 640 * 480 * 4  # bytes read == bytes written by one copybench flat_copy launch
# the profiled workload, when it is not the default C1 step (tools/profile_round.sh with BENCH_ARGS="--config c2" ...)
N_PART = int(os.environ.get("RBS_PROFILE_N", "2000"))
NPX = int(os.environ.get("RBS_PROFILE_NPX", str(640 * 480)))
WORKLOAD = os.environ.get("RBS_PROFILE_WORKLOAD", "bench.py default (C1: 2000 particles, 640x480, update=true, 30-frame sequence)")
cal_f = os.path.join(src, "cal_fetch", "cal_counter_collection.csv")
fetch_scale, write_scale = 2.0, 1.0
if os.path.exists(cal_f):
    cf, _ = per_dispatch(cal_f)
    cw, _ = per_dispatch(os.path.join(src, "cal_write", "cal_counter_collection.csv"))
    f = [v for (k, c), v in cf.items() if k.startswith("void flat_copy<8, true>")]
    w = [v for (k, c), v in cw.items() if k.startswith("void flat_copy<8, true>")]
    if f and w:
        fetch_scale = KNOWN / (f[0] * 1024.0)
        write_scale = KNOWN / (w[0] * 1024.0)
        out["calibration"] = {"kernel": "copybench flat_copy<8,nt>, 2.4576e9 B read + 2.4576e9 B written",
                              "FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0],
                              "bytes_per_FETCH_KiB_unit": fetch_scale * 1024, "bytes_per_WRITE_KiB_unit": write_scale * 1024,
                              "fetch_correction": fetch_scale, "write_correction": write_scale}
total = 0.0
for (k, c), v in sorted(fetch.items()):
    if any(t in k for t in ("rbs_copy_kernel", "rbs_copy_rows_kernel", "rbs_copy_window_kernel", "rbs_raster_kernel", "rbs_prep_kernel", "rbs_frame_prep_kernel",
                                "rbs_scan_kernel", "rbs_reduce_kernel", "frame_aux_kernel")):
        wv = write.get((k, "WRITE_SIZE"), 0.0)
        # the x2 fetch correction is calibrated for 16 B/lane streams (the copy kernel); the raster
        # kernel's narrow reads are uncalibrated and reported with the same factor as an upper bound
        b = v * 1024 * fetch_scale + wv * 1024 * write_scale
        hit, miss = l2.get((k, "TCC_HIT_sum")), l2.get((k, "TCC_MISS_sum"))
        out["kernels"][k] = {"dispatches": nf[(k, c)], "FETCH_SIZE_KiB": v, "WRITE_SIZE_KiB": wv,
                             "hbm_bytes_per_dispatch_corrected": b,
                             "l2_hit_rate": (hit / (hit + miss)) if hit is not None and (hit + miss) > 0 else None}
        total += b
out["hbm_bytes_per_loglikes_call"] = total
out["algorithmic_bytes_per_loglikes_call"] = 2.0 * 4.0 * NPX * N_PART
out["traffic_over_algorithmic"] = total / (2.0 * 4.0 * NPX * N_PART)
out["workload"] = WORKLOAD
out["state_layout"] = layout
# roofline.traffic in bench.py is per launch of the dominant kernel: the raster kernel on windowed
# planes, the copy kernel on whole planes
dom = "rbs_raster_kernel" if layout == "window" else "rbs_copy_rows_kernel"
dom_bytes = sum(v["hbm_bytes_per_dispatch_corrected"] for k, v in out["kernels"].items() if dom in k)
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_hbm.json"), "w"), indent=1)
if layout == "window" and "RBS_PROFILE_N" not in os.environ:
    json.dump({"state_layout": layout, "kernel": dom, "hbm_bytes_per_launch": dom_bytes,
               "hbm_bytes_per_loglikes_call_all_kernels": total, "source": f"profiles/{tag}_pmc_hbm.json",
               "workload": "bench.py default (C1: 2000 particles, 640x480, update=true, 30-frame sequence)"},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
sq_path = os.path.join(src, "sq", "sq_counter_collection.csv")
if os.path.exists(sq_path):
    sq, nd = per_dispatch(sq_path)
    k = [kk for (kk, c) in sq if "rbs_raster_kernel" in kk]
    if k:
        k = k[0]
        g = lambda c: sq.get((k, c), 0.0)
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves
        summary = {"precision": os.environ.get("RBS_PROFILE_PRECISION", "f64"), "state_layout": layout, "kernel": k, "workload": WORKLOAD,
                   # (only what this pass collected: a counter that was not in the --pmc list is omitted, not reported as 0)
                   "per_dispatch": {c: g(c) for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                                      "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                      "SQ_BUSY_CYCLES", "SQ_INSTS_SALU") if (k, c) in sq},
                   "valu_instructions_per_particle": g("SQ_INSTS_VALU") / float(N_PART),
                   "cycles_per_valu_instruction": 4.0 * g("SQ_ACTIVE_INST_VALU") / max(g("SQ_INSTS_VALU"), 1.0),
                   "fraction_of_wave_time_issuing_valu": g("SQ_ACTIVE_INST_VALU") / max(g("SQ_WAVE_CYCLES"), 1.0),
                   "fraction_waiting": g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0),
                   "fraction_issue_stalled": g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0)}
        # the instruction mix (its own --pmc pass): what bench.py prices the kernel's issue ceiling with
        mix_path = os.path.join(src, "mix", "mix_counter_collection.csv")
        if os.path.exists(mix_path):
            mx, _ = per_dispatch(mix_path)
            km = [kk for (kk, c) in mx if "rbs_raster_kernel" in kk]
            if km:
                for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"):
                    summary["per_dispatch"][c] = mx.get((km[0], c), 0.0)
                n64 = sum(summary["per_dispatch"][c] for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))
                summary["f64_share_of_valu_instructions"] = n64 / max(g("SQ_INSTS_VALU"), 1.0)
        json.dump(summary, open(os.path.join(dst, f"{tag}_raster_sq.json"), "w"), indent=1)
        print(json.dumps(summary, indent=1))
print(json.dumps(out, indent=1)[:1500])
