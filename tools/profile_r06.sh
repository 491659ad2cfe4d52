#!/bin/bash
# tools/profile_r06.sh -- on the GPU box: round 6's committed profiles = tools/profile_all.sh r06 (C1 default, whole planes, C2, C4 slice)
# + the C1 step in occlusion_mode REFERENCE (r06_exact) + the default bench line.
bash tools/profile_all.sh r06
BENCH_ARGS="--occlusion reference" bash tools/profile_round.sh r06_exact > /dev/null 2>&1
RBS_PROFILE_WORKLOAD="bench.py --occlusion reference (C1, rbs_config.occlusion_mode = REFERENCE: stamped planes)" python tools/summarize_profile.py r06_exact > /dev/null 2>&1
BENCH_ARGS="--occlusion reference" bash tools/sq_profile.sh r06_exact > /dev/null 2>&1
cp gpurun_out/sq_r06_exact/summary.json profiles/r06_exact_raster_sq_detail.json 2>/dev/null
rm -rf gpurun_out/prof_r06_exact gpurun_out/sq_r06_exact
cp profiles/r06_exact_* gpurun_out/profiles_r06/ 2>/dev/null
python bench.py > gpurun_out/profiles_r06/r06_bench_default.json 2> gpurun_out/profiles_r06/r06_bench_default.err
ls gpurun_out/profiles_r06
