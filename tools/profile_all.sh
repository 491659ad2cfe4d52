#!/bin/bash
# tools/profile_all.sh rNN -- on the GPU box: every profile the round commits (C1 default, whole planes, C2, C4 slice: kernel
# stats + HBM PMC + SQ passes each), condensed on the box; the summaries land in gpurun_out/profiles_rNN/ (copy into profiles/).
tag=${1:-r04}
keep=gpurun_out/profiles_$tag; rm -rf $keep; mkdir -p $keep
run() {   # name, bench args, env for the summary
  local name=$1 args=$2; shift 2
  BENCH_ARGS="$args" bash tools/profile_round.sh $name > /dev/null 2>&1
  env "$@" python tools/summarize_profile.py $name > /dev/null 2>&1
  BENCH_ARGS="$args" bash tools/sq_profile.sh $name > /dev/null 2>&1
  cp gpurun_out/sq_$name/summary.json profiles/${name}_raster_sq_detail.json 2>/dev/null
  rm -rf gpurun_out/prof_$name gpurun_out/sq_$name
}
run $tag ""
run ${tag}_dense "--layout dense"
run ${tag}_c2 "--config c2" RBS_PROFILE_N=6666 "RBS_PROFILE_WORKLOAD=bench.py --config c2 (C2: 6 666 particles x meshes M1+M2+M3, 640x480, update=true)"
run ${tag}_c4 "--config c4_slice" RBS_PROFILE_N=6250 RBS_PROFILE_NPX=1228800 "RBS_PROFILE_WORKLOAD=bench.py --config c4_slice (C4 per-GPU slice: 6 250 particles, M4 = 50 880 triangles, 1280x960, update=true)"
cp profiles/${tag}_* profiles/pmc_traffic.json $keep/
ls $keep
