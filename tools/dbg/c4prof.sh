name=r04_c4; args="--config c4_slice"
BENCH_ARGS="$args" bash tools/profile_round.sh $name > /dev/null 2>&1
env RBS_PROFILE_N=6250 RBS_PROFILE_NPX=1228800 "RBS_PROFILE_WORKLOAD=bench.py --config c4_slice (C4 per-GPU slice: 6 250 particles, M4 = 50 880 triangles, 1280x960, update=true)" python tools/summarize_profile.py $name > /dev/null 2>&1
BENCH_ARGS="$args" bash tools/sq_profile.sh $name > /dev/null 2>&1
cp gpurun_out/sq_$name/summary.json profiles/${name}_raster_sq_detail.json 2>/dev/null
mkdir -p gpurun_out/c4prof; cp profiles/r04_c4_* gpurun_out/c4prof/
RBS_LIB_PATH=$PWD/build_variants/phase.so python tools/phase_timing.py --update 1 --mesh m4 --particles 3000 --cols 1280 2>&1 | tail -12 > gpurun_out/c4prof/phase_c4.txt
cat gpurun_out/c4prof/phase_c4.txt
