bash tools/profile_round.sh r01 > gpurun_out/prof_r01.log 2>&1
RBS_STATE=dense bash tools/profile_round.sh r01_dense > gpurun_out/prof_r01_dense.log 2>&1
bash tools/sq_profile.sh r01 > gpurun_out/sq_r01.log 2>&1
timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json
run() { echo "== $*"; python bench.py --no-cpu-baseline --no-dense-leg "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('val=%.0f step_ms=%.4f raster_ms=%.4f copy_ms=%.4f win=%.3f frac=%.3f'%(d['value'],d['ms_per_step'],r['raster_kernel_ms'],r['copy_kernel_ms'],r['stored_window_fraction_of_plane'],r['frac']))
"; }
run
run --update 0
run --sequence 0
run --mesh m1,m2,m3 --particles 6666 --steps 50
run --particles 25000 --steps 20 --warmup 3
run --mesh m4 --cols 1280 --rows 960 --particles 6250 --steps 5 --warmup 2
run --cols 80 --rows 60
RBS_STATE=dense python bench.py --no-cpu-baseline --no-dense-leg 2>/dev/null | tail -1 | cut -c1-300
timeout 900 python tools/tracker_fps.py 2>&1 | grep -o '"filter": "[a-z]*", "evaluation_count": [0-9]*\|"value": [0-9.]*\|sensor_device_ms_per_frame": [0-9.]*' | paste - - - 
timeout 600 python tools/tracker_fps.py 20000 m1,m2,m3 2>&1 | grep -o '"filter": "[a-z]*", "evaluation_count": [0-9]*\|"value": [0-9.]*\|sensor_device_ms_per_frame": [0-9.]*' | paste - - - 
timeout 300 python tools/pcie_inclusive.py 2>&1 | tail -3
