#!/bin/bash
# tools/reproduce_baseline.sh -- every row of BASELINE.md section 3 on one MI355X (run on the GPU box):
# both likelihood precisions (F64 = the library default and the headline, F32 = opt-in).
row() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-22s %10.0f particle-likelihoods/s  %.4f ms/step  raster %.4f ms  copy %.4f ms  windows %.3f of a plane' % ('$1', d['value'], d['ms_per_step'], r['raster_kernel_ms'], r['copy_kernel_ms'], r.get('stored_window_fraction_of_plane', 0)))
"; }
for p in f64 f32; do
  for c in c1 c1_readonly c2 c3_slice c4_slice default_res; do
    python bench.py --config $c --quick --precision $p 2>/dev/null | row "$c $p"
  done
  python bench.py --quick --layout dense --precision $p 2>/dev/null | row "c1 dense $p"
  python bench.py --quick --particles 200000 --steps 10 --warmup 2 --precision $p 2>/dev/null | row "c3 200000 slabs $p"
done
python tools/tracker_fps.py 200,2000,20000,100000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tracker %-6s filter %6d particles  %8.0f frames/s  sensor %.3f ms/frame' % (d['filter'], d['evaluation_count'], d['value'], d['sensor_device_ms_per_frame']))
"
python tools/tracker_fps.py 20000 m1,m2,m3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tracker C2 %-6s filter %6d evaluations x %d objects  %8.0f frames/s' % (d['filter'], d['evaluation_count'], d['objects'], d['value']))
"
# BASELINE.md section 3a (round 5): the host legs from C++ (host-pointer step, look-ahead, through the plugin surface) and the moving-object legs
python tools/host_legs.py --config c1 2>/dev/null | grep -E '_value|_ms_per_step'
python bench.py --sweep-only 2>/dev/null | grep -E 'sweep_value|sweep_window_fraction\"|tracker(_scalar_background)?_(fps|value|window_fraction)|\"value\"|stored_fraction'
python bench.py --sweep-only --particles 20000 2>/dev/null | grep -E 'tracker(_scalar_background)?_(fps|window_fraction)'
