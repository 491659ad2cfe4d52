#!/bin/bash
# tools/reproduce_baseline.sh -- every row of BASELINE.md section 3 on one MI355X (run on the GPU box)
for c in c1 c1_readonly c2 c3_slice c4_slice default_res; do
  python bench.py --config $c --quick 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-12s %10.0f particle-likelihoods/s  %.4f ms/step  raster %.4f ms  windows %.3f of a plane' % ('$c', d['value'], d['ms_per_step'], r['raster_kernel_ms'], r['stored_window_fraction_of_plane']))
"
done
python bench.py --quick --layout dense 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-12s %10.0f particle-likelihoods/s  %.4f ms/step  copy %.4f ms = %.2f of 8 TB/s' % ('c1 dense', d['value'], d['ms_per_step'], r['copy_kernel_ms'], r['frac']))
"
python tools/tracker_fps.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tracker %-6s filter %6d particles  %8.0f frames/s' % (d['filter'], d['evaluation_count'], d['value']))
"
