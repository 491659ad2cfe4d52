#!/bin/bash
# occlusion_mode REFERENCE on C1: whole planes against window-sized slabs (packed rows: the ages' cache lines are fully used)
one() { python bench.py --quick --occlusion $1 --slab-px $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{}); print('$1 slab_px $2', 'value %.3f M/s' % (d['value']/1e6), 'ms/step %.4f'%d['ms_per_step'], 'raster_ms', r.get('raster_kernel_ms'), 'copy_ms', r.get('copy_kernel_ms'))
"; }
for i in 1 2; do for s in 0 8192 16384 38400; do one reference $s; done; one device 0; one device 16384; done
