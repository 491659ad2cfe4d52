#!/bin/bash
# round 6, VERDICT r5 #6: C2's windowed copy -- the strip list (default) against the walk over the whole region (RBS_COPY_WALK=1)
one() { env $1 python bench.py --quick --config c2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('[$1]', 'value %.3f M/s'%(d['value']/1e6), 'ms/step %.4f'%d['ms_per_step'], 'raster_ms %.4f'%r['raster_kernel_ms'], 'copy_ms %.4f'%r['copy_kernel_ms'])
"; }
for e in "$@"; do one "$e"; one "$e"; done
