#!/bin/bash
# tools/dbg/bimodal.sh [runs]: the default bench as separate processes -- step / raster / copy times beside the addresses of the planes
for i in $(seq ${1:-10}); do
  RBS_BENCH_PRINT_PTRS=1 python bench.py --quick 2> /tmp/bm.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('step_ms=%.4f'%d['ms_per_step'], 'raster_ms=%.4f'%r['raster_kernel_ms'], 'copy_ms=%.4f'%r['copy_kernel_ms'], end=' ')
"; grep "planes at" /tmp/bm.err | head -1
done
