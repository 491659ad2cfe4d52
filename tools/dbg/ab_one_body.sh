#!/bin/bash
# round 6, VERDICT r5 #7: the single-body specialisation of the raster kernel (RBS_ONE_BODY=1), scan unroll 2 / 3 / 4
one() { env $1 python bench.py --quick 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('[$1]', 'value %.3f M/s'%(d['value']/1e6), 'ms/step %.4f'%d['ms_per_step'], 'raster_ms %.4f'%r['raster_kernel_ms'], 'copy_ms %.4f'%r['copy_kernel_ms'])
"; }
for rep in 1 2; do
one "RBS_ONE_BODY=0"
one "RBS_ONE_BODY=1"
one "RBS_ONE_BODY=1 RBS_LIB_PATH=$PWD/build_variants/one_u3.so"
one "RBS_ONE_BODY=1 RBS_LIB_PATH=$PWD/build_variants/one_u4.so"
done
