#!/bin/bash
# A/B of the two occlusion modes on one GPU (bench.py --quick): C1 twice ("all": + the other configurations);
# further arguments: build_variants/<name>.so to run the reference mode with as well.
one() {   # mode, bench args, library
env ${3:+RBS_LIB_PATH=$PWD/build_variants/$3.so} python bench.py --quick $2 --occlusion $1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{}); print('$1 $2 $3', 'value %.3f M/s' % (d['value']/1e6), 'ms/step %.4f'%d['ms_per_step'], 'raster_ms', r.get('raster_kernel_ms'), 'copy_ms', r.get('copy_kernel_ms'))
"
}
all=0; [ "$1" = "all" ] && { all=1; shift; }
for i in 1 2; do for occ in device reference; do one $occ ""; done; done
for v in "$@"; do one reference "" $v; one reference "" $v; done
if [ $all = 1 ]; then for occ in device reference; do for c in c2 c3_slice c4_slice; do one $occ "--config $c"; done; done; fi
