#!/bin/bash
# tools/ab_lib.sh "<bench args>" variant[:ENV=..,ENV2=..] ... : bench with build_variants/<variant>.so
# ("base" = the in-tree library); prints step / raster / copy kernel times.
args=$1; shift
for spec in "$@"; do
  v=${spec%%:*}; e=""; [[ "$spec" == *:* ]] && e=$(echo "${spec#*:}" | tr ',' ' ')
  lib=""; [ "$v" != base ] && lib="RBS_LIB_PATH=$PWD/build_variants/$v.so"
  env $lib $e python bench.py --quick $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('[$spec]', '$args', 'step_ms=%.4f'%d['ms_per_step'], 'raster_ms=%.4f'%r['raster_kernel_ms'], 'copy_ms=%.4f'%r['copy_kernel_ms'], 'val=%.0f'%d['value'])
"
done
