#!/bin/bash
# A/B the variants in build_variants/ on the GPU box: tools/ab.sh "<bench args>" v1 v2 ...
args=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = base ]; then unset RBS_LIB_PATH; else export RBS_LIB_PATH=$PWD/build_variants/$v.so; fi
  for rep in 1 2; do
    python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', '$args', 'ms=%.4f'%d['roofline']['kernel_ms'], 'GBps=%.0f'%d['roofline']['achieved'], 'val=%.0f'%d['value'])
"
  done
done
