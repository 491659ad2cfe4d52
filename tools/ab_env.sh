#!/bin/bash
# tools/ab_env.sh "<bench args>" "ENV=1 ENV2=2" "..." : bench under different environments
args=$1; shift
for e in "$@"; do
  for rep in 1 2; do
    env $e python bench.py --quick $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$e]', '$args', 'step_ms=%.4f'%d['ms_per_step'], 'copy_ms=%.4f'%d['roofline']['kernel_ms'], 'GBps=%.0f'%d['roofline']['achieved'], 'val=%.0f'%d['value'])
"
  done
done
