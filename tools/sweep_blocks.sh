#!/bin/bash
# within-one-box sweep of the persistent raster grid size (3 interleaved rounds)
for round in 1 2 3; do
  for b in 320 352 384 416 448 480 512 576 640; do
    RBS_RASTER_BLOCKS=$b python bench.py --quick --steps 30 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks=$b round=$round step_ms=%.4f copy_ms=%.4f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
