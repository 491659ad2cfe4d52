#!/bin/bash
for rep in 1 2; do
for e in 0 100000; do echo "== RBS_TRACKER_SPLIT_MAX=$e"; RBS_TRACKER_SPLIT_MAX=$e python tools/ab_tracker_native.py 200 2000 6000 8000 12000 20000 2>/dev/null | tail -1; done
done
