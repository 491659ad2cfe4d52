#!/bin/bash
# tools/ab_host.sh "ENV=.. ENV2=.." ... : the host-pointer leg of bench.py under each environment
for spec in "$@"; do
  env $spec python bench.py --no-cpu-baseline --no-dense-leg --no-f64-leg --no-pmc --no-tracker-fps --steps 200 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$spec]', 'resident=%.0f host_api=%.0f staged=%.0f loglikes_only=%.0f' % (d['value'], d['host_api_value'], d['host_api_staged_frame_value'], d['host_api_loglikes_only_value']))
"
done
