#!/bin/bash
# A/B of host_bench modes under environment settings: MODES="--plugin --borrowed" plugin_ab.sh "ENV=.. ENV=.." ...   (run on the GPU box)
cd $GRAFT_REPO_ROOT
python tools/dbg/write_workload.py /tmp/wl.bin > gpurun_out/pt_wl.log 2>&1
MODES=${MODES:---plugin}
for rep in 1 2 3 4 5 6 7 8; do
for cfg in "$@"; do
  for mode in $MODES; do
    echo -n "[$cfg] ${mode}: "
    env $cfg tests/cpp/host_bench ${mode#host_api} /tmp/wl.bin 3000 100 | grep host_bench
  done
done
done
