#!/bin/bash
out=gpurun_out/r05j; mkdir -p $out
python -m pytest tests/test_gpu_round5.py -m gpu -x -q -k "split or borrowed" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/tests.log
bash tools/ab_lib.sh "--config c1" base:RBS_SPLIT=1 noeager:RBS_SPLIT=1 base:RBS_SPLIT=1 noeager:RBS_SPLIT=1 > $out/ab.log 2>&1; cat $out/ab.log
mkdir -p /tmp/v/noeager; cp build_variants/noeager.so /tmp/v/noeager/librbsensor_mi355x.so
for rep in 1 2; do for v in base noeager; do
  if [ $v = base ]; then LP=""; else LP=/tmp/v/$v; fi
  echo "== $v"; LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH python tools/host_legs.py --config c1 2>&1 | grep "plugin_api_ms"
done; done | tee $out/host_legs.log
