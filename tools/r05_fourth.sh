#!/bin/bash
out=gpurun_out/r05d; mkdir -p $out
python -m pytest tests/test_gpu_round5.py -m gpu -x -q > $out/round5_tests.log 2>&1; echo "round5 tests rc=$?"; tail -3 $out/round5_tests.log
for v in base u3 u3w4 w4 w6; do
  if [ $v = base ]; then LP=""; else mkdir -p /tmp/v/$v; cp build_variants/$v.so /tmp/v/$v/librbsensor_mi355x.so; LP=/tmp/v/$v; fi
  echo "== $v"; LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH python tools/host_legs.py --config c1 2>&1 | grep "plugin_api_value\|plugin_api_ms\|plugin_api_copying_ms\|host_api_native_ms" | tee -a $out/host_legs.log
done
bash tools/ab_lib.sh "--config c1" base:RBS_SPLIT=1 u3:RBS_SPLIT=1 u3w4:RBS_SPLIT=1 w4:RBS_SPLIT=1 w6:RBS_SPLIT=1 base > $out/ab.log 2>&1; cat $out/ab.log
