#!/bin/bash
out=gpurun_out/r05h; mkdir -p $out
python -m pytest tests/test_gpu_round5.py tests/test_cpp_shim.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
for rep in 1 2 3; do python tools/host_legs.py --config c1 2>&1 | grep "plugin_api_value\|plugin_api_ms\|plugin_api_copying_ms\|host_api_native_ms" ; done | tee $out/host_legs.log
