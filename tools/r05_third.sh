#!/bin/bash
out=gpurun_out/r05c; mkdir -p $out
python -m pytest tests/test_gpu_round5.py -m gpu -x -q > $out/round5_tests.log 2>&1; echo "round5 tests rc=$?"; tail -4 $out/round5_tests.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "two_ranks or cannot_attach" > $out/bench2.log 2>&1; echo "two-rank bench tests rc=$?"; tail -4 $out/bench2.log
python tools/host_legs.py --config c1 2>&1 | grep -v "^#\|amdgpu.ids" | tee $out/host_legs.log
python -m pytest tests/test_cpp_shim.py -m gpu -x -q > $out/shim.log 2>&1; echo "shim rc=$?"; tail -3 $out/shim.log
