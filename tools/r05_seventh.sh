#!/bin/bash
out=gpurun_out/r05g; mkdir -p $out
python -m pytest tests/test_gpu_round5.py -m gpu -x -q -s -k "shared_trail" > $out/stp_test.log 2>&1; echo "stp tests rc=$?"; grep -E "mean window|shared trail entered|passed|failed|Error" $out/stp_test.log | head
RBS_STP_ENTER=0.0 RBS_STP_EVERY=2 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_reference_semantics.py tests/test_tracker.py tests/test_gpu_differential.py -m gpu -q -x > $out/stp_suite.log 2>&1; echo "suite under forced shared trail rc=$?"; tail -5 $out/stp_suite.log
( time python bench.py --sweep-only ) > $out/sweep.json 2> $out/sweep.err; echo "sweep rc=$?"; grep -E "sweep_value|sweep_window_fraction\"|sweep_tracker_fps|sweep_tracker_value|tracker_window|scalar_background_fps|\"value\"|stored_fraction" $out/sweep.json
