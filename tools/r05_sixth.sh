#!/bin/bash
out=gpurun_out/r05f; mkdir -p $out
python -m pytest tests/test_gpu_round5.py -m gpu -x -q -s -k "shared_trail" > $out/stp_test.log 2>&1; echo "stp test rc=$?"; grep -E "mean window|passed|failed|Error|assert" $out/stp_test.log | head
RBS_STP_ENTER=0.0 RBS_STP_EVERY=2 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_reference_semantics.py tests/test_tracker.py -m gpu -q -x > $out/stp_suite.log 2>&1; echo "suite under forced shared trail rc=$?"; tail -5 $out/stp_suite.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "vga_long" > $out/vga.log 2>&1; echo "vga rc=$?"; grep -E "120 frames|passed|failed" $out/vga.log
( time python bench.py --sweep-only ) > $out/sweep.json 2> $out/sweep.err; echo "sweep rc=$?"; cat $out/sweep.json; tail -3 $out/sweep.err
bash tools/ab_lib.sh "--config c1" base base:RBS_SHARED_TRAIL=0 > $out/ab.log 2>&1; cat $out/ab.log
