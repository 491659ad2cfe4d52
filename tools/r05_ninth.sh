#!/bin/bash
out=gpurun_out/r05i; mkdir -p $out
python -m pytest tests/test_gpu_round5.py -m gpu -x -q -s -k "shared_trail" > $out/stp_test.log 2>&1; echo "stp tests rc=$?"; grep -E "mean window|shared trail entered|passed|failed|^E " $out/stp_test.log | head
RBS_STP_ENTER=0.0 RBS_STP_EVERY=2 timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_parity.py tests/test_tracker.py -m gpu -q -x > $out/stp_suite.log 2>&1; echo "slab + parity suites under forced shared trail rc=$?"; tail -5 $out/stp_suite.log
python -m pytest tests/test_gpu_slabs.py tests/test_gpu_peers.py tests/test_gpu_multidevice.py -m gpu -q -x > $out/plain.log 2>&1; echo "slabs/peers/multidevice (default) rc=$?"; tail -3 $out/plain.log
