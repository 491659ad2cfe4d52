#!/bin/bash
out=gpurun_out/r05e; mkdir -p $out
( time python -m pytest tests/test_gpu_reference_semantics.py -m gpu -x -q -s -k "c4_geometry or c2_sequence or c1_sequence" ) > $out/lazy.log 2>&1; echo "lazy tests rc=$?"; grep -E "vs LAZY|passed|failed|real" $out/lazy.log
( time python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "vga_long" ) > $out/vga.log 2>&1; echo "vga rc=$?"; tail -4 $out/vga.log
( time python bench.py --sweep-only ) > $out/sweep.json 2> $out/sweep.err; echo "sweep rc=$?"; cat $out/sweep.json; tail -3 $out/sweep.err
