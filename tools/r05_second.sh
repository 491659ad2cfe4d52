#!/bin/bash
# round 5, second GPU sitting: fixed tests, the self-validating multi-rank bench, plugin-surface timing, SQ passes of the split launch
out=gpurun_out/r05b; mkdir -p $out
python -m pytest tests/test_gpu_round5.py -m gpu -x -q > $out/round5_tests.log 2>&1; echo "round5 tests rc=$?"; tail -3 $out/round5_tests.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "two_ranks or cannot_attach" > $out/bench2.log 2>&1; echo "two-rank bench tests rc=$?"; tail -5 $out/bench2.log
python -m pytest tests/test_cpp_shim.py tests/test_gpu_parity.py -m gpu -x -q -k "shim or two_rank_sharding or native_host" > $out/shim.log 2>&1; echo "shim + sharding rc=$?"; tail -3 $out/shim.log
for lib in "" build_variants/d160t.so; do
  echo "== host legs, lib=${lib:-in-tree}"
  RBS_LIB_PATH=${lib:+$PWD/$lib} python tools/host_legs.py --config c1 2>&1 | grep -v "^#" | tee -a $out/host_legs.log
done
bash tools/ab_lib.sh "--config c1" base d160t:RBS_SPLIT=1 d160t >> $out/ab.log 2>&1; cat $out/ab.log
# SQ counters: the one-kernel launch, and both kernels of the split launch (in-tree 4-wave geometry kernel; d160t = 3 waves, no spills)
SQ_KERNELS="rbs_raster_kernel" bash tools/sq_profile.sh r05_mono > $out/sq_mono.txt 2>&1; tail -40 $out/sq_mono.txt
SQ_KERNELS="rbs_depth_kernel rbs_eval_kernel" bash tools/sq_profile.sh r05_split RBS_SPLIT=1 > $out/sq_split.txt 2>&1; tail -90 $out/sq_split.txt
SQ_KERNELS="rbs_depth_kernel rbs_eval_kernel" bash tools/sq_profile.sh r05_split_d160t RBS_SPLIT=1 RBS_LIB_PATH=$PWD/build_variants/d160t.so > $out/sq_split_d160t.txt 2>&1; tail -90 $out/sq_split_d160t.txt
cp gpurun_out/sq_r05_mono/summary*.json $out/ 2>/dev/null
for t in r05_split r05_split_d160t; do for f in gpurun_out/sq_$t/summary_*.json; do cp $f $out/${t}_$(basename $f); done; done
rm -rf gpurun_out/sq_r05_mono/p* gpurun_out/sq_r05_split/p* gpurun_out/sq_r05_split_d160t/p*
