#!/bin/bash
# round 5, first GPU sitting: new entry points, the split launch's parity, A/B of the split against the one-kernel launch
out=gpurun_out/r05a; mkdir -p $out
python -m pytest tests/test_gpu_round5.py -m gpu -x -q > $out/round5_tests.log 2>&1; echo "round5 tests rc=$?"; tail -5 $out/round5_tests.log
RBS_SPLIT=1 RBS_SPLIT_ITEMS_PER_PARTICLE=48 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py tests/test_gpu_reference_semantics.py -m gpu -q -x > $out/split_suite.log 2>&1; echo "split suite rc=$?"; tail -3 $out/split_suite.log
for cfgname in c1 c2 c3_slice; do
  bash tools/ab_lib.sh "--config $cfgname" base base:RBS_SPLIT=1 d128:RBS_SPLIT=1 d160:RBS_SPLIT=1 e6:RBS_SPLIT=1 d160e6:RBS_SPLIT=1 d128e6:RBS_SPLIT=1 base >> $out/ab.log 2>&1
done
RBS_SPLIT_ITEMS_PER_PARTICLE=16 bash tools/ab_lib.sh "--config c4_slice" base base:RBS_SPLIT=1 d160:RBS_SPLIT=1 >> $out/ab.log 2>&1
cat $out/ab.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  RBS_SPLIT=$v rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/kt_split$v -o kt -- python $GRAFT_REPO_ROOT/bench.py --quick --config c1 --steps 200 > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$out/kt_split$v/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== split=$v $f"; head -8 "$f"
done
