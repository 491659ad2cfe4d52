#!/bin/bash
# tools/profile_round.sh rNN  (BENCH_ARGS="--layout dense" tools/profile_round.sh rNN_dense for whole planes;
# BENCH_ARGS="--precision f32" ... rNN_f32 for the opt-in float32 likelihood; the default command is precision F64)
# -- run on the GPU box (gpurun): rocprofv3 kernel stats + HBM PMC
# passes of the default bench command; raw outputs under gpurun_out/prof_<tag>, summaries are
# copied to profiles/ by tools/summarize_profile.py (run locally afterwards).
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
BENCH="python bench.py --quick --steps 20 --warmup 3 $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- $BENCH > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o fetch -- $BENCH > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o write -- $BENCH > $out/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/l2 -o l2 -- $BENCH > $out/l2.log 2>&1
# SQ counters of the same step: where the raster kernel's wave time goes
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $out/sq -o sq -- $BENCH > $out/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d $out/mix -o mix -- $BENCH > $out/mix.log 2>&1
# calibration of the counters on a known byte count (plain float4 stream copy, 2 x 2.4576 GB)
if [ -x build_variants/copybench ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/cal_fetch -o cal -- build_variants/copybench quick > $out/cal_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/cal_write -o cal -- build_variants/copybench quick > $out/cal_write.log 2>&1
fi
find $out -name "*.csv" | head -40
tail -2 $out/stats.log
