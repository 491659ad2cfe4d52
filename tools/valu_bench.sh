#!/bin/bash
# tools/valu_bench.sh [round-tag] -- the VALU issue ceilings bench.py prices the raster kernel against:
#   1. tools/valu_bench.hip plainly -> profiles/<tag>_valu_issue_microbench.txt
#   2. the same binary (two classes, fewer iterations) under rocprofv3 --pmc, counters only:
#      SQ_WAVES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU per dispatch
#      -> profiles/<tag>_valu_issue_pmc.csv  (occupancy and clock from the hardware's own counters)
# Run on the GPU box from the repository root.
set -e
tag=${1:-r04}
mkdir -p build_variants gpurun_out profiles
hipcc -O3 --offload-arch=gfx950 -o build_variants/valu_bench tools/valu_bench.hip
./build_variants/valu_bench > gpurun_out/${tag}_valu_issue_microbench.txt
cp gpurun_out/${tag}_valu_issue_microbench.txt profiles/ 2>/dev/null || true
export TMPDIR=/tmp
root=$PWD
for cls in v_fma_f32 v_fma_f64 mix_f64_f32; do
  for pmc in "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
    d=/tmp/vb_$cls_$(echo $pmc | tr ' ' '_')
    rm -rf $d
    (cd /tmp && rocprofv3 --pmc $pmc --kernel-trace -d $d -o vb --output-format csv -- $root/build_variants/valu_bench $cls 6000 > /dev/null 2>&1) || true
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 tools/valu_bench_pmc.py $cls "$f" >> gpurun_out/${tag}_valu_issue_pmc.txt
  done
done
cat gpurun_out/${tag}_valu_issue_pmc.txt
