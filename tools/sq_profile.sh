#!/bin/bash
# tools/sq_profile.sh TAG [ENV=...]: SQ counters of the raster kernel on the default bench step
# (run on the GPU box; SQ_KERNELS="name1 name2": kernels to summarise, default the raster kernel).  One rocprofv3 pass per counter group; raw csv under gpurun_out/sq_TAG.
tag=${1:-sq}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/sq_$tag; rm -rf $out; mkdir -p $out
BENCH="python bench.py --quick --steps 10 --warmup 3 $BENCH_ARGS"
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64" \
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_WAVES" ; do
  i=$((i+1))
  env "$@" rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o p -- $BENCH > $out/p$i.log 2>&1
done
python tools/sq_summary.py $out $SQ_KERNELS
