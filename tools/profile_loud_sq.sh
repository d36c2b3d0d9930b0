#!/bin/bash
# SQ counters of the loudness bank chunk-parallel kernels (cfg3: 1024 streams x 8 ch x 64 blocks); outputs under gpurun_out/$1
set -u
TAG=${1:-loud_sq}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/tools/bench_meters.py loudness"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_INSTS_SMEM --output-format csv -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU_TRANS --output-format csv -d $OUT/pmc3 -o p -- $CMD > $OUT/pmc3.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py $OUT > /dev/null
grep "loud_\|kernel stats" $OUT/summary.txt
