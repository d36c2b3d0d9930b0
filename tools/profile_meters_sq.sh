#!/bin/bash
# SQ counters of the cfg3 / cfg4 meter kernels (tools/bench_meters.py); outputs under gpurun_out/$1
set -u
TAG=${1:-meters_sq}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_INSTS_SMEM --output-format csv -d $OUT/pmc2 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH --output-format csv -d $OUT/pmc3 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py > $OUT/pmc3.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py $OUT > /dev/null
grep -v "^==" $OUT/summary.txt | grep "stereometer_kernel\|loudness_roles\|oscilloscope_kernel" 
