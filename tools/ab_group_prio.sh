#!/bin/bash
# side-stream priorities of the capture group (tuning build): bash tools/ab_group_prio.sh "0,0,0,0" "0,0,1,0" ...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
export OMX_HIP_LIB=$R/openmeters_amd/csrc/libomx_hip_tuning.so
for pass in 1 2; do
for pr in "$@"; do
  echo "== prio $pr"
  OMX_GROUP_PRIO=$pr python3 $R/tools/bench_group_ragged.py lock 400 | cut -c1-60
  OMX_GROUP_PRIO=$pr python3 $R/tools/bench_group_ragged.py ragged 400 | cut -c1-60
done; done
