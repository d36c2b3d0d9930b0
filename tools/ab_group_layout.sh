#!/bin/bash
# side-stream layouts of the capture group at the regular cadence (tuning build): bash tools/ab_group_layout.sh "0,1,1,2" "0,0,1,2" ...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
export OMX_HIP_LIB=$R/openmeters_amd/csrc/libomx_hip_tuning.so
for pass in 1 2; do
for lay in "$@"; do
  echo "== layout $lay"
  OMX_GROUP_LAYOUT=$lay python3 $R/tools/bench_group_ragged.py lock 400 | cut -c1-60
  OMX_GROUP_LAYOUT=$lay python3 $R/tools/bench_group_ragged.py ragged 400 | cut -c1-60
done; done
