#!/bin/bash
# chunk length of the waveform bank's chunk-parallel form (tuning build): bash tools/ab_wave_items.sh <items...>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
export OMX_HIP_LIB=$R/openmeters_amd/csrc/libomx_hip_tuning.so
for pass in 1 2; do
for it in "$@"; do
  echo "== items $it"; OMX_WAVE_CHUNK_ITEMS=$it timeout 200 python3 $R/tools/bench_meters.py waveform 1024 2>&1 | grep "waveform:"
done; done
