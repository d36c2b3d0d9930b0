#!/bin/bash
# A/B of loudness bank variants at cfg3: bash tools/ab_loud.sh <lib tags...>  (ab_libs/libomx_<tag>.so; "base" = the product library)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for pass in 1 2; do
for t in "$@"; do
  if [ "$t" = base ]; then unset OMX_HIP_LIB; else export OMX_HIP_LIB=$R/ab_libs/libomx_$t.so; fi
  echo "== $t"; timeout 200 python3 $R/tools/bench_meters.py loudness 2>&1 | grep "cfg3"
done; done
