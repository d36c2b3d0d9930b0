#!/bin/bash
# kernel-trace stats of the bench step for two library builds on one box: bash tools/prof_ab.sh <libA> <libB>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for lib in "$@"; do
  tag=$(basename $lib .so)
  export OMX_HIP_LIB=$R/$lib
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o t -- python3 $R/bench.py --steps 20 --no-secondary --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"
  if [ -n "$f" ]; then head -8 "$f" | cut -c1-220; else echo "no kernel_stats.csv"; fi
done
