#!/bin/bash
# HIP runtime calls of one tool invocation, per-call averages: bash tools/prof_hip_api.sh <tag> <python tool and args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1; shift
rm -rf $R/gpurun_out/api_$tag
timeout 400 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $R/gpurun_out/api_$tag -o t -- python3 "$@" > $R/gpurun_out/api_$tag.log 2>&1 < /dev/null
grep "capture group\|us per call" $R/gpurun_out/api_$tag.log
f=$(find $R/gpurun_out/api_$tag -name "*hip_api_stats.csv" | head -1)
[ -n "$f" ] && head -16 "$f" | cut -d, -f1-4,6,7
