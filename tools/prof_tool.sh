#!/bin/bash
# kernel-trace stats of one tool invocation: bash tools/prof_tool.sh <tag> <python tool and args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1; shift
rm -rf $R/gpurun_out/prof_$tag
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o t -- python3 "$@" > $R/gpurun_out/prof_$tag.log 2>&1 < /dev/null
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then grep "omx::" "$f" | cut -d, -f1-7 | cut -c1-170 | head -30; else echo "no stats"; tail -5 $R/gpurun_out/prof_$tag.log; fi
