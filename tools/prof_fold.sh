#!/bin/bash
# kernel durations of tools/bench_fold.py, one rocprofv3 run per shape (the kernel name does not carry the shape)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for shape in 64,4096,256,1024 64,4096,256,256 64,1024,256,1024 64,4096,1024,256 16,4096,256,1024 256,4096,256,1024 64,16384,1024,256; do
  rm -rf $R/gpurun_out/prof_fold
  OMX_HIP_LIB=${FOLD_LIB:-$R/openmeters_amd/csrc/libomx_hip.so} timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fold -o t -- python3 $R/tools/bench_fold.py $shape > $R/gpurun_out/prof_fold.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/prof_fold -name "*kernel_stats.csv" | head -1)
  echo "== S,W,hop,hops = $shape"
  [ -n "$f" ] && grep window_sums "$f" | cut -d, -f1-7 | cut -c1-140
done
