#!/bin/bash
# A/B of alternative product builds over a list of bench tools: bash tools/ab_tools.sh "tools/bench_sizes.py tools/bench_zp.py"
for lib in ab_libs/libomx_*.so; do
  echo "== $(basename $lib)"
  for t in $1; do OMX_HIP_LIB=$PWD/$lib python $t 2>/dev/null | grep -v "^$"; done
done
