#!/bin/bash
# A/B of alternative product builds on one box, spectrum leg of cfg2 (see tools/ab_bench.sh)
for rep in 1 2; do
for lib in ab_libs/libomx_*.so; do
  OMX_HIP_LIB=$PWD/$lib python tools/bench_spectrum_4096.py 2>/dev/null | tail -1
done
done
