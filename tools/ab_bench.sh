#!/bin/bash
# A/B of alternative product builds on one box: every ab_libs/libomx_<tag>.so runs the headline bench (spectrogram only)
# through OMX_HIP_LIB.  usage (on the GPU box): bash tools/ab_bench.sh [steps]
STEPS=${1:-60}
for lib in ab_libs/libomx_*.so; do
  tag=$(basename $lib .so)
  OMX_HIP_LIB=$PWD/$lib python bench.py --steps $STEPS --warmup 5 --no-spectrum --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['value']/1e6,2), 'Mframes/s  kernel_ms', round(d['roofline']['kernel_ms'],4))"
done
