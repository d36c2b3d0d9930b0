#!/bin/bash
# same-box A/B of library builds on the headline step: bash tools/ab_round6.sh <lib> <lib> ...   (two passes over the list)
for pass in 1 2; do
for lib in "$@"; do
  echo "== $lib (pass $pass)"
  OMX_HIP_LIB=$PWD/$lib python bench.py --steps 100 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames/s %.4g  ms/step %.4f  K2 ms %.4f' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms']))"
done
done
