#!/bin/bash
# A/B the compile-time variants of the fused STFT kernel (run on the GPU box)
for v in ${VARIANTS:-0 12 100 1 2 3 13}; do
  OMX_K2_VARIANT=$v python bench.py --steps 10 --warmup 2 --no-spectrum --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', round(d['value']/1e6,2), 'Mframes/s kernel_ms', round(d['roofline']['kernel_ms'],3))"
done
