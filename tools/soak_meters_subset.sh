#!/bin/bash
# the waveform / stereometer / meter-block soak cases on many bases (after a change to the chunk-parallel meter kernels): bash tools/soak_meters_subset.sh <first> <count>
FIRST=${1:-50000}; COUNT=${2:-100}
fail=0
for b in $(seq $FIRST $((FIRST + COUNT - 1))); do
  out=$(OMX_SOAK_SEED=$b python -m pytest tests/test_gpu_soak.py -q -m gpu -k "waveform or stereometer or meter_block" 2>&1 | grep -E "passed|failed|^FAILED|AssertionError" | head -6)
  echo "base $b: $out"
  echo "$out" | grep -q "failed" && fail=$((fail + 1))
done
echo "bases with failures: $fail of $COUNT"
