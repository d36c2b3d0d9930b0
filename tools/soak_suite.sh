#!/bin/bash
# The in-suite soak (tests/test_gpu_soak.py) on several seed bases in a row (run on the GPU box): bash tools/soak_suite.sh <first base> <count>
FIRST=${1:-1000}; COUNT=${2:-5}
for b in $(seq $FIRST $((FIRST + COUNT - 1))); do
  OMX_SOAK_SEED=$b python -m pytest tests/test_gpu_soak.py -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|AssertionError" | head -8 | sed "s/^/base $b: /"
done
