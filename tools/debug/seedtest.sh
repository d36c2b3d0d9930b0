# the waveform chunk-parallel soak case on a range of bases: bash tools/debug/seedtest.sh <first> <count>
FIRST=${1:-60000}; COUNT=${2:-60}; fail=0
for b in $(seq $FIRST $((FIRST + COUNT - 1))); do
  out=$(OMX_SOAK_SEED=$b timeout 600 python -m pytest tests/test_gpu_soak.py -q -m gpu -k "waveform_chunk_parallel or ragged_waveform" 2>&1 | grep -E "passed|failed|AssertionError: " | cut -c1-300)
  echo "base $b: $out"; echo "$out" | grep -q failed && fail=$((fail + 1))
done
echo "bases with failures: $fail of $COUNT"
