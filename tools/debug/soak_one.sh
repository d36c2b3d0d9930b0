# re-run one soak case several times (is a red case deterministic?): bash tools/debug/soak_one.sh <base> <test-id substring> [repeats] [lib]
cd $GRAFT_REPO_ROOT
BASE=$1; KEY=$2; N=${3:-5}
for i in $(seq 1 $N); do
  OMX_SOAK_SEED=$BASE python -m pytest tests/test_gpu_soak.py -q -m gpu -k "$KEY" 2>&1 | grep -E "passed|failed|AssertionError" | cut -c1-260 | head -3
done
