# same-box A/B of K2's barrier / store-order variants, then the spectrogram and spectrum parity tests on the product build
cd $GRAFT_REPO_ROOT
for rep in 1 2; do bash tools/ab_bench.sh 60; done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_machine.py -q -m gpu -x 2>&1 | tail -4
