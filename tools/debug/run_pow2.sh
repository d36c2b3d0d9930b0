cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_matrix.py tests/test_gpu_state_machine.py -q -m gpu -x -k "reassigned or window or spectrogram or ragged_bank or equivalent" 2>&1 | tail -4
python tools/bench_stream.py --calls 300
python tools/bench_stream.py --calls 300 --frames 1024
