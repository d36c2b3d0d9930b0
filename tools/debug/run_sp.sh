cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_matrix.py tests/test_gpu_state_machine.py tests/test_golden.py tests/test_kat_spectrum.py tests/test_kat_spectrogram.py -q -m gpu -k "spectrum or classic or quiet or golden or kat" 2>&1 | tail -4
python tools/bench_spectrum_4096.py 2>/dev/null | tail -2
python tools/bench_classic.py 2>/dev/null | tail -3
for tag in xf8 xf4; do
  echo "== $tag"; OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so python tools/bench_meters.py waveform 1024 2>/dev/null | grep -i "waveform" | cut -c1-200
done
python -m pytest tests/test_gpu_parity_meters.py tests/test_gpu_waveform_forms.py tests/test_gpu_state_machine.py -q -m gpu -k "waveform" 2>&1 | tail -3
