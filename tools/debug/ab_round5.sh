# same-box A/B of the round-5 candidates: spectrum loads (spold / spbuf), waveform exchange granularity (xf8 / xf4)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in spold spbuf; do
  echo "== $tag"; OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so python tools/bench_spectrum_4096.py 2>/dev/null | tail -1
done
for tag in xf8 xf4; do
  echo "== $tag"; OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so python tools/bench_meters.py waveform 1024 2>/dev/null | grep -i "waveform" | cut -c1-200
done
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_meters.py tests/test_gpu_waveform_forms.py tests/test_gpu_state_machine.py -q -m gpu -k "spectrum or waveform" 2>&1 | tail -4
