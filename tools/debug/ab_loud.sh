# A/B of ab_libs/libomx_*.so on the cfg3 loudness call — usage: gpurun -- bash tools/debug/ab_loud.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in ab_libs/libomx_*.so; do
  echo "== $lib"; OMX_HIP_LIB=$PWD/$lib python tools/bench_meters.py loudness 30 2>/dev/null | head -1 | cut -c1-120
done
done
python -m pytest tests/test_gpu_parity_meters.py tests/test_gpu_state_machine.py tests/test_gpu_fullsize.py tests/test_exact_f64.py -q -m gpu -k "loudness or cfg3" 2>&1 | tail -4
