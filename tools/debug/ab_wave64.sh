# same-box A/B: waveform pass B with the low / mid bands in f32 (wbase) or f64 (wf64, wf64w3 = the same under waves_per_eu 3), then the
# two soak seeds that sat above the three-way bar and the waveform parity tests on the f64 form
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in ${TAGS:-wbase wf64 wf64w3}; do
  echo "== $tag"; OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so python tools/bench_meters.py waveform 1024 2>/dev/null | grep -i "waveform" | cut -c1-220
done
done
export OMX_HIP_LIB=$PWD/ab_libs/libomx_${FINAL:-wf64}.so
python - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import conftest, pytest
sys.exit(pytest.main(["-q", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_gpu_parity_meters.py", "tests/test_gpu_waveform_forms.py", "-k", "waveform", "-x"]))
PY
python tools/debug/wave_seeds.py 21051365 21056365 9527360
