# same-box A/B of the spectrum kernel's round-5 steps: bufold (waterfall loops), bufnew (descriptors pinned), product (packed hop pairs,
# one reduction round, equalisation riding the DC removal, DPP reductions)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in ${TAGS:-bufold bufnew product}; do
  if [ $tag = product ]; then unset OMX_HIP_LIB; else export OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so; fi
  echo "== $tag"; python tools/bench_spectrum_4096.py 2>/dev/null | tail -1
  python tools/bench_spectrum_sizes.py 2>&1 | grep spectrum
done
done
unset OMX_HIP_LIB
python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_machine.py tests/test_gpu_pipeline.py tests/test_gpu_capture_chunks.py -q -m gpu -x -k "spectrum or capture or group" 2>&1 | tail -8
