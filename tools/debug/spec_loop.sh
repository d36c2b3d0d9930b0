# same-box A/B: the spectrum kernel's pair loop (p<P>w<W>: P hop pairs per workgroup, register budget for W workgroups per CU)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in ${TAGS:-product p1w4 p8w3 p8w4 p4w3 p16w3}; do
  if [ $tag = product ]; then unset OMX_HIP_LIB; else export OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so; fi
  python tools/bench_spectrum_4096.py 2>/dev/null | tail -1
done
done
export OMX_HIP_LIB=$PWD/ab_libs/libomx_${FINAL:-p8w3}.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_machine.py tests/test_gpu_pipeline.py tests/test_gpu_capture_chunks.py -q -m gpu -x -k "spectrum or capture or group" 2>&1 | tail -4
