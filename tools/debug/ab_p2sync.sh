# same-box A/B: __syncthreads() (p2sync: stft_pow2_kernels.hip of the previous commit) against LDS-only barriers in the pow2 spectrogram kernels
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in p2sync product; do
  if [ $tag = product ]; then unset OMX_HIP_LIB; else export OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so; fi
  echo "== $tag"
  python tools/bench_sizes.py 2>&1 | grep "W="
  python tools/bench_zp.py 2>&1 | grep "W="
  python tools/bench_stream.py --calls 150 2>&1 | grep captures
  python tools/bench_spectrum_sizes.py 2>&1 | grep spectrum
done
done
unset OMX_HIP_LIB
python -m pytest tests -q -m gpu -x -k "not soak" 2>&1 | tail -5
