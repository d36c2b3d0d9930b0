# same-box A/B: 8192 / 16384-point kernels (and the 16384 spectrum kernel's inner transforms) with __syncthreads() inside fft_device.hpp (bigsync)
# against LDS-only barriers (product)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for tag in bigsync product; do
  if [ $tag = product ]; then unset OMX_HIP_LIB; else export OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so; fi
  echo "== $tag"; python tools/bench_sizes.py 2>&1 | grep "W=8192\|W=16384"
  python tools/bench_zp.py 2>&1 | grep "F=16384\|F=8192"
  python tools/bench_spectrum_sizes.py 2>&1 | grep "16384\|8192"
done
done
unset OMX_HIP_LIB
python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_machine.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -4
