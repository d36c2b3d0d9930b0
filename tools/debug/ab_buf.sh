# same-box A/B: buffer descriptors as the compiler sees them (bufold: waterfall loops around every ring load of the spectrum kernels,
# the non-Hann tri kernels, the 8192 / 16384 kernels) against descriptors pinned to SGPRs (the product build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in bufold product; do
  if [ $tag = product ]; then unset OMX_HIP_LIB; else export OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so; fi
  echo "== $tag"; python tools/bench_spectrum_4096.py 2>/dev/null | tail -1
  python tools/bench_spectrum_sizes.py 2>&1 | grep spectrum
  python tools/bench_sizes.py 2>&1 | grep "W=8192\|W=16384"
  python tools/bench_windows.py 2>&1 | grep 4096
done
done
unset OMX_HIP_LIB
python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_machine.py -q -m gpu -x 2>&1 | tail -3
