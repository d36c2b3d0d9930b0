# same-box A/B: the chunk-parallel meter kernels with __syncthreads() (chsync: the previous commit's loudness / stereometer / waveform
# chunked sources) against LDS-only barriers (product) — the barrier sits right behind the next tile's prefetch
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for tag in chsync product; do
  if [ $tag = product ]; then unset OMX_HIP_LIB; else export OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so; fi
  echo "== $tag"; python tools/bench_meters.py loudness 2>&1 | grep "cfg3" | cut -c1-200
  python tools/bench_meters.py stereometer 2>&1 | grep "cfg4" | cut -c1-120
  python tools/bench_meters.py waveform 1024 2>&1 | grep "waveform" | cut -c1-120
done
done
unset OMX_HIP_LIB
python -m pytest tests -q -m gpu -x -k "loudness or stereometer or waveform or chunk" 2>&1 | tail -4
