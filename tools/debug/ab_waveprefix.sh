# same-box A/B: waveform prefix kernel with two batches of segment sums in flight (product) against one (wavehead); then smoke() and the waveform tests
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for tag in wavehead product; do
  if [ $tag = product ]; then unset OMX_HIP_LIB; else export OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so; fi
  echo "== $tag"; python tools/bench_meters.py waveform 1024 2>&1 | grep "waveform" | cut -c1-120
done
done
unset OMX_HIP_LIB
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python -m pytest tests -q -m gpu -x -k "waveform" 2>&1 | tail -3
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wavep -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py waveform 1024 0 > /dev/null 2>&1 )
head -9 gpurun_out/wavep/t_kernel_stats.csv | cut -c1-130
