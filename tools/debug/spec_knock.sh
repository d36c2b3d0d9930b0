# pricing of the spectrum kernel's parts on one box: product against builds without stores (knock1), without the transform (knock2),
# without the ring loads (knock4); then the kernel-only durations of the product from a kernel trace
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in product knock1 knock2 knock4; do
  if [ $tag = product ]; then unset OMX_HIP_LIB; else export OMX_HIP_LIB=$PWD/ab_libs/libomx_$tag.so; fi
  python tools/bench_spectrum_4096.py 2>/dev/null | tail -1
done
done
unset OMX_HIP_LIB
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/spec_trace -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_spectrum_4096.py > /dev/null 2>&1 )
head -6 gpurun_out/spec_trace/t_kernel_stats.csv | cut -c1-150
