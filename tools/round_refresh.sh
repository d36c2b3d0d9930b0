export OMX_PROFILE_COMMIT=7265f51
bash tools/profile_bench.sh r12 > gpurun_out/r12_profile.log 2>&1
python bench.py > gpurun_out/r12_bench_line.json 2> gpurun_out/r12_bench_line.log
python bench.py --config cfg5 --no-cpu-baseline --no-secondary > gpurun_out/r12_bench_line_cfg5.json 2> gpurun_out/r12_bench_line_cfg5.log
{
echo "== tools/bench_sizes.py =="; python tools/bench_sizes.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_zp.py =="; python tools/bench_zp.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_spectrum_sizes.py =="; python tools/bench_spectrum_sizes.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_meters.py =="; python tools/bench_meters.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_meters.py waveform, OMX_WAVEFORM_SINGLE=1 (the one-wavefront kernel) =="; OMX_WAVEFORM_SINGLE=1 python tools/bench_meters.py waveform 2>&1 | grep -v amdgpu.ids
echo "== tools/latency_c.sh (single-stream handles, host in / host out, 256-frame blocks) =="; bash tools/latency_c.sh 2>&1 | tail -7
echo "== tools/microbench/graph_latency =="; tools/microbench/graph_latency 2>&1 | tail -4
echo "== tools/microbench/valu_issue (one wavefront: cycles per VALU instruction) =="; tools/microbench/valu_issue 2>&1 | tail -17
echo "== tools/determinism_stress.py 16 300 =="; python tools/determinism_stress.py 16 300 2>&1 | tail -1
} > gpurun_out/r12_other_shapes.txt 2>&1
python tools/parity_report.py r02 > gpurun_out/r12_parity.log 2>&1
cp profiles/parity_r02.txt gpurun_out/parity_r02.txt; tail -3 gpurun_out/r12_parity.log
cat gpurun_out/r12_bench_line.json | head -c 600
