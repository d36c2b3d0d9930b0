# End-of-round measurement set (run on the GPU box through gpurun): bash tools/round_refresh.sh <tag> <parity tag> <commit>
TAG=${1:-r13}; PTAG=${2:-r03}
export OMX_PROFILE_COMMIT=${3:-unknown}
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.log
python bench.py --config cfg5 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_line_cfg5.json 2> gpurun_out/${TAG}_bench_line_cfg5.log
{
echo "== tools/bench_sizes.py =="; python tools/bench_sizes.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_zp.py =="; python tools/bench_zp.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_spectrum_sizes.py =="; python tools/bench_spectrum_sizes.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_meters.py =="; python tools/bench_meters.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_meters.py waveform, OMX_WAVEFORM_SINGLE=1 (the one-wavefront kernel) =="; OMX_WAVEFORM_SINGLE=1 python tools/bench_meters.py waveform 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_pipeline.py (capture group against separate bank calls) =="; python tools/bench_pipeline.py 2>&1 | grep -v amdgpu.ids
echo "== tools/latency_c.sh (single-stream handles, host in / host out, 256-frame blocks) =="; bash tools/latency_c.sh 2>&1 | tail -7
echo "== tools/determinism_stress.py 16 300 =="; python tools/determinism_stress.py 16 300 2>&1 | tail -1
} > gpurun_out/${TAG}_other_shapes.txt 2>&1
{ echo "# tools/bench_ragged.py: the ragged entry points against the lock-step ones at equal work (every stream gets the same count)"; python tools/bench_ragged.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/${TAG}_ragged.txt
python tools/parity_report.py $PTAG > gpurun_out/${TAG}_parity.log 2>&1
cp profiles/parity_${PTAG}.txt gpurun_out/parity_${PTAG}.txt; tail -3 gpurun_out/${TAG}_parity.log
head -c 600 gpurun_out/${TAG}_bench_line.json
