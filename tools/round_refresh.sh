# End-of-round measurement set (run on the GPU box through gpurun): bash tools/round_refresh.sh <tag> <parity tag> <commit>
TAG=${1:-r13}; PTAG=${2:-r03}
export OMX_PROFILE_COMMIT=${3:-unknown}
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
# the traffic records first, copied to profiles/ on this box, so that the bench lines below echo records of THIS commit
cp gpurun_out/$TAG/traffic.json profiles/${TAG}_traffic.json
bash tools/profile_meters_pmc.sh ${TAG}_meters > gpurun_out/${TAG}_meters_pmc.txt 2>&1
cp gpurun_out/${TAG}_meters/meters_traffic.json profiles/${TAG}_meters_traffic.json
# ... and under their final names in gpurun_out/, the only directory that travels back from the box: tools/collect_round.sh copies
# gpurun_out/${TAG}_* into profiles/ and REFUSES when one of the two records is missing or carries another commit than the bench lines
# (the records bench.py echoes into roofline.traffic were the ones left behind, twice)
cp gpurun_out/$TAG/traffic.json gpurun_out/${TAG}_traffic.json
cp gpurun_out/${TAG}_meters/meters_traffic.json gpurun_out/${TAG}_meters_traffic.json
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.log
python bench.py --config cfg5 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_line_cfg5.json 2> gpurun_out/${TAG}_bench_line_cfg5.log
{
echo "== tools/bench_sizes.py =="; python tools/bench_sizes.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_zp.py =="; python tools/bench_zp.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_zp_big.py (zero padding beyond 16384 points) =="; python tools/bench_zp_big.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_spectrum_sizes.py =="; python tools/bench_spectrum_sizes.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_meters.py =="; python tools/bench_meters.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_meters.py waveform, OMX_WAVEFORM_SINGLE=1 (the one-wavefront kernel) =="; OMX_WAVEFORM_SINGLE=1 python tools/bench_meters.py waveform 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_meter_forms.py / bench_wave_forms.py (sequential kernels against the chunk-parallel forms over bank and call sizes) =="; python tools/bench_meter_forms.py 2>&1 | grep streams; python tools/bench_wave_forms.py 2>&1 | grep streams
echo "== tools/bench_pipeline.py (capture group against separate bank calls) =="; python tools/bench_pipeline.py 2>&1 | grep -v amdgpu.ids
echo "== tools/latency_c.sh (single-stream handles, host in / host out, 256-frame blocks) =="; bash tools/latency_c.sh 2>&1 | tail -7
echo "== tools/bench_windows.py (4096 / 256 per window kind) =="; python tools/bench_windows.py 2>&1 | grep 4096
echo "== tools/bench_scope_rates.py =="; python tools/bench_scope_rates.py 2>&1 | grep oscillo
echo "== tools/bench_stream.py --each (the reference's cadence: one batcher block per call) =="; python tools/bench_stream.py --each 2>&1 | grep captures
echo "== tools/bench_stream.py --frames 1024 =="; python tools/bench_stream.py --frames 1024 --calls 100 2>&1 | grep captures
echo "== tools/determinism_stress.py 16 300 =="; python tools/determinism_stress.py 16 300 2>&1 | tail -1
} > gpurun_out/${TAG}_other_shapes.txt 2>&1
{ echo "# tools/bench_ragged.py: the ragged entry points against the lock-step ones at equal work (every stream gets the same count)"; python tools/bench_ragged.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/${TAG}_ragged.txt
python tools/parity_report.py $PTAG > gpurun_out/${TAG}_parity.log 2>&1
cp profiles/parity_${PTAG}.txt gpurun_out/parity_${PTAG}.txt; tail -3 gpurun_out/${TAG}_parity.log
head -c 600 gpurun_out/${TAG}_bench_line.json

# kernel trace of the streaming cadence (1024 captures x one 256-frame block per call, six visuals)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_stream -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_stream.py --calls 200 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_stream.log 2>&1 )
cp gpurun_out/${TAG}_stream/t_kernel_stats.csv gpurun_out/${TAG}_stream_kernel_stats.csv 2>/dev/null

# kernel trace of the waveform bank's chunk-parallel form (1024 streams x 16384 frames, history off)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_wave -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py waveform 1024 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_wave.log 2>&1 )
cp gpurun_out/${TAG}_wave/t_kernel_stats.csv gpurun_out/${TAG}_wave_kernel_stats.csv 2>/dev/null

# ... and with RMS history on (round 6: running totals kept between calls)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_wave_history -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py waveform 1024 1 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_wave_history.log 2>&1 )
cp gpurun_out/${TAG}_wave_history/t_kernel_stats.csv gpurun_out/${TAG}_wave_history_kernel_stats.csv 2>/dev/null
