cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "oscilloscope or scope" 2>&1 | tail -15 > gpurun_out/r13_scope_tests.txt
for T in ${SCOPE_TS:-512}; do
export OMX_SCOPE_THREADS=$T
echo "== threads $T" >> gpurun_out/r13_scope_phases1.txt
timeout 300 python tools/scope_phases.py >> gpurun_out/r13_scope_phases1.txt 2>&1
timeout 300 python - <<'PY' >> gpurun_out/r13_scope_bench.txt 2>&1
import sys; sys.argv=['x']
sys.path.insert(0,'tools')
import bench_meters as b
print(b.scope_stereo()['oscilloscope']['ms_per_call'])
PY
done
