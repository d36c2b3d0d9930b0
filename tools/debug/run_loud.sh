# per-kernel durations of the cfg3 loudness call (kernel trace) — usage: gpurun -- bash tools/debug/run_loud.sh
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/bench_meters.py loudness 20
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/loud_kt -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py loudness 20 > /dev/null 2>&1
python3 - <<PY
import csv, glob, os
for f in glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/loud_kt/*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "omx::" in r["Name"]:
            print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f}')
PY
cd $GRAFT_REPO_ROOT && python -m pytest tests/test_gpu_parity_meters.py tests/test_gpu_state_machine.py tests/test_gpu_fullsize.py tests/test_exact_f64.py -q -m gpu -k "loudness or cfg3" 2>&1 | tail -8
