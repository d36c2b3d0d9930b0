# K2 quick check: parity of the 4096 kernel + the bench line — usage: gpurun -- bash tools/debug/run_k2.sh
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_matrix.py tests/test_gpu_fullsize.py tests/test_exact_f64.py -q -m gpu -x -k "reassigned or cfg2 or window or 4096 or exact" 2>&1 | tail -4
python bench.py --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])
"
