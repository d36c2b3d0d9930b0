export OMX_HIP_LIB=$PWD/openmeters_amd/csrc/libomx_hip_tuning.so
for i in 1 2; do
OMX_INGEST_SINGLE=1 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one frame per lane ', d['ms_per_step'], d['roofline']['kernel_ms'])"
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('four frames per lane', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
