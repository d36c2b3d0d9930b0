export TMPDIR=/tmp
for lib in openmeters_amd/csrc/tuning/libomx_hip_base.so openmeters_amd/csrc/libomx_hip.so; do
  echo "== $lib"; export OMX_HIP_LIB=$PWD/$lib
  rm -rf /tmp/abp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o p -- python3 tools/bench_meters.py > /tmp/abp.log 2>&1
  python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/abp/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('loud_','stereo_','scope_','waveform_roles')): print(f"{n[:70]:70s} {int(r['Calls']):4d} {float(r['AverageNs'])/1e3:9.1f} us")
PY
  grep -E "cfg3|cfg4|1024 streams" /tmp/abp.log | cut -c1-110
done
