export TMPDIR=/tmp
rm -rf /tmp/abp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o p -- python3 tools/bench_meters.py scope > /tmp/abp.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/abp/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'scope_' in n: print(f"{n[:70]:70s} {int(r['Calls']):4d} {float(r['AverageNs'])/1e3:9.1f} us")
PY
grep -E "cfg4 osc" /tmp/abp.log | cut -c1-110
