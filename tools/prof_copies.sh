#!/bin/bash
# what the streaming cadence copies per call: bash tools/prof_copies.sh  (memory-copy + HIP runtime trace of tools/bench_stream.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf $R/gpurun_out/prof_copies
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --stats --output-format csv -d $R/gpurun_out/prof_copies -o t -- python3 $R/tools/bench_stream.py --calls 100 "$@" > $R/gpurun_out/prof_copies.log 2>&1 < /dev/null
grep "captures" $R/gpurun_out/prof_copies.log
ls $R/gpurun_out/prof_copies
python3 - <<PY
import csv, collections, glob
f = glob.glob("$R/gpurun_out/prof_copies/*memory_copy_trace.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    print("copies", len(rows), "columns", list(rows[0].keys()) if rows else None)
    c = collections.Counter((r.get("Direction"), r.get("Bytes") or r.get("Size")) for r in rows)
    for k, v in c.most_common(40): print(v, k)
f = glob.glob("$R/gpurun_out/prof_copies/*hip_api_stats.csv") + glob.glob("$R/gpurun_out/prof_copies/*hip_stats.csv")
for g in f:
    for line in open(g).read().splitlines()[:25]: print(line[:140])
PY
