#!/bin/bash
# Per-kernel times of a tools/*.py size sweep (run on the GPU box): tools/profile_sizes.sh bench_sizes.py
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_sizes"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o sizes -- python3 "$REPO/tools/$1" > "$OUT/run.log" 2>&1
tail -8 "$OUT/run.log" | grep -v simple_timer
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print(f'{r["Name"][:100]:100s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"]) / 1e3:10.1f} pct={r["Percentage"]}')
PY
