#!/bin/bash
# Kernel-trace stats for the non-headline shapes (run on the GPU box via gpurun); outputs under gpurun_out/$1
set -u
TAG=${1:-others}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in bench_meters bench_classic bench_sizes bench_splat bench_spectrum_sizes bench_pipeline; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$t -o t -- python3 $GRAFT_REPO_ROOT/tools/$t.py > $OUT/$t.log 2>&1
done
python3 - <<PY
import csv, glob, os
out = "$OUT"
lines = []
for t in ("bench_meters", "bench_classic", "bench_sizes", "bench_splat", "bench_spectrum_sizes", "bench_pipeline"):
    lines.append(f"== {t}.py: program output ==")
    with open(os.path.join(out, t + ".log")) as fh:
        lines += [l.rstrip() for l in fh if "->" in l or "real time" in l]
    lines.append(f"== {t}.py: rocprofv3 --kernel-trace --stats (omx kernels) ==")
    for f in glob.glob(os.path.join(out, t, "*kernel_stats.csv")):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "omx::" in row["Name"]:
                    lines.append(f'{row["Name"][:96]:96s} calls={row["Calls"]:>5s} avg_ns={float(row["AverageNs"]):>12.0f}')
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
