#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the cfg3 / cfg4 meter kernels and of the waveform bank at ONE size
# (1024 streams, RMS history off "@1024" and on "#1024": per-launch records); outputs under gpurun_out/$1
set -u
TAG=${1:-meters_pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py nowave > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py nowave > $OUT/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/wf_fetch -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py waveform 1024 0 > $OUT/wf_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/wf_write -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py waveform 1024 0 > $OUT/wf_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/wh_fetch -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py waveform 1024 1 > $OUT/wh_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/wh_write -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_meters.py waveform 1024 1 > $OUT/wh_write.log 2>&1
python3 - <<PY
import csv, glob, os
from collections import defaultdict
out = "$OUT"
agg = defaultdict(lambda: defaultdict(list))
for d, suffix in (("fetch", ""), ("write", ""), ("wf_fetch", "@1024"), ("wf_write", "@1024"), ("wh_fetch", "#1024"), ("wh_write", "#1024")):
    for f in glob.glob(os.path.join(out, d, "*counter_collection.csv")):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "omx::" in k:
                    agg[k.split("(")[0][:60].strip() + suffix][row["Counter_Name"]].append(float(row["Counter_Value"]))
import json
lines, rec = [], {"commit": os.environ.get("OMX_PROFILE_COMMIT"), "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE x2)", "kernels": {}}
for k, cs in agg.items():
    fe = sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [1])), 1)
    wr = sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [1])), 1)
    lines.append(f"{k:66s} FETCH_SIZE {fe:12.0f} KiB  WRITE_SIZE {wr:12.0f} KiB  -> HBM bytes per launch (2*F + W)*1024 = {(2 * fe + wr) * 1024 / 1e9:.3f} GB")
    rec["kernels"][k] = {"fetch_size_kib": fe, "write_size_kib": wr, "hbm_bytes_per_launch": (2 * fe + wr) * 1024, "launches": len(cs.get("FETCH_SIZE", []))}
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(rec, open(os.path.join(out, "meters_traffic.json"), "w"), indent=1)
print("\n".join(lines))
PY
