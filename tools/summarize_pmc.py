"""Summarise rocprofv3 CSV outputs (kernel stats + PMC passes) into one small text file."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "*kernel_stats.csv")):
    lines.append("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    with open(f) as fh:
        for row in list(csv.DictReader(fh))[:6]:
            name = row["Name"][:70]
            lines.append(f'{name:70s} calls={row["Calls"]:>4s} avg_ns={float(row["AverageNs"]):>12.0f} pct={row["Percentage"]}')
for p in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(p):
        continue
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(p, "*counter_collection.csv")):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "omx::" not in k:
                    continue
                agg[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    lines.append(f"== {os.path.basename(p)} (per-dispatch mean over omx kernels) ==")
    for k, cs in agg.items():
        for c, vals in sorted(cs.items()):
            lines.append(f"{k:40s} {c:24s} mean={sum(vals)/len(vals):.6g} n={len(vals)}")
# HBM traffic of the dominant kernel, per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes:
# FETCH_SIZE / WRITE_SIZE are in KiB and come from separate passes; on gfx950 FETCH_SIZE reports half of the bytes of a wide
# coalesced read, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
import json
traffic = {}
for p in sorted(glob.glob(os.path.join(out, "pmc*"))):
    for f in glob.glob(os.path.join(p, "*counter_collection.csv")):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "stft_reassigned_4096_" in row.get("Kernel_Name", "") and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    traffic.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
    fetch = sum(traffic["FETCH_SIZE"]) / len(traffic["FETCH_SIZE"])
    write = sum(traffic["WRITE_SIZE"]) / len(traffic["WRITE_SIZE"])
    # provenance: which commit and which kernel time this traffic figure belongs to (bench.py echoes it as a tagged record only)
    commit, kernel_ms, transforms = os.environ.get("OMX_PROFILE_COMMIT"), None, None
    try:
        with open(os.path.join(out, "bench_line.json")) as fh:
            line = json.loads(fh.read().strip().splitlines()[-1])
        kernel_ms, transforms = line["roofline"]["kernel_ms"], line["roofline"].get("transforms_per_frame")
    except Exception:
        pass
    rec = {"kernel": "stft_reassigned_4096_tri_kernel", "fetch_size_kib": fetch, "write_size_kib": write,
           "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0, "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE x2)",
           "launches": len(traffic["FETCH_SIZE"]), "commit": commit, "kernel_ms": kernel_ms, "transforms_per_frame": transforms,
           "workload": {"config": "cfg2", "streams_per_gpu": 64, "columns_per_step_per_gpu": 65536}}
    with open(os.path.join(out, "traffic.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    lines.append(f"== HBM traffic per launch (stft_reassigned_4096_tri_kernel): {rec['hbm_bytes_per_launch'] / 1e9:.3f} GB ==")
txt = "\n".join(lines)
print(txt)
with open(os.path.join(out, "summary.txt"), "w") as fh:
    fh.write(txt + "\n")
