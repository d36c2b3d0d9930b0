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
txt = "\n".join(lines)
print(txt)
with open(os.path.join(out, "summary.txt"), "w") as fh:
    fh.write(txt + "\n")
