#!/bin/bash
# After `gpurun -- bash tools/round_refresh.sh <tag> <parity tag> <commit>`: copy the round's records from gpurun_out/ into profiles/.
# Fails (nothing copied) unless BOTH traffic records are there, are not older than the bench line, and carry the commit the run was
# told it profiles — the two files bench.py echoes into roofline.traffic must never lag the lines that quote them.
# usage (here, not on the box): bash tools/collect_round.sh <tag> <commit>
set -e
TAG=$1; COMMIT=$2
cd "$(dirname "$0")/.."
for f in gpurun_out/${TAG}_traffic.json gpurun_out/${TAG}_meters_traffic.json gpurun_out/${TAG}_bench_line.json; do
  [ -s "$f" ] || { echo "collect_round: $f is missing or empty" >&2; exit 1; }
done
python3 - "$TAG" "$COMMIT" <<'PY'
import json, os, sys
tag, commit = sys.argv[1], sys.argv[2]
for name in (f"gpurun_out/{tag}_traffic.json", f"gpurun_out/{tag}_meters_traffic.json"):
    rec = json.load(open(name))
    if rec.get("commit") != commit:
        sys.exit(f"collect_round: {name} was profiled at commit {rec.get('commit')!r}, not {commit!r}")
# the bench line must quote THIS round's record (bench.py echoes the newest profiles/*_traffic.json it finds on the box)
line = [json.loads(l) for l in open(f"gpurun_out/{tag}_bench_line.json") if l.startswith("{")][-1]
src = (line.get("roofline") or {}).get("traffic_source") or {}
if src.get("file") != f"profiles/{tag}_traffic.json" or src.get("commit") != commit:
    sys.exit(f"collect_round: the bench line echoes {src.get('file')!r} at commit {src.get('commit')!r}, expected profiles/{tag}_traffic.json at {commit!r}")
PY
for f in gpurun_out/${TAG}_*; do
  [ -f "$f" ] && cp "$f" profiles/
done
[ -f gpurun_out/${TAG}/summary.txt ] && cp gpurun_out/${TAG}/summary.txt profiles/${TAG}_bench_summary.txt
[ -f gpurun_out/${TAG}/trace/t_kernel_stats.csv ] && cp gpurun_out/${TAG}/trace/t_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
echo "collect_round: profiles/${TAG}_* refreshed from gpurun_out/ (commit $COMMIT)"
