# which HIP calls / copies one omx_capture_group_ingest makes at the reference's cadence — usage: gpurun -- bash tools/debug/stream_trace.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stream_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_stream.py --calls 100 > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python3 - <<PY
import csv, glob, collections, os
out = "$OUT"
for f in glob.glob(os.path.join(out, "*hip_api_trace.csv")):
    c = collections.Counter(r["Function"] for r in csv.DictReader(open(f)))
    print("HIP API calls:", {k: v for k, v in c.most_common(25)})
for f in glob.glob(os.path.join(out, "*memory_copy_trace.csv")):
    rows = list(csv.DictReader(open(f)))
    print("memory copies:", len(rows), "columns", list(rows[0].keys()) if rows else None)
    c = collections.Counter((r.get("Direction"), r.get("Bytes") or r.get("Size")) for r in rows)
    for k, v in c.most_common(30):
        print("   ", k, v)
for f in glob.glob(os.path.join(out, "*kernel_trace.csv")):
    c = collections.Counter(r["Kernel_Name"][:60] for r in csv.DictReader(open(f)))
    print("kernels:", {k: v for k, v in c.most_common(30)})
PY
