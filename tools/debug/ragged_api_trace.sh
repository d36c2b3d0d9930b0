cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ragged_api
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/rg.py <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import numpy as np, torch, openmeters_amd
from openmeters_amd import capi
from openmeters_amd.pipeline import CaptureGroup
api = openmeters_amd.api(); dev = torch.device("cuda", 0)
S, F, FS = 1024, 256, 48000.0
pos = capi.positions_fallback(2)
FR = np.full(S, F, np.uint32)
pcm = (0.1 * (torch.rand((S, F, 2), device=dev) - 0.5)).contiguous()
g = CaptureGroup(api, S, loudness=capi.LoudnessConfig())
for k in range(60): g.ingest_ragged(pcm.data_ptr(), F, FR, 2, FS, pos)
torch.cuda.synchronize()
PY
rocprofv3 --hip-trace --output-format csv -d $OUT -o t -- python3 /tmp/rg.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/*hip_api_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in rows[len(rows) // 2:]:   # steady state
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg[r["Function"]]; a[0] += 1; a[1] += d; a[2] = max(a[2], d)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"{k:40s} calls {v[0]:5d} total_us {v[1]:10.1f} max_us {v[2]:8.1f}")
PY
