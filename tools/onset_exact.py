import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/oracle")
import numpy as np
import conftest, openmeters_amd
from openmeters_amd.capi import Api, AudioBlock, SpectrogramConfig, SpectrogramProcessor
from parity import reassigned_column_metrics
import exact_f64 as ex
import test_gpu_state_machine as t
omx = openmeters_amd.api(); oracle = Api(conftest._build_oracle(), "omxo_")
FS = 48000.0
for kind in (1, 3, 4):
  for W, hop in ((4096, 100), (2048, 64)):
    rng = np.random.default_rng(5)
    H = 2 * W
    ncols = 40
    total = H + hop * (ncols - 1)
    sig = t.signal(rng, total, 2, 0, FS, False)
    sig[: total - 3000] = 0.0     # silence, then the onset inside the last windows
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, window=kind, use_reassignment=True, history_length=8192)
    mid = ((sig[:, 0] + sig[:, 1]) * np.float32(0.5)).astype(np.float32)
    res = {}
    for name, api in (("hip", omx), ("oracle", oracle)):
        up = SpectrogramProcessor(api, cfg).process_block(AudioBlock(sig.reshape(-1), 2, FS))
        worst = dict(power=0, freq=0, time=0, freq_strong=0)
        for c in range(ncols):
            col = up.new_columns[c].astype(np.float64)
            if len(col) == 0: continue
            pts, _ = ex.reassigned_column(mid[c * hop:], kind, W, 1, hop, FS)
            if len(pts) == 0 or pts[:, 2].max() < 1e-10: continue
            m = reassigned_column_metrics(col, pts, FS, hop)
            for k in worst: worst[k] = max(worst[k], m[k])
        res[name] = worst
    print(f"window {kind} W {W} hop {hop}: hip vs exact {res['hip']}  oracle vs exact {res['oracle']}")
