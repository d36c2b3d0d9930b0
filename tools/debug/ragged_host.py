"""host time of omx_capture_group_ingest_ragged per visual (1024 captures x 256 frames)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch, openmeters_amd
from openmeters_amd import capi
from openmeters_amd.pipeline import CaptureGroup
import bench_stream
api = openmeters_amd.api(); dev = torch.device("cuda", 0)
S, F, FS = 1024, 256, 48000.0
pos = capi.positions_fallback(2)
FR = np.full(S, F, np.uint32)
pcm = (0.1 * (torch.rand((S, F * 8, 2), device=dev) - 0.5)).contiguous()
chunks = [pcm[:, k * F:(k + 1) * F].contiguous() for k in range(8)]
allc = bench_stream.default_configs()
for vis in [list(allc)] + [[v] for v in allc]:
    cfgs = {k: allc[k] for k in vis}
    for name, ragged in (("lock-step", False), ("ragged", True)):
        g = CaptureGroup(api, S, **cfgs)
        call = (lambda k: g.ingest_ragged(chunks[k % 8].data_ptr(), F, FR, 2, FS, pos)) if ragged else (lambda k: g.ingest(chunks[k % 8].data_ptr(), F, 2, FS, pos))
        for k in range(40): call(k)
        torch.cuda.synchronize()
        host = 0.0
        t0 = time.perf_counter()
        for k in range(200):
            h0 = time.perf_counter(); call(k); host += time.perf_counter() - h0
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        print(f"{'+'.join(v[:5] for v in vis):40s} {name:10s}: {total / 200 * 1e6:7.1f} us per call, host {host / 200 * 1e6:7.1f} us")
