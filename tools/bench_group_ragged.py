"""The six-visual capture group alone, lock-step or ragged at equal work (run on the GPU box; the last lines of tools/bench_ragged.py as
a tool of their own, for kernel traces): python tools/bench_group_ragged.py lock|ragged [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import openmeters_amd
from openmeters_amd import capi
from openmeters_amd.pipeline import CaptureGroup
import bench_stream

mode = sys.argv[1] if len(sys.argv) > 1 else "ragged"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
api = openmeters_amd.api()
dev = torch.device("cuda", 0)
pos = capi.positions_fallback(2)
S, F, FS = 1024, 256, 48000.0
n = torch.arange(F * 8, device=dev, dtype=torch.float32)
base = (0.4 * torch.sin(2 * torch.pi * 440.0 * n / 48000.0))[None, :, None] * torch.tensor([1.0, -0.7], device=dev)[None, None, :]
pcm = (base + 0.01 * (torch.rand((S, F * 8, 2), device=dev) - 0.5)).contiguous()
chunks = [pcm[:, k * F:(k + 1) * F].contiguous() for k in range(8)]
g = CaptureGroup(api, S, stats=True, **bench_stream.default_configs())
k = [0]
counts = np.full(S, F, np.uint32)   # (a prebuilt array: the list -> array conversion of 1024 entries costs the tool ~40 us per call)


def call():
    if mode == "lock":
        g.ingest(chunks[k[0] % 8].data_ptr(), F, 2, FS, pos)
    else:
        g.ingest_ragged(chunks[k[0] % 8].data_ptr(), F, counts, 2, FS, pos)
    k[0] += 1


for _ in range(40):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
host = 0.0
for _ in range(calls):
    h0 = time.perf_counter()
    call()
    host += time.perf_counter() - h0
torch.cuda.synchronize()
print(f"capture group {mode}: {(time.perf_counter() - t0) / calls * 1e6:.1f} us per call, host enqueue {host / calls * 1e6:.1f} us per call")
