"""debug: host enqueue time against total time of the six-visual group at the reference's cadence"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import openmeters_amd
from openmeters_amd import capi
from openmeters_amd.pipeline import CaptureGroup
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench_stream as bs
api = openmeters_amd.api()
dev = torch.device("cuda", 0)
streams, frames = 1024, 256
group = CaptureGroup(api, streams, **bs.default_configs())
pos = capi.positions_fallback(2)
pcm = (torch.rand((streams, frames * 8, 2), device=dev) - 0.5).contiguous()
chunks = [pcm[:, k * frames:(k + 1) * frames].contiguous() for k in range(8)]
stream = torch.cuda.current_stream().cuda_stream
for k in range(40):
    group.ingest(chunks[k % 8].data_ptr(), frames, 2, 48000.0, pos, stream)
torch.cuda.synchronize()
calls = 200
t0 = time.perf_counter()
for k in range(calls):
    group.ingest(chunks[k % 8].data_ptr(), frames, 2, 48000.0, pos, stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / calls:.1f} us per call; total {1e6 * (t2 - t0) / calls:.1f} us per call; drain after the loop {1e6 * (t2 - t1):.0f} us")
# one call at a time, synchronised: the device-side latency of one call
lat = []
for k in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    group.ingest(chunks[k % 8].data_ptr(), frames, 2, 48000.0, pos, stream)
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t0)
lat.sort()
print(f"one call, synchronised: median {1e6 * lat[len(lat) // 2]:.1f} us, min {1e6 * lat[0]:.1f} us")
