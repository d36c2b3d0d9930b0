"""Feasibility: does a VALU-bound pass A overlap a memory-bound pass B when two half banks run on two HIP streams?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import openmeters_amd
from openmeters_amd import banks, capi
api = openmeters_amd.api()
dev = torch.device("cuda", 0)
FS = 48000.0
C, blocks = 8, 64
frames = 256 * blocks
def make(S):
    n = torch.arange(frames, device=dev, dtype=torch.float64)
    pcm = torch.empty((S, frames, C), device=dev, dtype=torch.float32)
    for c in range(C):
        pcm[:, :, c] = (0.5 * torch.sin(2 * np.pi * (997.0 + 10 * c) * n / FS)).to(torch.float32)[None, :]
    return pcm, banks.LoudnessBank(api, capi.LoudnessConfig(), S, C)
def timed(fn, reps=20):
    for _ in range(14): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
p1, b1 = make(1024)
print("one bank of 1024:", timed(lambda: b1.process_device(p1.data_ptr(), 256, blocks, C, FS, capi.SURROUND, torch.cuda.current_stream().cuda_stream)))
for parts in (2, 4):
    S = 1024 // parts
    items = [make(S) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def run():
        for (p, b), st in zip(items, streams):
            b.process_device(p.data_ptr(), 256, blocks, C, FS, capi.SURROUND, st.cuda_stream)
    print(f"{parts} banks of {S} on {parts} streams:", timed(run))
    s0 = torch.cuda.current_stream().cuda_stream
    def run1():
        for (p, b) in items:
            b.process_device(p.data_ptr(), 256, blocks, C, FS, capi.SURROUND, s0)
    print(f"{parts} banks of {S} on one stream:", timed(run1))
