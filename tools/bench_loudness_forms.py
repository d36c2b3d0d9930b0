"""Loudness bank: sequential kernels vs the chunk-parallel form on one box (cfg5 shard shape 1024 x 2 ch, cfg3 1024 x 8 ch)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import openmeters_amd  # noqa: E402
from openmeters_amd import banks, capi  # noqa: E402

api = openmeters_amd.api()
dev = torch.device("cuda", 0)
FS = 48000.0
stream = torch.cuda.current_stream().cuda_stream
for S, C, blocks in ((1024, 2, 64), (1024, 8, 64), (256, 2, 64)):
    frames = 256 * blocks
    n = torch.arange(frames, device=dev, dtype=torch.float64)
    pcm = torch.empty((S, frames, C), device=dev, dtype=torch.float32)
    for c in range(C):
        pcm[:, :, c] = (0.5 * torch.sin(2 * np.pi * (997.0 + 10.0 * c) * n / FS)).to(torch.float32)[None, :]
    pos = capi.SURROUND if C == 8 else capi.positions_fallback(C)
    for form, name in ((1, "sequential"), (2, "chunk-parallel")):
        bank = banks.LoudnessBank(api, capi.LoudnessConfig(), S, C)
        bank.set_option(capi.OPT_KERNEL_FORM, form)
        run = lambda: bank.process_device(pcm.data_ptr(), 256, blocks, C, FS, pos, stream)
        for _ in range(12):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        snap = bank.fetch(0, blocks - 1)
        print(f"{S} streams x {C} ch x {blocks} blocks, {name}: {dt * 1e3:.3f} ms/call -> {S * C * frames / dt / 1e9:.1f} G channel-samples/s, "
              f"{frames / dt / FS:.0f}x real time; LUFS-S {snap.short_term_loudness:.4f} M {snap.momentary_loudness:.4f}")
