"""cfg4 stereometer: sequential kernels vs the chunk-parallel form, same box (256 streams x 64 blocks of 256 per call)."""
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
for S, blocks in ((256, 64), (1024, 64), (256, 16)):
    frames = 256 * blocks
    n = torch.arange(frames, device=dev, dtype=torch.float64)
    pcm = torch.empty((S, frames, 2), device=dev, dtype=torch.float32)
    left = (0.8 * torch.sin(2 * np.pi * 440.0 * n / FS)).to(torch.float32)
    pcm[:, :, 0] = left[None]
    pcm[:, :, 1] = -0.7 * left[None] + 0.01 * torch.randn((S, frames), device=dev)
    pos = capi.positions_fallback(2)
    for form, name in ((1, "sequential"), (2, "chunk-parallel")):
        st = banks.StereometerBank(api, capi.StereometerConfig(analyze_bands=True, correlation_window=0.05, segment_duration=0.02,
                                                               target_sample_count=2000), S)
        st.set_option(capi.OPT_KERNEL_FORM, form)
        run = lambda: st.process_device(pcm.data_ptr(), 256, blocks, 2, FS, pos, stream)
        run(); run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{S} streams x {blocks} blocks, {name}: {dt * 1e3:.3f} ms/call -> {S * blocks / dt / 1e6:.2f} M blocks/s, {frames / dt / FS:.0f}x real time")
