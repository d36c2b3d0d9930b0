"""Oscilloscope bank per sample rate (run on the GPU box): 256 streams x 2 ch, 64 batcher blocks per call.  27.3 ... 54.6 kHz run the wide
form (scope_fast_kernels.hip, 8192-point autocorrelation); other rates the single-pass kernel of round 1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
dev = torch.device("cuda", 0)
S, blocks = 256, 64
for fs in (44100.0, 48000.0, 88200.0, 96000.0, 192000.0):
    block = int(round(256 * fs / 48000.0))
    frames = block * blocks
    n = torch.arange(frames, device=dev, dtype=torch.float64)
    pcm = torch.empty((S, frames, 2), device=dev, dtype=torch.float32)
    for s in range(S):
        f = 440.0 * 2.0 ** ((s % 24) / 12.0)
        left = (0.8 * torch.sin(2 * np.pi * f * n / fs)).to(torch.float32)
        pcm[s, :, 0] = left
        pcm[s, :, 1] = -0.7 * left
    pos = capi.positions_fallback(2)
    sc = banks.OscilloscopeBank(api, capi.OscilloscopeConfig(sample_rate=fs, segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2,
                                                             trigger_source=capi.CH_LEFT, channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT), S)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        sc.process_device(pcm.data_ptr(), block, blocks, 2, fs, pos, stream)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        sc.process_device(pcm.data_ptr(), block, blocks, 2, fs, pos, stream)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    hdr, _ = sc.fetch(0, blocks - 1)
    print(f"oscilloscope {fs:.0f} Hz, block {block}: {ms:.2f} ms per {S} x {blocks} blocks -> {S * blocks / ms / 1e3:.2f} M blocks/s, "
          f"{frames / fs / (ms / 1e3):.0f}x real time; locked={hdr.locked} period={hdr.period:.2f}")
