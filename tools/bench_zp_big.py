"""Spectrogram throughput of the zero-padded shapes beyond 16384 points (window W padded to F = zp * W; the GUI offers zero padding
up to 32x of 1024 ... 16384-point windows, reference src/ui/settings/spectrogram.rs:13), few columns per call (a reassigned column
holds up to F / 2 + 1 points of 12 bytes).  `--classic` adds the non-reassigned columns.  Prints wall time per call (run on the GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
shapes = ((2048, 16, 64), (2048, 32, 64), (4096, 8, 256), (4096, 32, 256), (8192, 4, 512), (8192, 32, 512), (16384, 2, 1024), (16384, 32, 1024))
modes = (True, False) if "--classic" in sys.argv else (True,)
args = [a for a in sys.argv[1:] if a != "--classic"]
if args:
    shapes = tuple(tuple(int(x) for x in a.split("x")) for a in args)
for reassign in modes:
    for W, zp, hop in shapes:
        F = W * zp
        S = 8
        cols = max(1, min(64, (1 << 21) // F))      # columns per stream and call
        frames = 2 * W + hop * (cols - 1)
        pcm = (torch.rand((S, frames + hop * cols * 3, 2), device="cuda:0") - 0.5).contiguous()
        bank = banks.SpectrogramBank(api, capi.SpectrogramConfig(fft_size=W, hop_size=hop, zero_padding_factor=zp, use_reassignment=reassign,
                                                                  history_length=8192), S)
        pos = capi.positions_fallback(2)
        bank.process_device(pcm[:, :frames].contiguous().data_ptr(), frames, 2, 48000.0, pos)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for it in range(n):
            chunk = pcm[:, frames + it * hop * cols: frames + (it + 1) * hop * cols].contiguous()
            bank.process_device(chunk.data_ptr(), hop * cols, 2, 48000.0, pos)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{'reassigned' if reassign else 'classic   '} W={W} x{zp} (F={F}) hop={hop}: {dt*1e3:.3f} ms per {S * cols} frames -> "
              f"{S * cols / dt / 1e3:.1f} k frames/s = {S * cols / dt * hop / 48000.0 / S:.1f}x real time per stream")
        bank.close()
