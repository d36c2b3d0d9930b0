"""cfg1-shaped classic STFT (1024-pt Hann, hop 256, u16 dB codes) on many streams (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
for W, S, cols in ((1024, 64, 4096), (2048, 64, 2048), (4096, 64, 1024)):
    frames = W + 256 * (cols - 1)
    pcm = (torch.rand((S, frames + 256 * cols * 3, 2), device="cuda:0") - 0.5).contiguous()
    bank = banks.SpectrogramBank(api, capi.SpectrogramConfig(fft_size=W, hop_size=256, use_reassignment=False, history_length=8192), S)
    pos = capi.positions_fallback(2)
    bank.process_device(pcm[:, :frames].contiguous().data_ptr(), frames, 2, 48000.0, pos)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for it in range(3):
        chunk = pcm[:, frames + it * 256 * cols: frames + (it + 1) * 256 * cols].contiguous()
        up = bank.process_device(chunk.data_ptr(), 256 * cols, 2, 48000.0, pos)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    n = S * cols
    print(f"classic W={W}: {ms:.3f} ms per {n} frames -> {n / ms / 1e3:.2f} M frames/s, "
          f"{n * (256 * 2 * 4 + (W // 2 + 1) * 2) / ms / 1e6:.1f} GB/s algorithmic")
