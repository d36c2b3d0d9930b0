"""Reassigned STFT throughput for the fused window sizes (run on the GPU box): W in {1024, 2048, 4096}, hop 64 / 256."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
S = 64
for W, hop in ((1024, 256), (2048, 64), (2048, 256), (4096, 256), (8192, 512), (16384, 1024)):
    cols = (65536 if W <= 8192 else 4096) // S   # 16384 reassigned runs the generic kernel
    frames = 2 * W + hop * (cols - 1)
    pcm = (torch.rand((S, frames + hop * cols * 4, 2), device="cuda:0") - 0.5).contiguous()
    bank = banks.SpectrogramBank(api, capi.SpectrogramConfig(fft_size=W, hop_size=hop, use_reassignment=True, history_length=8192), S)
    bank.set_option(capi.OPT_KERNEL_TIMING, 1)
    pos = capi.positions_fallback(2)
    bank.process_device(pcm[:, :frames].contiguous().data_ptr(), frames, 2, 48000.0, pos)
    bank.kernel_time()
    for it in range(4):
        chunk = pcm[:, frames + it * hop * cols: frames + (it + 1) * hop * cols].contiguous()
        bank.process_device(chunk.data_ptr(), hop * cols, 2, 48000.0, pos)
    torch.cuda.synchronize()
    ms, n = bank.kernel_time()
    frames_per = S * cols
    bytes_per = hop * 2 * 4 + 4 + 12 * (W // 2 - 1)
    print(f"W={W} hop={hop}: kernel {ms:.3f} ms per {frames_per} frames -> {frames_per / ms / 1e3:.1f} M frames/s, "
          f"{frames_per * bytes_per / ms / 1e6:.0f} GB/s algorithmic ({n} launches)")
