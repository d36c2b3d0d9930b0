"""Reassigned STFT throughput of the zero-padded shapes (window W padded to F = zp * W), 64 streams; OMX_FORCE_GENERIC-style
comparison is in the unit tests — this prints kernel time per launch (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
S = 64
for W, zp, hop in ((2048, 2, 64), (4096, 2, 256), (1024, 16, 256), (2048, 8, 64), (4096, 4, 256), (8192, 2, 512)):
    F = W * zp
    cols = (16384 if F <= 8192 else 4096) // S
    frames = 2 * W + hop * (cols - 1)
    pcm = (torch.rand((S, frames + hop * cols * 4, 2), device="cuda:0") - 0.5).contiguous()
    bank = banks.SpectrogramBank(api, capi.SpectrogramConfig(fft_size=W, hop_size=hop, zero_padding_factor=zp, use_reassignment=True,
                                                              history_length=8192), S)
    bank.set_option(capi.OPT_KERNEL_TIMING, 1)
    pos = capi.positions_fallback(2)
    bank.process_device(pcm[:, :frames].contiguous().data_ptr(), frames, 2, 48000.0, pos)
    bank.kernel_time()
    for it in range(4):
        chunk = pcm[:, frames + it * hop * cols: frames + (it + 1) * hop * cols].contiguous()
        bank.process_device(chunk.data_ptr(), hop * cols, 2, 48000.0, pos)
    torch.cuda.synchronize()
    ms, n = bank.kernel_time()
    print(f"W={W} x{zp} (F={F}) hop={hop}: kernel {ms:.3f} ms per {S * cols} frames -> {S * cols / ms / 1e3:.2f} M frames/s ({n} launches)")
