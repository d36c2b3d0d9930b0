"""Generic-kernel shapes (zero padding, small windows, 16384 reassigned) — run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
S = 64
for W, zp, hop, reassign in ((2048, 2, 64, True), (1024, 4, 256, True), (4096, 2, 256, True), (512, 1, 128, True), (2048, 2, 256, False),
                             (16384, 1, 1024, True)):
    cols = 4096 // S
    frames = 2 * W + hop * (cols - 1)
    pcm = (torch.rand((S, frames + hop * cols * 3, 2), device="cuda:0") - 0.5).contiguous()
    bank = banks.SpectrogramBank(api, capi.SpectrogramConfig(fft_size=W, hop_size=hop, use_reassignment=reassign, zero_padding_factor=zp,
                                                             history_length=8192), S)
    bank.set_option(capi.OPT_KERNEL_TIMING, 1)
    pos = capi.positions_fallback(2)
    bank.process_device(pcm[:, :frames].contiguous().data_ptr(), frames, 2, 48000.0, pos)
    bank.kernel_time()
    for it in range(3):
        chunk = pcm[:, frames + it * hop * cols: frames + (it + 1) * hop * cols].contiguous()
        bank.process_device(chunk.data_ptr(), hop * cols, 2, 48000.0, pos)
    torch.cuda.synchronize()
    ms, n = bank.kernel_time()
    print(f"W={W} zp={zp} hop={hop} reassign={reassign}: {ms:.3f} ms per {S * cols} frames -> {S * cols / ms / 1e3:.2f} M frames/s")
