"""Spectrum bank throughput per FFT size (run on the GPU box): default reference shape is 16384 / hop 1024."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
S = 64
for N in (1024, 4096, 8192, 16384):
    hop = N // 16
    hops = 1024
    frames = N + hop * (hops - 1)
    pcm = (torch.rand((S, frames + hop * hops * 3, 2), device="cuda:0") - 0.5).contiguous()
    bank = banks.SpectrumBank(api, capi.SpectrumConfig(fft_size=N, hop_size=hop), S, emit_all_hops=True)
    pos = capi.positions_fallback(2)
    bank.process_device(pcm[:, :frames].contiguous().data_ptr(), frames, 2, 48000.0, pos)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for it in range(3):
        chunk = pcm[:, frames + it * hop * hops: frames + (it + 1) * hop * hops].contiguous()
        bank.process_device(chunk.data_ptr(), hop * hops, 2, 48000.0, pos)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    print(f"spectrum N={N} hop={hop}: {ms:.3f} ms per {S * hops} hops -> {S * hops / ms / 1e3:.2f} M hops/s "
          f"({S * hops / ms * 1e3 / (48000.0 / hop) / S:.0f}x real time for {S} streams)")
