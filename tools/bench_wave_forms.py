"""Waveform bank: sequential kernels against the chunk-parallel form over bank and call sizes (where the by-shape rule should switch)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
FS = 48000.0
pos = capi.positions_fallback(2)
for S, frames in ((1, 16384), (4, 16384), (16, 16384), (64, 16384), (256, 16384), (64, 4096), (256, 4096), (1024, 4096), (1024, 1024), (4096, 1024), (64, 65536)):
    pcm = (torch.rand((S, frames, 2), device="cuda:0") - 0.5).contiguous()
    row = []
    for form in (1, 2):
        bank = banks.WaveformBank(api, capi.WaveformConfig(analyze_bands=True, track_history=False), S)
        bank.set_option(capi.OPT_KERNEL_FORM, form)
        run = lambda: bank.process_device(pcm.data_ptr(), frames, 2, FS, pos, 0)
        run(); run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 5 * 1e3)
        assert bank.last_form() == form
        bank.close()
    print(f"{S:5d} streams x {frames:6d} frames ({S * frames / 1e6:6.2f} M stream-frames): sequential {row[0]:7.3f} ms   chunk-parallel {row[1]:7.3f} ms   ratio {row[0] / row[1]:5.2f}")
