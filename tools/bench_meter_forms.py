"""Loudness / stereometer banks: sequential kernels against the chunk-parallel forms over bank and call sizes (where the by-shape rules
should switch)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
FS = 48000.0


def timed(run):
    run(); run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3


for S, C, blocks in ((1, 2, 64), (1, 8, 64), (16, 2, 64), (64, 2, 64), (256, 2, 64), (1024, 2, 16), (1024, 2, 8), (64, 8, 16), (1, 2, 256), (16, 2, 8)):
    pcm = (torch.rand((S, 256 * blocks, C), device="cuda:0") - 0.5).contiguous()
    pos = capi.SURROUND if C == 8 else capi.positions_fallback(C)
    row = []
    for form in (1, 2):
        bank = banks.LoudnessBank(api, capi.LoudnessConfig(), S, C)
        bank.set_option(capi.OPT_KERNEL_FORM, form)
        row.append(timed(lambda: bank.process_device(pcm.data_ptr(), 256, blocks, C, FS, pos, 0)))
        row.append(bank.last_form())
        bank.close()
    print(f"loudness    {S:5d} streams x {C} ch x {blocks:4d} blocks: sequential {row[0]:7.3f} ms (form {row[1]})   chunk-parallel {row[2]:7.3f} ms (form {row[3]})   ratio {row[0] / row[2]:5.2f}")
for S, blocks in ((1, 64), (16, 64), (64, 64), (256, 64), (256, 8), (1024, 8), (1, 256), (16, 8)):
    pcm = (torch.rand((S, 256 * blocks, 2), device="cuda:0") - 0.5).contiguous()
    pos = capi.positions_fallback(2)
    row = []
    for form in (1, 2):
        bank = banks.StereometerBank(api, capi.StereometerConfig(analyze_bands=True), S)
        bank.set_option(capi.OPT_KERNEL_FORM, form)
        row.append(timed(lambda: bank.process_device(pcm.data_ptr(), 256, blocks, 2, FS, pos, 0)))
        row.append(bank.last_form())
        bank.close()
    print(f"stereometer {S:5d} streams x {blocks:4d} blocks: sequential {row[0]:7.3f} ms (form {row[1]})   chunk-parallel {row[2]:7.3f} ms (form {row[3]})   ratio {row[0] / row[2]:5.2f}")
