"""Host-buffer (PCIe-inclusive) rate of the headline workload: the same 64-stream x 262 144-frame step as bench.py, but the PCM
starts in host memory and `omx_spectrogram_bank_process` copies it to the device itself (pageable numpy array, then a
torch-pinned one).  Never `value` in bench.py — quoted in DESIGN.md §5 only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
S, F, hop = 64, 256 * 1024, 256
cfg = capi.SpectrogramConfig(fft_size=4096, hop_size=hop, history_length=8192, use_reassignment=True)
rng = np.random.default_rng(1)
pageable = (rng.random((S, F, 2), dtype=np.float32) - 0.5)
pinned_t = torch.from_numpy(pageable).pin_memory()
pinned = pinned_t.numpy()
dev = pinned_t.cuda()
for label, run in (("device-resident", lambda b: b.process_device(dev.data_ptr(), F, 2, 48000.0, capi.positions_fallback(2))),
                   ("host pageable", lambda b: b.process_host(pageable, 2, 48000.0)),
                   ("host pinned", lambda b: b.process_host(pinned, 2, 48000.0))):
    bank = banks.SpectrogramBank(api, cfg, S)
    for _ in range(2):
        run(bank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        run(bank)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{label:16s}: {dt * 1e3:7.2f} ms/step -> {S * F / hop / dt / 1e6:6.2f} M frames/s  (PCM {S * F * 8 / 1e6:.0f} MB/step"
          + (f", {S * F * 8 / dt / 1e9:.1f} GB/s incl. compute)" if label != "device-resident" else ")"))
