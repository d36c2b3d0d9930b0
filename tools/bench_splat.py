"""cfg2-sized splat accumulation: 64 streams x 1024 reassigned columns -> [64][1024][512] f32 images (time-major) (run on the GPU box)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
S, cols = 64, 1024
frames = 8192 + 256 * (cols - 1)
t = torch.arange(frames, device="cuda:0", dtype=torch.float32)
pcm = (0.4 * torch.sin(t * 0.05 + 1e-6 * t * t)[None, :, None] + 0.001 * torch.randn((S, frames, 2), device="cuda:0")).contiguous()
bank = banks.SpectrogramBank(api, capi.SpectrogramConfig(fft_size=4096, hop_size=256, history_length=8192), S)
up = bank.process_device(pcm.data_ptr(), frames, 2, 48000.0, capi.positions_fallback(2))
f = api.fn("spectrogram_splat", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p])
for sf in (1.0, 2.0):
    view = capi.splat_view(api, 1024.0 * sf, 512.0 * sf, scale_factor=sf)
    acc = torch.empty((S, view.width, view.height), device="cuda:0", dtype=torch.float32)
    db = torch.empty_like(acc)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(6):
        if it == 1:
            ev[0].record()
        api.check(f(up.d_points, up.d_counts, 1, S, cols, up.column_stride, up.reassigned_power_scale, C.byref(view), None, acc.data_ptr(),
                    db.data_ptr()))
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 5
    n_points = S * cols * 2047
    bytes_moved = n_points * 12 + 3 * acc.numel() * 4
    print(f"scale_factor {sf}: {ms:.3f} ms per {S * cols} columns -> {S * cols / ms / 1e3:.1f} M columns/s, "
          f"{n_points * sf * sf / ms / 1e6:.1f} G atomics/s, ~{bytes_moved / ms / 1e6:.0f} GB/s of points + image traffic")
