"""Phase breakdown of the fused STFT kernel (run on the GPU box with OMX_K2_VARIANT=7)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMX_K2_VARIANT", "9")  # 9 = phase-timing build of the default kernel, 7 = of variant 12
import numpy as np
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
S, cols = 64, 1024
frames = 8192 + 256 * (cols - 1)
pcm = (torch.rand((S, frames, 2), device="cuda:0") - 0.5).contiguous()
bank = banks.SpectrogramBank(api, capi.SpectrogramConfig(fft_size=4096, hop_size=256, history_length=8192), S)
f = api.fn("debug_k2_phase_cycles", C.c_int, [C.POINTER(C.c_uint64), C.c_uint32, C.c_int])
out = (C.c_uint64 * 12)()
bank.process_device(pcm.data_ptr(), frames, 2, 48000.0, capi.positions_fallback(2))
torch.cuda.synchronize()
f(out, 12, 1)
for _ in range(5):
    bank.process_device(pcm[:, :256 * cols].contiguous().data_ptr(), 256 * cols, 2, 48000.0, capi.positions_fallback(2))
torch.cuda.synchronize()
f(out, 12, 1)
c = np.array(out[:], np.float64)
names = ["setup+load", "fwd FFT", "Hilbert build (rest)", "inverse FFT", "gather+window", "dual FFT", "third FFT", "compaction+store",
         "  Hilbert: spectrum write + barrier", "  reassign arithmetic + ballots", "  count barrier", "-"]
for n, v in zip(names, c):
    print(f"{n:16s} {v / c.sum() * 100:5.1f} %   {v / (5 * S * cols):9.0f} cycles/frame")
print("total cycles/frame", c.sum() / (5 * S * cols))
