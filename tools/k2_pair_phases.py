"""Phase breakdown of the shipped pair kernel (tuning build, OMX_K2_VARIANT=52): run on the GPU box as
OMX_HIP_LIB=$PWD/openmeters_amd/csrc/libomx_hip_tuning.so python tools/k2_pair_phases.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OMX_K2_VARIANT"] = "52"
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
reps = 5
for _ in range(reps):
    bank.process_device(pcm[:, :256 * cols].contiguous().data_ptr(), 256 * cols, 2, 48000.0, capi.positions_fallback(2))
torch.cuda.synchronize()
f(out, 12, 1)
c = np.array(out[:], np.float64)
names = ["setup, window loads issued", "forward dual (waits for the loads)", "  barrier before the inverse", "inverse dual", "imag gather + slices",
         "windowed duals (both columns)", "bins, reassignment, compaction, stores (both)", "  natural-order copy + barrier", "  partner reads + Hilbert spectra",
         "  real-part loads issued", "  barrier after the forward dual", "-"]
pairs = reps * S * cols / 2
for n, v in zip(names, c):
    print(f"{n:48s} {v / c.sum() * 100:5.1f} %   {v / pairs:9.0f} cycles per pair")
print("total cycles per pair (thread 0's clock, 100 MHz-independent shader clock)", c.sum() / pairs)
