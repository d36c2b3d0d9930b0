"""Phase breakdown of the oscilloscope kernel (run on the GPU box; sets OMX_SCOPE_PHASES=1)."""
import ctypes as C
import os
import sys

os.environ["OMX_SCOPE_PHASES"] = "1"
# the product library ignores tuning variables: the phase clocks live in the tuning build (make -C openmeters_amd/csrc TUNING=1)
os.environ.setdefault("OMX_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openmeters_amd", "csrc", "libomx_hip_tuning.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
dev = torch.device("cuda", 0)
S, blocks = 256, 64
frames = 256 * blocks
n = torch.arange(frames * 3, device=dev, dtype=torch.float64)
pcm = torch.empty((S, frames * 3, 2), device=dev, dtype=torch.float32)
for s in range(S):
    f = 440.0 * 2.0 ** ((s % 24) / 12.0)
    left = (0.8 * torch.sin(2 * np.pi * f * n / 48000.0)).to(torch.float32)
    pcm[s, :, 0] = left
    pcm[s, :, 1] = -0.7 * left
sc = banks.OscilloscopeBank(api, capi.OscilloscopeConfig(segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2,
                                                         trigger_source=capi.CH_LEFT, channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT), S)
pos = capi.positions_fallback(2)
f = api.fn("debug_scope_phase_cycles", C.c_int, [C.POINTER(C.c_uint64), C.c_uint32, C.c_int])
out = (C.c_uint64 * 10)()
sc.process_device(pcm[:, :frames].contiguous().data_ptr(), 256, blocks, 2, 48000.0, pos)
torch.cuda.synchronize()
f(out, 10, 1)
sc.process_device(pcm[:, frames:2 * frames].contiguous().data_ptr(), 256, blocks, 2, 48000.0, pos)
torch.cuda.synchronize()
f(out, 10, 1)
c = np.array(out[:], np.float64)
names = ["bookkeeping", "span load issued", "retune / template", "template statistics + span mean, work", "coarse-to-fine search",
         "candidate vs reference", "reference update", "snapshot + header", "  search: sweeps + barrier", "  search: scores + argmax"]
for name, v in zip(names, c):
    print(f"{name:44s} {v / c.sum() * 100:5.1f} %   {v / (S * blocks):9.0f} cycles/block")
print("total cycles/block", c.sum() / (S * blocks))
