"""Race detector for the fused kernels (run on the GPU box): every shape processes the SAME input many times from a fresh bank and the
outputs (point counts, points / codes, traces) must be bit-identical run to run — a missing barrier shows up as a run that differs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
S = int(sys.argv[2]) if len(sys.argv) > 2 else 24
pos = capi.positions_fallback(2)
bad = 0


def dev(ptr, shape, typestr):
    class V:
        __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}
    return torch.as_tensor(V(), device="cuda:0")


shapes = [(4096, 256, 1, True, capi.WINDOW_HANN), (1024, 256, 1, True, capi.WINDOW_HANN), (2048, 64, 1, True, capi.WINDOW_HAMMING),
          (8192, 512, 1, True, capi.WINDOW_HANN), (16384, 1024, 1, True, capi.WINDOW_HANN), (2048, 256, 2, True, capi.WINDOW_HANN),
          (4096, 512, 2, True, capi.WINDOW_HAMMING), (1024, 256, 16, True, capi.WINDOW_HANN), (4096, 1024, 4, True, capi.WINDOW_HANN),
          (8192, 1024, 2, True, capi.WINDOW_HANN), (2048, 256, 1, True, capi.WINDOW_BLACKMAN), (8192, 512, 1, True, capi.WINDOW_BLACKMAN),
          (16384, 2048, 1, True, capi.WINDOW_BLACKMAN_HARRIS), (4096, 256, 1, False, capi.WINDOW_HANN), (1000, 250, 1, True, capi.WINDOW_HANN)]
for W, hop, zp, reassign, window in shapes:
    cols = 24 if W * zp >= 8192 else 64
    H = 2 * (1 << int(np.ceil(np.log2(W)))) if reassign else W * zp
    frames = H + hop * (cols - 1) + 1   # + 1: an odd second call start below
    g = torch.Generator(device="cuda:0").manual_seed(W + hop + zp)
    pcm = (torch.rand((S, frames + hop * cols, 2), device="cuda:0", generator=g) - 0.5).contiguous()
    ref = None
    for rep in range(REPS):
        bank = banks.SpectrogramBank(api, capi.SpectrogramConfig(fft_size=W, hop_size=hop, zero_padding_factor=zp, window=window,
                                                                  use_reassignment=reassign, history_length=8192), S)
        outs = []
        for lo, hi in ((0, frames), (frames, frames + hop * cols)):
            up = bank.process_device(pcm[:, lo:hi].contiguous().data_ptr(), hi - lo, 2, 48000.0, pos)
            torch.cuda.synchronize()
            if up is None:
                continue
            n = int(up.n_columns)
            if reassign:
                cnt = dev(up.d_counts, (S, n), "<i4").clone()
                pts = dev(up.d_points, (S, n, int(up.column_stride), 3), "<i4").clone()
                mask = torch.arange(int(up.column_stride), device="cuda:0")[None, None, :] < cnt[:, :, None]
                outs += [cnt, torch.where(mask[..., None], pts, torch.zeros_like(pts))]
            else:
                outs.append(dev(up.d_codes, (S, n, int(up.column_stride)), "<i2").clone())
        assert outs and all(int(o.numel()) > 0 for o in outs) and int(outs[0].max()) > 0, "the shape produced nothing"
        if ref is None:
            ref = outs
        else:
            same = len(outs) == len(ref) and all(torch.equal(a, b) for a, b in zip(outs, ref))
            if not same:
                bad += 1
                print(f"NONDETERMINISTIC: W={W} hop={hop} zp={zp} reassign={reassign} window={window} rep={rep}")
                break
        del bank
    print(f"spectrogram W={W} hop={hop} zp={zp} reassign={reassign} window={window}: {REPS} runs compared")

for N, hop in ((4096, 256), (8192, 512), (16384, 1024), (1000, 250)):
    hops = 24
    frames = N + hop * (hops - 1)
    pcm = (torch.rand((S, frames, 2), device="cuda:0") - 0.5).contiguous()
    ref = None
    for rep in range(REPS):
        bank = banks.SpectrumBank(api, capi.SpectrumConfig(fft_size=N, hop_size=hop), S, emit_all_hops=True)
        up = bank.process_device(pcm.data_ptr(), frames, 2, 48000.0, pos)
        torch.cuda.synchronize()
        t = dev(up.d_traces, (S, int(up.n_hops_out), 4, int(up.bins)), "<i4").clone()
        if ref is None:
            ref = t
        elif not torch.equal(t, ref):
            bad += 1
            print(f"NONDETERMINISTIC: spectrum N={N} rep={rep}")
            break
        del bank
    print(f"spectrum N={N} hop={hop}: {REPS} runs compared")
print("RESULT:", "all deterministic" if bad == 0 else f"{bad} shapes differ")
sys.exit(1 if bad else 0)
