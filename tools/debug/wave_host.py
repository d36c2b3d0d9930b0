"""waveform bank, 1024 streams x 16384 frames (chunk-parallel form): how long the HOST spends inside a call (enqueue only) against the
device time per call — a call whose host side is longer than its kernels is launch-bound, not kernel-bound"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
S, frames = 1024, 16384
for history in (False, True):
    cfg = capi.WaveformConfig(scroll_speed=300.0, max_columns=1024, analyze_bands=True, track_history=history)
    bank = banks.WaveformBank(api, cfg, S)
    pcm = [(torch.rand((S, frames, 2), device="cuda:0") - 0.5).contiguous() for _ in range(2)]
    pos = capi.positions_fallback(2)
    for k in range(4):
        bank.process_device(pcm[k & 1].data_ptr(), frames, 2, 48000.0, pos)
    torch.cuda.synchronize()
    reps = 30
    host = []
    t0 = time.perf_counter()
    for k in range(reps):
        h0 = time.perf_counter()
        bank.process_device(pcm[k & 1].data_ptr(), frames, 2, 48000.0, pos)
        host.append(time.perf_counter() - h0)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    host.sort()
    print(f"history={int(history)}: host inside a call median {host[len(host) // 2] * 1e6:.0f} us (min {host[0] * 1e6:.0f}); all {reps} calls enqueued after {t_enq * 1e3:.2f} ms, "
          f"finished after {t_all * 1e3:.2f} ms = {t_all / reps * 1e3:.3f} ms per call; form {bank.last_form()}")
