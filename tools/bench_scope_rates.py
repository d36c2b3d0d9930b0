"""Oscilloscope bank per sample rate (run on the GPU box): 256 streams x 2 ch, 64 batcher blocks per call.  27.3 ... 54.6 kHz run the wide
form (scope_fast_kernels.hip, 8192-point autocorrelation); 54.6 ... 218 kHz the big estimate kernel + the wide trigger pass on capped LDS
with hand-over (round 4); other rates the single-pass kernel of round 1.  `rates()` feeds `secondary.oscilloscope_rates` of bench.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import openmeters_amd
from openmeters_amd import banks, capi


def rates(which=(44100.0, 48000.0, 88200.0, 96000.0, 192000.0), S=256, blocks=64, out=sys.stdout):
    api = openmeters_amd.api()
    dev = torch.device("cuda", 0)
    res = {}
    for fs in which:
        block = int(round(256 * fs / 48000.0))
        frames = block * blocks
        n = torch.arange(frames, device=dev, dtype=torch.float64)
        semis = torch.arange(S, device=dev, dtype=torch.float64) % 24
        f = 440.0 * 2.0 ** (semis / 12.0)
        left = (0.8 * torch.sin(2 * np.pi * f[:, None] * n[None, :] / fs)).to(torch.float32)
        pcm = torch.stack([left, -0.7 * left], dim=2).contiguous()
        pos = capi.positions_fallback(2)
        sc = banks.OscilloscopeBank(api, capi.OscilloscopeConfig(sample_rate=fs, segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2,
                                                                 trigger_source=capi.CH_LEFT, channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT), S)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            sc.process_device(pcm.data_ptr(), block, blocks, 2, fs, pos, stream)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(3):
            sc.process_device(pcm.data_ptr(), block, blocks, 2, fs, pos, stream)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 3
        hdr, _ = sc.fetch(0, blocks - 1)
        print(f"oscilloscope {fs:.0f} Hz, block {block}: {ms:.2f} ms per {S} x {blocks} blocks -> {S * blocks / ms / 1e3:.2f} M blocks/s, "
              f"{frames / fs / (ms / 1e3):.0f}x real time; locked={hdr.locked} period={hdr.period:.2f}", file=out)
        res[f"{fs:.0f}_hz"] = {"block_frames": block, "ms_per_call": ms, "blocks_per_s": S * blocks / (ms * 1e-3), "x_real_time": frames / fs / (ms * 1e-3)}
        sc.close()
    return {"workload": f"{S} streams x 2 ch, {blocks} batcher blocks per call, Stable trigger, 2 cycles", **res}


if __name__ == "__main__":
    rates()
