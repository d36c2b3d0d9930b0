"""Prints parity statistics of the HIP spectrogram against the CPU oracle (run on the GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import openmeters_amd  # noqa: E402
from openmeters_amd import capi  # noqa: E402
from openmeters_amd.capi import AudioBlock, SpectrogramConfig, SpectrogramProcessor  # noqa: E402
from parity import classic_column_metrics, reassigned_column_metrics  # noqa: E402
from signals import exp_sweep, xorshift32_noise  # noqa: E402

api = openmeters_amd.api()
oracle = capi.Api(os.path.join(ROOT, "oracle", "libomx_oracle.so"), "omxo_")
print("device available:", openmeters_amd.device_available())


def stream_pcm(s, n):
    left = exp_sweep(n, phase0=2 * np.pi * s / 64) + xorshift32_noise(0x9E3779B9 ^ s, n, 1e-3)
    return np.stack([left, 0.8 * left], 1).reshape(-1).astype(np.float32)


for (W, hop, zp, reassign, ncols) in [(4096, 256, 1, True, 24), (1024, 256, 1, True, 16), (2048, 512, 4, True, 4),
                                     (1024, 256, 1, False, 16), (4096, 256, 1, False, 16), (256, 32, 1, True, 8)]:
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, zero_padding_factor=zp, use_reassignment=reassign,
                            history_length=8192)
    read_len = 2 * W if reassign else W
    n = read_len + hop * (ncols - 1)
    worst = {}
    for s in (0, 17):
        pcm = stream_pcm(s, n + 20000)[2 * 20000:]
        got = SpectrogramProcessor(api, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
        want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
        assert len(got.new_columns) == len(want.new_columns) == ncols, (len(got.new_columns), len(want.new_columns))
        for h, o in zip(got.new_columns, want.new_columns):
            m = reassigned_column_metrics(h, o, 48000.0, hop) if reassign else classic_column_metrics(h, o)
            for k, v in m.items():
                worst[k] = max(worst.get(k, 0), v)
    print(f"W={W} hop={hop} zp={zp} reassign={reassign}: {worst}")
