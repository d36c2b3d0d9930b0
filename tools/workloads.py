"""Synthetic workloads of BASELINE.json / SURVEY §8(d), vectorised over streams (numpy, host side): what `bench.py` and
the tools feed to the banks.  Formula-only; `tests/test_cpu_workloads.py` checks them sample for sample against the scalar
definitions in `tests/signals.py` / `tests/golden_inputs.py`."""
from __future__ import annotations

import numpy as np

FS = 48000.0


def xorshift32_noise_bank(seeds, n, amplitude):
    """[len(seeds), n] f32: white noise from xorshift32 (x ^= x << 13; x ^= x >> 17; x ^= x << 5), uniform in
    [-amplitude, amplitude), one independent generator per seed (0 is replaced by 1) — SURVEY §8(d)."""
    x = np.asarray(seeds, dtype=np.uint32).copy()
    x[x == 0] = 1
    out = np.empty((n, x.shape[0]), dtype=np.float64)
    for i in range(n):
        x ^= x << np.uint32(13)
        x ^= x >> np.uint32(17)
        x ^= x << np.uint32(5)
        out[i] = x
    return ((out.T / 4294967296.0 * 2.0 - 1.0) * amplitude).astype(np.float32)


def exp_sweep_bank(phases, n, f0=20.0, f1=20000.0, seconds=10.0, amplitude=0.5):
    """[len(phases), n] f32: exponential sine sweep f0 -> f1 over `seconds`, per-stream start phase (f64, rounded)."""
    t = np.arange(n, dtype=np.float64) / FS
    k = np.log(f1 / f0)
    base = 2.0 * np.pi * f0 * seconds / k * (np.exp(t / seconds * k) - 1.0)
    return (amplitude * np.sin(base[None, :] + np.asarray(phases, dtype=np.float64)[:, None])).astype(np.float32)


def cfg2_bank(first_stream, n_streams, frames):
    """cfg2 / cfg5 generator: stream s = sweep with start phase 2 pi s / 64 + white noise -60 dBFS from
    xorshift32(0x9E3779B9 ^ s); R = 0.8 L.  Returns f32 [n_streams][frames][2]."""
    s = np.arange(first_stream, first_stream + n_streams, dtype=np.int64)
    left = exp_sweep_bank(2.0 * np.pi * s / 64.0, frames) + xorshift32_noise_bank((0x9E3779B9 ^ s) & 0xFFFFFFFF, frames, 1e-3)
    pcm = np.empty((n_streams, frames, 2), np.float32)
    pcm[:, :, 0] = left
    pcm[:, :, 1] = np.float32(0.8) * left
    return pcm


def cfg1_pcm(frames):
    """cfg1: 2 ch, sweep 20 Hz -> 20 kHz over 10 s, amplitude 0.5, R = 0.8 L.  f32 [frames][2]."""
    left = exp_sweep_bank([0.0], frames)[0]
    return np.stack([left, np.float32(0.8) * left], 1).astype(np.float32)
