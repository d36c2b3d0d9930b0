"""Formula-only generators of the BASELINE.json / SURVEY §8(d) synthetic workloads (shared by the golden
fixture generator and the tests that replay them)."""
import numpy as np

from signals import exp_sweep, xorshift32_noise

FS = 48000.0


def cfg1_pcm(frames):
    """cfg1: 2 ch, exponential sweep 20 Hz -> 20 kHz over 10 s, amplitude 0.5, R = 0.8 L."""
    left = exp_sweep(frames)
    return np.stack([left, np.float32(0.8) * left], 1).astype(np.float32)


def cfg2_pcm(s, frames):
    """cfg2: stream s = sweep with start phase 2*pi*s/64 + white noise -60 dBFS from xorshift32(0x9E3779B9 ^ s)."""
    left = exp_sweep(frames, phase0=2 * np.pi * s / 64) + xorshift32_noise(0x9E3779B9 ^ s, frames, 1e-3)
    return np.stack([left, np.float32(0.8) * left], 1).astype(np.float32)


def cfg3_pcm(s, frames, channels=8):
    """cfg3: channel c = 0.5 sin(2 pi (997 + 10 c + 0.01 s) n / fs), LFE (index 3) at 60 Hz."""
    n = np.arange(frames, dtype=np.float64)
    out = np.empty((frames, channels), np.float32)
    for c in range(channels):
        f = 60.0 if c == 3 else 997.0 + 10.0 * c + 0.01 * s
        out[:, c] = (0.5 * np.sin(2 * np.pi * f * n / FS)).astype(np.float32)
    return out


def cfg4_pcm(s, frames):
    """cfg4: L = 440*2^((s mod 24)/12) Hz saw / sine / square by s mod 3 (0.8 amp), R = -0.7 L + noise -40 dBFS."""
    f = 440.0 * 2.0 ** ((s % 24) / 12.0)
    c = f * np.arange(frames, dtype=np.float64) / FS
    fr = c - np.floor(c)
    kind = s % 3
    left = (2.0 * fr - 1.0) if kind == 0 else (np.sin(2 * np.pi * c) if kind == 1 else np.where(fr < 0.5, 1.0, -1.0))
    left = (0.8 * left).astype(np.float32)
    right = (-0.7 * left + xorshift32_noise(0x9E3779B9 ^ s, frames, 1e-2)).astype(np.float32)
    return np.stack([left, right], 1)
