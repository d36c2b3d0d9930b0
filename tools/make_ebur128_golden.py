"""Independent BS.1770 / EBU R128 fixtures for the two reference tests that compare against the `ebur128` crate
(reference src/visuals/loudness/processor.rs:366-398 LUFS-S, :426-454 true peak).  `ebur128 0.1.10` is an un-vendored
dev-dependency (a Rust port of libebur128) and cannot run here, so this script restates libebur128's PUBLISHED algorithm in
scipy — a different code path from both the oracle and the product (library IIR / FIR routines, f64 throughout):

  * K-weighting (libebur128 `ebur128_init_filter`): high shelf f0 = 1681.974450955533 Hz, G = 3.999843853973347 dB,
    Q = 0.7071752369554196 (Vb = Vh^0.4996667741545416) cascaded with the RLB high-pass f0 = 38.13547087602444 Hz,
    Q = 0.5003270373238773, both by bilinear transform with K = tan(pi f0 / fs); at 48 kHz this reproduces the BS.1770-4
    coefficient table (asserted below).
  * short-term loudness: -0.691 + 10 log10( sum_c G_c * mean(y_c^2 over the newest 30 * samples_in_100ms frames) ),
    G = 1.0 for L / R / C, 1.41 for the surrounds, 0 for LFE; default channel maps of 2 / 4 / 5 / 6 channels
    (L R | L R Ls Rs | L R C Ls Rs | L R C LFE Ls Rs).
  * true peak (libebur128 `interp_create(49, factor)`): 49-tap Hann-windowed sinc, polyphase split by `j % factor`, taps with
    |c| <= 1e-6 dropped; factor 4 below 96 kHz, 2 below 192 kHz, none from 192 kHz up; peak = max |phase outputs| (phase 0 is
    the sample itself); dBTP = 20 log10(peak).

Inputs are the reference tests' formula signals (`util/audio.rs:28-33` sine in f32).  Output: tests/golden/ebur128_scipy.json.
Run:  python tools/make_ebur128_golden.py
"""
import json
import os
import sys

import numpy as np
from scipy.signal import lfilter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from signals import sine_wave  # noqa: E402


def k_weighting(fs):
    f0, G, Q = 1681.974450955533, 3.999843853973347, 0.7071752369554196
    K = np.tan(np.pi * f0 / fs)
    Vh = 10.0 ** (G / 20.0)
    Vb = Vh ** 0.4996667741545416
    a0 = 1.0 + K / Q + K * K
    pb = np.array([(Vh + Vb * K / Q + K * K) / a0, 2.0 * (K * K - Vh) / a0, (Vh - Vb * K / Q + K * K) / a0])
    pa = np.array([1.0, 2.0 * (K * K - 1.0) / a0, (1.0 - K / Q + K * K) / a0])
    f0, Q = 38.13547087602444, 0.5003270373238773
    K = np.tan(np.pi * f0 / fs)
    rb = np.array([1.0, -2.0, 1.0])
    ra = np.array([1.0, 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K), (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K)])
    return np.convolve(pb, rb), np.convolve(pa, ra), (pb, pa, ra)


def channel_weights(channels):
    L, S, U = 1.0, 1.41, 0.0
    return {2: [L, L], 4: [L, L, S, S], 5: [L, L, L, S, S], 6: [L, L, L, U, S, S]}[channels]


def short_term_lufs(mono, channels, fs):
    b, a, _ = k_weighting(float(fs))
    y = lfilter(b, a, mono.astype(np.float64))
    interval = ((int(fs) + 5) // 10) * 30
    assert len(y) >= interval
    ms = float(np.mean(y[-interval:] ** 2))
    return -0.691 + 10.0 * np.log10(sum(channel_weights(channels)) * ms)


def true_peak_db(x, fs):
    factor = 4 if fs < 96000 else (2 if fs < 192000 else 1)
    x = x.astype(np.float64)
    peak = float(np.abs(x).max())
    if factor > 1:
        taps = 49
        j = np.arange(taps)
        m = j - (taps - 1) / 2.0
        with np.errstate(invalid="ignore", divide="ignore"):
            c = np.where(m == 0, 1.0, np.sin(m * np.pi / factor) / (m * np.pi / factor))
        c = c * 0.5 * (1.0 - np.cos(2.0 * np.pi * j / (taps - 1)))
        c[np.abs(c) <= 1e-6] = 0.0
        for f in range(factor):
            sub = c[f::factor]                      # tap t of phase f = c[f + t * factor]
            peak = max(peak, float(np.abs(lfilter(sub, [1.0], x)).max()))
    return 20.0 * np.log10(peak)


def main():
    b, a, (pb, pa, ra) = k_weighting(48000.0)
    assert np.allclose(pb, [1.53512485958697, -2.69169618940638, 1.19839281085285], atol=1e-12)
    assert np.allclose(pa, [1.0, -1.69065929318241, 0.73248077421585], atol=1e-12)
    assert np.allclose(ra, [1.0, -1.99004745483398, 0.99007225036621], atol=1e-12)
    out = {"generator": "tools/make_ebur128_golden.py (scipy restatement of libebur128's published algorithm)",
           "short_term": [], "true_peak": []}
    for fs in (44100.0, 48000.0, 96000.0):          # processor.rs:366-398
        mono = sine_wave(1000.0, fs, int(np.float32(fs) * np.float32(4.0)), 0.5)
        for channels in (2, 4, 5, 6):
            out["short_term"].append({"sample_rate": fs, "channels": channels, "seconds": 4.0, "freq": 1000.0, "amp": 0.5,
                                      "lufs_s": short_term_lufs(mono, channels, fs)})
    for fs in (48000.0, 96000.0, 192000.0):         # processor.rs:426-454
        x = sine_wave(17000.0, fs, int(np.float32(fs) * np.float32(0.01)), 0.9)
        out["true_peak"].append({"sample_rate": fs, "seconds": 0.01, "freq": 17000.0, "amp": 0.9, "dbtp": true_peak_db(x, fs)})
    # a second true-peak family: inter-sample peaks of fs/4 with a 45 degree phase (classic +3 dB case) and a full 0.5 s
    for fs in (44100.0, 48000.0, 96000.0):
        n = np.arange(int(fs * 0.05))
        x = (0.8 * np.sin(2.0 * np.pi * 0.25 * n + np.pi / 4.0)).astype(np.float32)
        out["true_peak"].append({"sample_rate": fs, "seconds": 0.05, "freq": fs / 4.0, "amp": 0.8, "phase": "pi/4", "dbtp": true_peak_db(x, fs)})
    path = os.path.join(ROOT, "tests", "golden", "ebur128_scipy.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    for r in out["short_term"] + out["true_peak"]:
        print(r)


if __name__ == "__main__":
    main()
