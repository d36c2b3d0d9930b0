"""A failing seed of tests/test_gpu_parity_meters.py::test_waveform_chunk_parallel_random_sequences, taken apart (run on the GPU box):
per call, the fast / slow history powers of the chunk-parallel bank, the sequential bank, the oracle and the exact f64 recurrence at the
column where |HIP - exact| is largest.   python tools/debug/wave_seed.py <seed>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import openmeters_amd
from openmeters_amd import banks, capi
from openmeters_amd.capi import AudioBlock, WaveformConfig, WaveformProcessor
import test_gpu_parity_meters as m

seed = int(sys.argv[1])
omx = openmeters_amd.api()
oracle = capi.Api(os.path.join(ROOT, "oracle", "libomx_oracle.so"), "omxo_")
rng = np.random.default_rng(seed)
rate = float(rng.choice([22050.0, 32000.0, 44100.0, 48000.0, 96000.0]))
scroll = float(rng.choice([10.0, 77.7, 300.0, 650.0, 1000.0]))
history = bool(rng.integers(2))
sizes = [int(x) for x in rng.choice([1024, 1536, 2048, 4096, 6000, 8192, 12288, 256, 1000, 3001], size=8)]
S, total_frames = 3, sum(sizes)
pcm = np.stack([m.cfg4_pcm(int(seed % 1000) * 3 + s, total_frames) for s in range(S)])
pcm[2, :, 1] = pcm[2, :, 0] * np.float32(0.994) + pcm[2, :, 1] * np.float32(0.003)
at = 0
steps = []
while at < total_frames:
    n = int(rng.integers(500, 6000))
    g = np.float32(10.0 ** float(rng.uniform(-4.0, 0.0)))
    pcm[:, at:at + n] *= g
    steps.append((at, float(g)))
    at += n
print(f"seed {seed}: rate {rate}, scroll {scroll}, history {history}, sizes {sizes}")
print("level steps (frame, gain):", [(a, round(20 * np.log10(g), 1)) for a, g in steps])
cfg = WaveformConfig(sample_rate=rate, scroll_speed=scroll, max_columns=4096, analyze_bands=True, track_history=history)
exact = [m.WaveExact(pcm[s], rate, scroll) for s in range(S)]
print("column ends (stream 0):", exact[0].ends.tolist())
b2, b1 = banks.WaveformBank(omx, cfg, S), banks.WaveformBank(omx, cfg, S)
b2.set_option(capi.OPT_KERNEL_FORM, 2)
b1.set_option(capi.OPT_KERNEL_FORM, 1)
refs = [WaveformProcessor(oracle, cfg) for _ in range(S)]
at = total = 0
for n in sizes:
    chunk = pcm[:, at:at + n]
    at += n
    u2, u1 = b2.process_host(chunk, 2, rate), b1.process_host(chunk, 2, rate)
    form = b2.last_form()
    for s in range(S):
        w = refs[s].process_block(AudioBlock(chunk[s].reshape(-1), 2, rate))
        g2, _ = b2.fetch(s, int(u2.n_columns))
        g1, _ = b1.fetch(s, int(u1.n_columns))
        if not len(g2) or not history:
            continue
        cols = slice(total, total + len(g2))
        e_power = np.maximum(exact[s].power[cols], 1e-14)
        e_top = np.broadcast_to(np.maximum(exact[s].top_power[cols], 1e-14), e_power.shape)
        for name, g in (("chunk", g2), ("seq", g1), ("oracle", w.columns)):
            p = 10.0 ** (np.asarray(g, np.float64)[:, :, 5:].reshape(-1, 4, 2, 3) / 10.0)
            d = np.abs(p - e_power) / e_top
            k = np.unravel_index(np.argmax(d), d.shape)
            print(f"call n={n:6d} form {form} stream {s} {name:6s}: max |p - exact| / top = {d.max():.3e} at column {total + k[0]} (end {exact[s].ends[total + k[0]]}) channel {k[1]} window {k[2]} band {k[3]}; "
                  f"p = {p[k]:.6e} exact {e_power[k]:.6e} top {e_top[k]:.3e}")
    total += int(u2.n_columns)
