"""Per-call latency of the single-stream handles (host PCM in, host snapshot out) at the DspBatcher quantum (256 frames),
i.e. what a GUI-side drop-in pays per block.  Run on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import openmeters_amd
from openmeters_amd import capi
from openmeters_amd.capi import (AudioBlock, LoudnessConfig, LoudnessProcessor, OscilloscopeConfig, OscilloscopeProcessor,
                                 SpectrogramConfig, SpectrogramProcessor, SpectrumConfig, SpectrumProcessor, StereometerConfig,
                                 StereometerProcessor, WaveformConfig, WaveformProcessor)

api = openmeters_amd.api()
rng = np.random.default_rng(0)
t = np.arange(256 * 600) / 48000.0
x = (0.4 * np.sin(2 * np.pi * 440.0 * t) + 0.001 * rng.standard_normal(t.size)).astype(np.float32)
pcm = np.stack([x, -0.7 * x], 1)
procs = {
    "spectrogram 2048/64 reassigned (reference default)": SpectrogramProcessor(api, SpectrogramConfig(fft_size=2048, hop_size=64)),
    "spectrogram 4096/256 reassigned": SpectrogramProcessor(api, SpectrogramConfig(fft_size=4096, hop_size=256)),
    "spectrum 16384/1024 (reference default)": SpectrumProcessor(api, SpectrumConfig(fft_size=16384, hop_size=1024)),
    "loudness": LoudnessProcessor(api, LoudnessConfig()),
    "stereometer": StereometerProcessor(api, StereometerConfig(analyze_bands=True)),
    "oscilloscope": OscilloscopeProcessor(api, OscilloscopeConfig()),
    "waveform": WaveformProcessor(api, WaveformConfig()),
}
for name, p in procs.items():
    lat = []
    for b in range(600):
        blk = AudioBlock(pcm[b * 256:(b + 1) * 256].reshape(-1), 2, 48000.0)
        t0 = time.perf_counter()
        p.process_block(blk)
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat[100:]) * 1e6
    print(f"{name:52s} median {np.median(lat):7.1f} us   p99 {np.percentile(lat, 99):7.1f} us   ({5333.3 / np.median(lat):5.1f}x the 5.33 ms block period)")
