import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import openmeters_amd
from openmeters_amd import capi
from openmeters_amd.capi import AudioBlock, OscilloscopeConfig, OscilloscopeProcessor
api = openmeters_amd.api()
oracle = capi.Api(os.path.join(ROOT, "oracle", "libomx_oracle.so"), "omxo_")
cfg = OscilloscopeConfig(segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2, trigger_source=capi.CH_LEFT, channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT)
n = 256 * 60
t = np.arange(n) / 48000.0
left = (0.8 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
pcm = np.stack([left, -0.7 * left], 1).astype(np.float32)
a, b = OscilloscopeProcessor(api, cfg), OscilloscopeProcessor(oracle, cfg)
for k in range(0, n, 256):
    blk = AudioBlock(pcm[k:k + 256].reshape(-1), 2, 48000.0)
    g, w = a.process_block(blk), b.process_block(blk)
    if g is None or w is None:
        print(k // 256, g is None, w is None)
        continue
    ca, cb = a.last_capture(), b.last_capture()
    print(k // 256, ca, cb, a.last_cycle_rate(), b.last_cycle_rate(), float(np.abs(g.samples - w.samples).max()) if g.samples.shape == w.samples.shape else "shape")
