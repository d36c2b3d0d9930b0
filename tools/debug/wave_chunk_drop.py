"""debug: the 120 dB drop sequence through the chunk-parallel waveform bank against the oracle, worst entries"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
from openmeters_amd import banks, capi
from openmeters_amd.capi import WaveformConfig, WaveformProcessor, AudioBlock, Api
import openmeters_amd
omx = openmeters_amd.api()
oracle = Api(os.path.join(os.path.dirname(openmeters_amd.__file__), "..", "oracle", "libomx_oracle.so"), "omxo_")
FS = 48000.0
cfg = WaveformConfig(scroll_speed=300.0, max_columns=1024, analyze_bands=True, track_history=True)
n = 16384
rng = np.random.default_rng(5)
loud = rng.uniform(-1.0, 1.0, (3 * n, 2)).astype(np.float32)
quiet = (rng.uniform(-1.0, 1.0, (4 * n, 2)) * 1e-6).astype(np.float32)
pcm = np.concatenate([loud, quiet])[None]
form = int(sys.argv[1]) if len(sys.argv) > 1 else 2
bank = banks.WaveformBank(omx, cfg, 1)
bank.set_option(capi.OPT_KERNEL_FORM, form)
ref = WaveformProcessor(oracle, cfg)
for k in range(0, pcm.shape[1], n):
    up = bank.process_host(pcm[:, k:k + n], 2, FS)
    w = ref.process_block(AudioBlock(pcm[0, k:k + n].reshape(-1), 2, FS))
    got, _ = bank.fetch(0, int(up.n_columns))
    g, o = got[:, :, 5:].reshape(-1, 4, 2, 3).astype(np.float64), w.columns[:, :, 5:].reshape(-1, 4, 2, 3).astype(np.float64)
    pg, pw = 10 ** (g / 10), 10 ** (o / 10)
    top = pw.max(axis=1, keepdims=True)
    rel = np.abs(pg - pw) / top
    i = np.unravel_index(rel.argmax(), rel.shape)
    print(k, "worst rel", rel.max(), "at (col, ch, win, band)", i, "dB got/want", g[i], o[i], "top dB", 10 * np.log10(top[i[0], 0, i[2], i[3]]))
    c = np.abs(got[:, :, 2:5] - w.columns[:, :, 2:5])
    j = np.unravel_index(c.argmax(), c.shape)
    print("   colour worst", c.max(), j, got[:, :, 2:5][j], w.columns[:, :, 2:5][j])
