"""Debug aid: classic columns of test_spectrogram_random_operation_sequences for one seed — prints the bins more than one code apart (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import conftest, openmeters_amd
from openmeters_amd import capi
from openmeters_amd.capi import Api, AudioBlock, SpectrogramConfig, SpectrogramProcessor
import test_gpu_state_machine as t
omx = openmeters_amd.api(); oracle = Api(conftest._build_oracle(), "omxo_")
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
sizes = [256, 512, 1024, 2048, 4096]
cfg = SpectrogramConfig(fft_size=int(rng.choice(sizes)), hop_size=int(rng.choice([64, 100, 256, 777])),
                        use_reassignment=bool(rng.integers(2)), history_length=int(rng.choice([3, 64, 8192])))
a, b = SpectrogramProcessor(omx, cfg), SpectrogramProcessor(oracle, cfg)
prng = np.random.default_rng(seed + 7919)
rate, channels, t0 = 48000.0, 2, 0
for step in range(45):
    op = rng.random()
    if op < 0.08:
        cfg = SpectrogramConfig(sample_rate=rate, fft_size=int(rng.choice(sizes)), hop_size=int(rng.choice([64, 100, 256, 777])),
                                window=int(rng.integers(5)), use_reassignment=bool(rng.integers(2)),
                                zero_padding_factor=int(rng.choice([1, 1, 1, 2])), history_length=int(rng.choice([3, 64, 8192])))
        a.update_config(cfg); b.update_config(cfg); continue
    if op < 0.12:
        a.reset_audio(); b.reset_audio(); continue
    if op < 0.16:
        rate = float(rng.choice([44100.0, 48000.0, 96000.0]))
    if op < 0.20:
        channels = int(rng.choice([1, 2, 6]))
    frames = int(rng.choice([0, 1, 37, 256, 256, 1024, 3000, 9000]))
    silent = rng.random() < 0.15
    pcm = t.signal(rng, frames, channels, t0, rate, silent=silent)
    t.ulp_perturbed(pcm, prng)
    t0 += frames
    blk = AudioBlock(pcm.reshape(-1), channels, rate)
    g, w = a.process_block(blk), b.process_block(blk)
    if w is None or not w.new_columns or w.new_columns[0].ndim == 2:
        continue
    for c, (h, o) in enumerate(zip(g.new_columns, w.new_columns)):
        d = np.abs(h.astype(np.int64) - o.astype(np.int64))
        if d.max() >= 2:
            db = o.astype(np.float64) * (156.0 / 65535.0) - 144.0
            idx = np.flatnonzero(d >= 2)
            print("step", step, "cfg", cfg, "rate", rate, "ch", channels, "frames", frames, "silent", silent, "column", c, "of", len(w.new_columns),
                  "max dB", round(db.max(), 2), "bins >= 2 codes:", [(int(i), int(h[i]), int(o[i]), round(db[i], 2)) for i in idx[:12]], "count", len(idx))
