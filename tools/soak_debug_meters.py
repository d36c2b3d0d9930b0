"""Debug aid: the spectrum leg of test_meter_processors_random_block_sequences for one seed, printing where HIP and oracle traces differ (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import conftest, openmeters_amd
from openmeters_amd import capi
from openmeters_amd.capi import Api, AudioBlock, SpectrumConfig, SpectrumProcessor
import test_gpu_state_machine as t
omx = openmeters_amd.api(); oracle = Api(conftest._build_oracle(), "omxo_")
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
sc = SpectrumConfig(fft_size=int(rng.choice([512, 1024, 4096])), hop_size=int(rng.choice([128, 256, 1000])),
                    averaging_mode=int(rng.integers(3)), averaging_param=float(rng.choice([0.5, 0.9, 12.0])),
                    source=capi.CH_LEFT, secondary_source=capi.CH_SIDE)
if sc.averaging_mode == capi.AVG_EXPONENTIAL:
    sc.averaging_param = 0.7
print("config", sc)
a, b = SpectrumProcessor(omx, sc), SpectrumProcessor(oracle, sc)
rate, channels, t0 = 48000.0, 2, 0
for step in range(40):
    op = rng.random()
    if op < 0.06:
        a.reset_audio(); b.reset_audio(); print(step, "reset"); continue
    if op < 0.10:
        rate = float(rng.choice([44100.0, 48000.0, 96000.0]))
    if op < 0.14:
        channels = int(rng.choice([1, 2, 6]))
    frames = int(rng.choice([0, 1, 100, 256, 256, 960, 2048, 5000]))
    silent = rng.random() < 0.15
    pcm = t.signal(rng, frames, channels, t0, rate, silent=silent)
    t0 += frames
    blk = AudioBlock(pcm.reshape(-1), channels, rate)
    g, w = a.process_block(blk), b.process_block(blk)
    if w is None:
        print(step, "frames", frames, "rate", rate, "ch", channels, "silent", silent, "-> None"); continue
    for tr in range(2):
        for k in range(2):
            y = w.traces[tr][k].astype(np.float64)
            if not len(y): continue
            x = g.traces[tr][k].astype(np.float64)
            d = np.abs(x - y)
            clear = (y > -88.0) & (x > -88.0)
            loud = clear & (y > y.max() - 60.0)
            worst = int(np.argmax(np.where(loud, d, 0)))
            if d[worst] > 0.005:
                idx = np.argsort(-np.where(loud, d, 0))[:12]
                print("   raw at worst bin: hip", g.traces[tr][1][worst], "oracle", w.traces[tr][1][worst], " weighted neighbours hip", x[worst-2:worst+3], "oracle", y[worst-2:worst+3], "raw neighbours", w.traces[tr][1][worst-2:worst+3])
                print("   bins", [(int(i), round(x[i], 2), round(y[i], 2)) for i in idx], "n loud differing > 0.01:", int((np.where(loud, d, 0) > 0.01).sum()), "of", int(loud.sum()))
            print(step, "frames", frames, "rate", rate, "ch", channels, "silent", silent, "trace", tr, k, "max dB", round(y.max(), 2),
                  "worst loud bin", worst, "hip", round(x[worst], 3), "oracle", round(y[worst], 3), "diff", round(d[worst], 4))
