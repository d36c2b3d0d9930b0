"""replay tests/test_gpu_state_machine.py::test_spectrogram_random_operation_sequences for one seed and, at every column whose frequency
metric exceeds its bar, print the shape and the worst point three ways (HIP, oracle, exact f64).  usage: sg_seed.py <seed>"""
import os, sys
root = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np
import conftest, exact_f64
import openmeters_amd
from openmeters_amd import capi
from openmeters_amd.capi import Api, AudioBlock, SpectrogramConfig, SpectrogramProcessor
from parity import align_points
from test_gpu_state_machine import signal

seed = int(sys.argv[1])
omx = openmeters_amd.api()
oracle = Api(conftest._build_oracle(), "omxo_")
rng = np.random.default_rng(seed)
sizes = [256, 512, 1024, 2048, 4096]
cfg = SpectrogramConfig(fft_size=int(rng.choice(sizes)), hop_size=int(rng.choice([64, 100, 256, 777])),
                        use_reassignment=bool(rng.integers(2)), history_length=int(rng.choice([3, 64, 8192])))
a, b = SpectrogramProcessor(omx, cfg), SpectrogramProcessor(oracle, cfg)
b.debug_capture(True)
prng = np.random.default_rng(seed + 7919)
rate, channels, t0 = 48000.0, 2, 0
for step in range(45):
    op = rng.random()
    if op < 0.08:
        cfg = SpectrogramConfig(sample_rate=rate, fft_size=int(rng.choice(sizes)), hop_size=int(rng.choice([64, 100, 256, 777])),
                                window=int(rng.integers(5)), use_reassignment=bool(rng.integers(2)),
                                zero_padding_factor=int(rng.choice([1, 1, 1, 2])), history_length=int(rng.choice([3, 64, 8192])))
        a.update_config(cfg); b.update_config(cfg)
        continue
    if op < 0.12:
        a.reset_audio(); b.reset_audio()
        continue
    if op < 0.16:
        rate = float(rng.choice([44100.0, 48000.0, 96000.0]))
    if op < 0.20:
        channels = int(rng.choice([1, 2, 6]))
    frames = int(rng.choice([0, 1, 37, 256, 256, 1024, 3000, 9000]))
    silent = rng.random() < 0.15
    pcm = signal(rng, frames, channels, t0, rate, silent=silent)
    t0 += frames
    blk = AudioBlock(pcm.reshape(-1), channels, rate)
    g, w = a.process_block(blk), b.process_block(blk)
    # (the test's third processor consumes prng; irrelevant here)
    if w is None or not w.new_columns or w.new_columns[0].ndim != 2:
        continue
    eff = b.config()
    maxima = [float(o[:, 2].max()) if len(o) else 0.0 for o in w.new_columns]
    for i, (h, o) in enumerate(zip(g.new_columns, w.new_columns)):
        if len(h) == 0 or len(o) == 0 or maxima[i] < 1e-10:
            continue
        mp = max(o[:, 2].max(), h[:, 2].max())
        pairs, _, _ = align_points(h, o, float(mp))
        pa = np.array([p[0] for p in pairs], int); pb = np.array([p[1] for p in pairs], int)
        hh, oo = h[pa].astype(np.float64), o[pb].astype(np.float64)
        r = np.sqrt(oo[:, 2] / mp)
        df = np.abs(hh[:, 1] - oo[:, 1]) * r / (rate * 0.5)
        k = int(df.argmax())
        if df[k] > 3e-5:
            block = b.debug_captured(i)
            ex, bins = exact_f64.reassigned_column(block, window_kind=eff.window, window_size=eff.fft_size, zero_padding=eff.zero_padding_factor,
                                                   hop=w.hop_size, sample_rate=rate)
            near = ex[np.abs(ex[:, 1] - oo[k, 1]).argmin()] if len(ex) else None
            print(f"step {step} col {i}/{len(w.new_columns)} W={eff.fft_size} zp={eff.zero_padding_factor} hop={w.hop_size} window={eff.window} rate={rate} ch={channels} frames={frames} silent={silent}")
            print(f"   column max {mp:.3e}, loudest column of the update {max(maxima):.3e}; worst point: r={r[k]:.3e} P={oo[k,2]:.3e}")
            print(f"   f  HIP {hh[k,1]:.4f}  oracle {oo[k,1]:.4f}  exact {near[1] if near is not None else float('nan'):.4f}   t  HIP {hh[k,0]:.5f} oracle {oo[k,0]:.5f} exact {near[0] if near is not None else float('nan'):.5f}")
            print(f"   P  HIP {hh[k,2]:.5e} oracle {oo[k,2]:.5e} exact {near[2] if near is not None else float('nan'):.5e}")
