"""Generates tests/golden/*.npz from the CPU oracle (after it passes the ported KATs).

The reference cannot be executed in this environment (no Rust toolchain, un-vendored crates), and holds no
data fixtures of its own, so these vectors are produced by OUR restatement; they pin the oracle against
regressions and give the HIP path fixed targets.  Inputs are regenerated from formulas (tests/signals.py);
only expected outputs + the generating parameters are stored.   Run:  python tools/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from openmeters_amd import capi  # noqa: E402
from openmeters_amd.capi import (AudioBlock, LoudnessConfig, LoudnessProcessor, OscilloscopeConfig, OscilloscopeProcessor,  # noqa: E402
                                 SpectrogramConfig, SpectrogramProcessor, SpectrumConfig, SpectrumProcessor,
                                 StereometerConfig, StereometerProcessor)
from golden_inputs import cfg1_pcm, cfg2_pcm, cfg3_pcm, cfg4_pcm  # noqa: E402

oracle = capi.Api(os.path.join(ROOT, "oracle", "libomx_oracle.so"), "omxo_")
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
FS = 48000.0
meta = {}

# (i) cfg1: 2-ch sweep, classic 1024/256 Hann -> u16 columns (first 1 s)
pcm = cfg1_pcm(48000)
up = SpectrogramProcessor(oracle, SpectrogramConfig(fft_size=1024, hop_size=256, use_reassignment=False, history_length=8192)
                          ).process_block(AudioBlock(pcm.reshape(-1), 2, FS))
cols = np.stack(up.new_columns)
np.savez_compressed(os.path.join(OUT, "cfg1_classic.npz"), codes=cols[::8])
meta["cfg1_classic"] = dict(columns=len(up.new_columns), stored_every=8, frames=48000)

# (ii) cfg2: stream 5, 0.5 s, reassigned 4096/256 -> first / last 4 columns; spectrum last hop
pcm = cfg2_pcm(5, 24000)
up = SpectrogramProcessor(oracle, SpectrogramConfig(fft_size=4096, hop_size=256, use_reassignment=True, history_length=8192)
                          ).process_block(AudioBlock(pcm.reshape(-1), 2, FS))
keep = [0, 1, 2, 3, len(up.new_columns) - 4, len(up.new_columns) - 3, len(up.new_columns) - 2, len(up.new_columns) - 1]
arrs = {f"col{i}": up.new_columns[i] for i in keep}
snap = SpectrumProcessor(oracle, SpectrumConfig(fft_size=4096, hop_size=256, floor_db=-100.0)).process_block(AudioBlock(pcm.reshape(-1), 2, FS))
np.savez_compressed(os.path.join(OUT, "cfg2_reassigned.npz"), counts=np.array([len(c) for c in up.new_columns]),
                    weighted=snap.traces[0][0], raw=snap.traces[0][1], **arrs)
meta["cfg2_reassigned"] = dict(columns=len(up.new_columns), kept=keep, power_scale=up.reassigned_power_scale, stream=5, frames=24000)

# (iii) cfg3: 8-ch SURROUND, 4 s, blocks of 256 -> last 8 snapshots
pcm = cfg3_pcm(0, 256 * 750)
p = LoudnessProcessor(oracle, LoudnessConfig())
snaps = [p.process_block(AudioBlock(pcm[k:k + 256].reshape(-1), 8, FS, capi.SURROUND)) for k in range(0, pcm.shape[0], 256)]
last = snaps[-8:]
np.savez_compressed(os.path.join(OUT, "cfg3_loudness.npz"),
                    short_term=np.array([s.short_term_loudness for s in last], np.float32),
                    momentary=np.array([s.momentary_loudness for s in last], np.float32),
                    rms_fast=np.stack([s.rms_fast_db for s in last]), rms_slow=np.stack([s.rms_slow_db for s in last]),
                    true_peak=np.stack([s.true_peak_db for s in last]))
meta["cfg3_loudness"] = dict(blocks=750, stored_last=8)

# (iv) cfg4: stream 1 (sine), 20 x 1024 -> scope snapshot + cycle rate; stereometer correlations per block
pcm = cfg4_pcm(1, 1024 * 20)
sp = OscilloscopeProcessor(oracle, OscilloscopeConfig(segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2,
                                                      trigger_source=capi.CH_LEFT, channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT))
st = StereometerProcessor(oracle, StereometerConfig(analyze_bands=True, correlation_window=0.05, segment_duration=0.02,
                                                    target_sample_count=2000))
corr, snap = [], None
for k in range(0, pcm.shape[0], 1024):
    blk = AudioBlock(pcm[k:k + 1024].reshape(-1), 2, FS)
    snap = sp.process_block(blk) or snap
    w = st.process_block(blk)
    corr.append(w.correlations if w is not None else np.full(4, np.nan, np.float32))
np.savez_compressed(os.path.join(OUT, "cfg4_scope_stereo.npz"), samples=snap.samples, spc=np.array([snap.samples_per_channel]),
                    cycle_rate=np.array([sp.last_cycle_rate()], np.float32), correlations=np.stack(corr))
meta["cfg4_scope_stereo"] = dict(blocks=20, block_frames=1024, stream=1)

with open(os.path.join(OUT, "golden.json"), "w") as f:
    json.dump(dict(generator="tools/make_golden.py", source="CPU oracle (oracle/), inputs from tests/golden_inputs.py formulas",
                   sets=meta), f, indent=1)
print({k: os.path.getsize(os.path.join(OUT, k)) for k in os.listdir(OUT)})
