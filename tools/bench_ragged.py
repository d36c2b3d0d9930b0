"""Cost of the ragged entry points against the lock-step ones at equal work (run on the GPU box): every stream gets the same count, so
both calls do the same arithmetic; the difference is the per-stream plan kernel, the per-stream positions in the kernels, and — for the
meter banks — the sequential kernel the ragged path runs instead of the role / chunk-parallel forms."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
dev = torch.device("cuda", 0)
pos = capi.positions_fallback(2)
FS = 48000.0


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def line(name, lock_ms, ragged_ms, unit_count, unit):
    print(f"{name:46s} lock-step {lock_ms:7.3f} ms   ragged {ragged_ms:7.3f} ms   ({unit_count / ragged_ms / 1e3:8.2f} M {unit}/s ragged, "
          f"{unit_count / lock_ms / 1e3:8.2f} M lock-step)")


g = torch.Generator(device=dev).manual_seed(1)
# spectrogram, cfg2 shape: 64 streams x 1024 columns per call
S, F = 64, 256 * 1024
pcm = ((torch.rand((S, F, 2), device=dev, generator=g) - 0.5) * 0.5).contiguous()
cfg = capi.SpectrogramConfig(fft_size=4096, hop_size=256, use_reassignment=True, history_length=8192)
a, b = banks.SpectrogramBank(api, cfg, S), banks.SpectrogramBank(api, cfg, S)
lock = timed(lambda: a.process_device(pcm.data_ptr(), F, 2, FS, pos))
rag = timed(lambda: b.process_ragged(pcm.data_ptr(), F, [F] * S, 2, FS, pos))
line("spectrogram 4096/256 reassigned, 64 x 1024 cols", lock, rag, S * 1024, "frames")
# spectrum, same shape
sc = capi.SpectrumConfig(fft_size=4096, hop_size=256)
a, b = banks.SpectrumBank(api, sc, S, emit_all_hops=True), banks.SpectrumBank(api, sc, S, emit_all_hops=True)
lock = timed(lambda: a.process_device(pcm.data_ptr(), F, 2, FS, pos))
rag = timed(lambda: b.process_ragged(pcm.data_ptr(), F, [F] * S, 2, FS, pos))
line("spectrum 4096/256, 64 x 1024 hops", lock, rag, S * 1024, "hops")
del pcm
# meters: 1024 streams x 2 ch x 64 blocks (the cfg5 shard's shape)
S, blocks = 1024, 64
pcm = ((torch.rand((S, 256 * blocks, 2), device=dev, generator=g) - 0.5) * 0.5).contiguous()
nb = [blocks] * S
a, b = banks.LoudnessBank(api, capi.LoudnessConfig(), S, 2), banks.LoudnessBank(api, capi.LoudnessConfig(), S, 2)
lock = timed(lambda: a.process_device(pcm.data_ptr(), 256, blocks, 2, FS, pos), 10)
rag = timed(lambda: b.process_ragged(pcm.data_ptr(), 256, blocks, nb, 2, FS, pos), 10)
line("loudness, 1024 x 2 ch x 64 blocks", lock, rag, S * blocks, "blocks")
stc = capi.StereometerConfig(analyze_bands=True, correlation_window=0.05, segment_duration=0.02, target_sample_count=2000)
a, b = banks.StereometerBank(api, stc, S), banks.StereometerBank(api, stc, S)
lock = timed(lambda: a.process_device(pcm.data_ptr(), 256, blocks, 2, FS, pos), 10)
rag = timed(lambda: b.process_ragged(pcm.data_ptr(), 256, blocks, nb, 2, FS, pos), 10)
line("stereometer (bands), 1024 x 64 blocks", lock, rag, S * blocks, "blocks")
wc = capi.WaveformConfig(analyze_bands=True, track_history=False, max_columns=8192)
a, b = banks.WaveformBank(api, wc, S), banks.WaveformBank(api, wc, S)
fr = 256 * blocks
lock = timed(lambda: a.process_device(pcm.data_ptr(), fr, 2, FS, pos), 5)
rag = timed(lambda: b.process_ragged(pcm.data_ptr(), fr, [fr] * S, 2, FS, pos), 5)
line("waveform (bands), 1024 x 64 blocks", lock, rag, S * blocks, "blocks")
S2 = 256
osc = capi.OscilloscopeConfig(segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2, trigger_source=capi.CH_LEFT,
                              channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT)
t = torch.arange(256 * blocks, device=dev, dtype=torch.float64)
left = (0.8 * torch.sin(2 * np.pi * 440.0 * t / FS)).to(torch.float32)
p2 = torch.stack([left, -0.7 * left], 1)[None].repeat(S2, 1, 1).contiguous()
a, b = banks.OscilloscopeBank(api, osc, S2), banks.OscilloscopeBank(api, osc, S2)
lock = timed(lambda: a.process_device(p2.data_ptr(), 256, blocks, 2, FS, pos), 5)
rag = timed(lambda: b.process_ragged(p2.data_ptr(), 256, blocks, [blocks] * S2, 2, FS, pos), 5)
line("oscilloscope, 256 x 64 blocks", lock, rag, S2 * blocks, "blocks")
# the capture group itself at the reference's cadence: 1024 captures x 2 ch, all six visuals at the reference's default configs, one
# 256-frame chunk per capture and call (bench_stream.py's workload), summary rows on — lock-step ingest against ingest_ragged with
# equal per-capture counts (one shared projection launch, per-capture rows)
from openmeters_amd.pipeline import CaptureGroup
import bench_stream
S, F = 1024, 256
cfgs = bench_stream.default_configs()
n = torch.arange(F * 8, device=dev, dtype=torch.float32)
base = (0.4 * torch.sin(2 * torch.pi * 440.0 * n / 48000.0))[None, :, None] * torch.tensor([1.0, -0.7], device=dev)[None, None, :]
pcm = (base + 0.01 * (torch.rand((S, F * 8, 2), device=dev) - 0.5)).contiguous()
chunks = [pcm[:, k * F:(k + 1) * F].contiguous() for k in range(8)]
# one group at a time: two live groups own eight side streams between them, and which of the four hardware queues a stream lands on
# then decides the number (0.35 against 0.30 ms for the same ragged call)
group_counts = np.full(S, F, np.uint32)   # (prebuilt: converting a 1024-entry list costs the tool ~40 us per call, a sixth of the call)
state = {"k": 0}


def run_group(ragged_calls):
    g = CaptureGroup(api, S, stats=True, **cfgs)

    def call():
        if ragged_calls:
            g.ingest_ragged(chunks[state["k"] % 8].data_ptr(), F, group_counts, 2, FS, pos)
        else:
            g.ingest(chunks[state["k"] % 8].data_ptr(), F, 2, FS, pos)
        state["k"] += 1
    for _ in range(40):
        call()
    ms = timed(call, 200)
    torch.cuda.synchronize()
    g.close()
    return ms


lock = run_group(False)
rag = run_group(True)
line("capture group, six visuals + rows, 1024 x 256 frames", lock, rag, S, "chunks")
