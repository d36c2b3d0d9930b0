"""The reference's own call cadence as a measured configuration (run on the GPU box): VisualManager is fed ONE batcher block per
call — 256 frames at 48 kHz, or one catch-up chunk of up to 1024 (src/meter.rs:15-25, :40-69) — so a many-capture service calls
omx_capture_group_ingest once per 5.33 ms of audio with 256 frames per capture.  1024 captures x 2 ch, all six visuals at the
reference's default configs, PCM resident on the device: microseconds per call and multiples of real time.
  python tools/bench_stream.py [--streams 1024] [--frames 256] [--calls 300] [--visuals all|sg,sp,ld,st,sc,wf] [--json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import capi
from openmeters_amd.pipeline import CaptureGroup

NAMES = {"sg": "spectrogram", "sp": "spectrum", "ld": "loudness", "st": "stereometer", "sc": "oscilloscope", "wf": "waveform"}


def default_configs():
    return dict(spectrogram=capi.SpectrogramConfig(), spectrum=capi.SpectrumConfig(), loudness=capi.LoudnessConfig(),
                stereometer=capi.StereometerConfig(analyze_bands=True), oscilloscope=capi.OscilloscopeConfig(),
                waveform=capi.WaveformConfig(analyze_bands=True))


def run(api, streams, frames, calls, visuals, warmup=40):
    dev = torch.device("cuda", 0)
    cfgs = {k: v for k, v in default_configs().items() if k in visuals}
    group = CaptureGroup(api, streams, **cfgs)   # block_frames = 0: every call is ONE block, the reference's partition (registry.rs:407-417)
    pos = capi.positions_fallback(2)
    n = torch.arange(frames * 8, device=dev, dtype=torch.float32)
    base = (0.4 * torch.sin(2 * torch.pi * 440.0 * n / 48000.0))[None, :, None] * torch.tensor([1.0, -0.7], device=dev)[None, None, :]
    pcm = (base + 0.01 * (torch.rand((streams, frames * 8, 2), device=dev) - 0.5)).contiguous()
    chunks = [pcm[:, k * frames:(k + 1) * frames].contiguous() for k in range(8)]
    stream = torch.cuda.current_stream().cuda_stream
    for k in range(warmup):
        group.ingest(chunks[k % 8].data_ptr(), frames, 2, 48000.0, pos, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(calls):
        group.ingest(chunks[k % 8].data_ptr(), frames, 2, 48000.0, pos, stream)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / calls * 1e6
    return us


def streaming(streams=1024, calls=200, out=sys.stdout):
    """`secondary.streaming_256` of bench.py: the six-visual group at the reference's cadence — one 256-frame block per capture and call,
    and one 1024-frame catch-up chunk (meter.rs:15-25, :40-69)"""
    api = openmeters_amd.api()
    rec = {"workload": f"{streams} captures x 2 ch, all six visuals at the reference's default configs, one omx_capture_group_ingest per batcher block",
           "form": "one block per call: every meter bank on its SEQUENTIAL kernels (the reference's operation order), spectrum window folds carried "
                   "between calls (window_sums_carry_kernel), reassigned spectrogram 2048 / 64 fused",
           "parity_bar": "group chunks: rows of profiles/parity_r*.txt (tests/test_gpu_capture_chunks.py: every capture against oracle handles fed each chunk whole)"}
    for frames, key in ((256, "block_256"), (1024, "catch_up_1024")):
        us = run(api, streams, frames, calls if frames == 256 else max(calls // 4, 20), list(NAMES.values()))
        audio_us = frames / 48000.0 * 1e6
        rec[key] = {"us_per_call": round(us, 1), "x_real_time": round(audio_us / us, 2), "chunks_per_s": round(streams / us * 1e6)}
        print(f"streaming: {streams} captures x {frames} frames, six visuals: {us:.1f} us per call = {audio_us / us:.1f}x real time", file=out)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--calls", type=int, default=300)
    ap.add_argument("--visuals", default="all")
    ap.add_argument("--each", action="store_true", help="also time every visual on its own")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    api = openmeters_amd.api()
    sets = [list(NAMES.values()) if a.visuals == "all" else [NAMES[v] for v in a.visuals.split(",")]]
    if a.each:
        sets += [[v] for v in NAMES.values()]
    out = {}
    for vis in sets:
        us = run(api, a.streams, a.frames, a.calls, vis)
        audio_us = a.frames / 48000.0 * 1e6
        key = "+".join(v[:5] for v in vis) if len(vis) < 6 else "all six"
        out[key] = dict(us_per_call=round(us, 1), x_real_time=round(audio_us / us, 2))
        if not a.json:
            print(f"{a.streams} captures x {a.frames} frames, {key}: {us:.1f} us per call = {audio_us / us:.2f}x real time")
    if a.json:
        print(json.dumps(dict(workload=f"{a.streams} captures x 2 ch, {a.frames} frames per call, reference default configs", **out)))


if __name__ == "__main__":
    main()
