"""Throughput of the non-headline banks at the BASELINE.json shapes (cfg3 loudness, cfg4 scope + stereometer).
Not the driver's bench line (bench.py is); used to fill DESIGN.md and to tune K4/K5/K6/K7."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import openmeters_amd  # noqa: E402
from openmeters_amd import banks, capi  # noqa: E402

api = openmeters_amd.api()
dev = torch.device("cuda", 0)
FS = 48000.0
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def loudness(S=1024, C=8, blocks=64, reps=5, out=sys.stdout):
    frames = 256 * blocks
    n = torch.arange(frames, device=dev, dtype=torch.float64)
    pcm = torch.empty((S, frames, C), device=dev, dtype=torch.float32)
    for c in range(C):
        f = 60.0 if c == 3 else 997.0 + 10.0 * c
        pcm[:, :, c] = (0.5 * torch.sin(2 * np.pi * f * n / FS)).to(torch.float32)[None, :]
    bank = banks.LoudnessBank(api, capi.LoudnessConfig(), S, C)
    bank.set_option(capi.OPT_KERNEL_TIMING, 1)
    run = lambda: bank.process_device(pcm.data_ptr(), 256, blocks, C, FS, capi.SURROUND, stream)
    for _ in range(12):  # > 4 s of audio so all four windows are full
        run()
    bank.kernel_time()
    dt = timed(run, reps)
    kms, _ = bank.kernel_time()
    cs = S * C * frames
    # SURVEY §8(d) prices the REFERENCE formulation (sliding Kahan sums): 4 B PCM + 8 B ring write + 4 x 8 B expiring reads = 44 B per
    # channel-sample.  The chunk-parallel form (loudness_chunked.hip, what a bank call of this size runs) needs no expiring reads:
    # its compulsory traffic is 4 + 8 = 12 B, and it actually moves ~20.5 B (the PCM is read by three passes, + sub-block sums)
    print(f"cfg3 loudness: {S}x{C}ch, {blocks} blocks/call: {dt*1e3:.2f} ms/call (kernel {kms:.2f} ms) -> {cs/dt/1e9:.2f} G channel-samples/s, "
          f"{cs/dt/(S*C*FS):.0f}x real time; HBM: {cs*12/(kms*1e-3)/8e12*100:.1f}% of 8 TB/s at this form's 12 B/channel-sample "
          f"({cs*20.5/(kms*1e-3)/8e12*100:.1f}% counting its three PCM passes), {cs*44/(kms*1e-3)/8e12*100:.1f}% at the reference formulation's 44 B", file=out)
    snap = bank.fetch(0, blocks - 1)
    print("   stream0 last snapshot:", snap.short_term_loudness, snap.momentary_loudness, snap.true_peak_db[:3], file=out)
    return {"workload": f"{S} streams x {C} ch, {blocks} blocks of 256 per call", "channel_samples_per_s": cs / dt, "x_real_time": cs / dt / (S * C * FS),
            "ms_per_call": dt * 1e3, "kernel_ms": kms, "form": "chunk-parallel (loudness_chunked.hip)",
            "hbm_frac_compulsory_12B": cs * 12 / (kms * 1e-3) / 8e12, "hbm_frac_moved_20p5B": cs * 20.5 / (kms * 1e-3) / 8e12,
            "hbm_frac_reference_formulation_44B": cs * 44 / (kms * 1e-3) / 8e12,
            "momentary_lufs_stream0": float(snap.momentary_loudness)}


def scope_stereo(S=256, blocks=64, reps=5, out=sys.stdout):
    frames = 256 * blocks
    n = torch.arange(frames, device=dev, dtype=torch.float64)
    pcm = torch.empty((S, frames, 2), device=dev, dtype=torch.float32)
    for s in range(S):
        f = 440.0 * 2.0 ** ((s % 24) / 12.0)
        left = (0.8 * torch.sin(2 * np.pi * f * n / FS)).to(torch.float32)
        pcm[s, :, 0] = left
        pcm[s, :, 1] = -0.7 * left
    pos = capi.positions_fallback(2)
    st = banks.StereometerBank(api, capi.StereometerConfig(analyze_bands=True, correlation_window=0.05, segment_duration=0.02,
                                                           target_sample_count=2000), S)
    dt = timed(lambda: st.process_device(pcm.data_ptr(), 256, blocks, 2, FS, pos, stream), reps)
    print(f"cfg4 stereometer: {S} streams, {blocks} blocks/call: {dt*1e3:.2f} ms/call -> {S*blocks/dt/1e3:.0f} k blocks/s, "
          f"{S*frames/dt/(S*FS):.0f}x real time", file=out)
    res = {"stereometer": {"workload": f"{S} streams x 2 ch, {blocks} blocks of 256 per call", "blocks_per_s": S * blocks / dt,
                           "x_real_time": frames / dt / FS, "ms_per_call": dt * 1e3}}
    sc = banks.OscilloscopeBank(api, capi.OscilloscopeConfig(segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2,
                                                             trigger_source=capi.CH_LEFT, channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT), S)
    dt = timed(lambda: sc.process_device(pcm.data_ptr(), 256, blocks, 2, FS, pos, stream), reps)
    hdr, _ = sc.fetch(0, blocks - 1)
    print(f"cfg4 oscilloscope: {S} streams, {blocks} blocks/call: {dt*1e3:.2f} ms/call -> {S*blocks/dt/1e3:.1f} k blocks/s, "
          f"{S*frames/dt/(S*FS):.0f}x real time; stream0 locked={hdr.locked} period={hdr.period:.3f} spc={hdr.samples_per_channel}", file=out)
    res["oscilloscope"] = {"workload": f"{S} streams x 2 ch, {blocks} blocks of 256 per call", "blocks_per_s": S * blocks / dt,
                           "x_real_time": frames / dt / FS, "ms_per_call": dt * 1e3, "stream0_locked": int(hdr.locked),
                           "stream0_period": float(hdr.period)}
    return res


def waveform(blocks=64, reps=5, out=sys.stdout):
    """the waveform bank (band analysis on; with and without RMS history), 256-frame blocks x `blocks` per call."""
    frames = 256 * blocks
    res = {}
    for S in (64, 1024, 4096):
        g = torch.Generator(device=dev).manual_seed(S)
        pcm = (torch.rand((S, frames, 2), device=dev, generator=g) - 0.5).contiguous()
        pos = capi.positions_fallback(2)
        for history in (False, True):
            bank = banks.WaveformBank(api, capi.WaveformConfig(analyze_bands=True, track_history=history), S)
            run = lambda: bank.process_device(pcm.data_ptr(), frames, 2, FS, pos, stream)
            run()
            dt = timed(run, reps)
            print(f"waveform: {S} streams, history={int(history)}, {blocks} blocks/call: {dt*1e3:.2f} ms/call -> {S*blocks/dt/1e3:.0f} k blocks/s, "
                  f"{frames/dt/FS:.0f}x real time per stream", file=out)
            res[f"{S}_streams_history_{int(history)}"] = {"blocks_per_s": S * blocks / dt, "ms_per_call": dt * 1e3}
            bank.close()
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "waveform":
        waveform()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "scope":  # cfg4's two banks only (tools/profile_scope_sq.sh)
        scope_stereo()
        sys.exit(0)
    loudness()
    scope_stereo()
    waveform()
