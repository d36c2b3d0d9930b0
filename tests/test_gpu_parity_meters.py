"""Parity of the HIP loudness / stereometer / oscilloscope paths against the CPU oracle (`-m gpu`).

Tolerances (f32/f64 recurrences are evaluated in the reference's operation order with mul+add fusion
disabled, so most of these are far tighter in practice; device libm (logf/log10/expf) differs from
glibc by <= 2 ulp):
  loudness   LUFS / dB fields     |d| <= 1e-4 dB  (1e-5 relative of a 10 dB quantity)
  stereometer correlations        |d| <= 1e-6 ; decimated points bit-exact
  oscilloscope period             relative 1e-5 ; capture position within 1 sample (near-tie argmax flips are
                                  possible because tree reductions replace the reference's sequential sums);
                                  resampled trace within 1e-4 of the oracle where start/frac agree
"""
import os
import sys

import numpy as np
import pytest

from openmeters_amd import banks, capi
from openmeters_amd.capi import (AudioBlock, LoudnessConfig, LoudnessProcessor, OscilloscopeConfig, OscilloscopeProcessor,
                                 StereometerConfig, StereometerProcessor)
from parity import bar, check_chunked_rho, stereometer_band_rms
from signals import xorshift32_noise

pytestmark = pytest.mark.gpu
FS = 48000.0
# Stable-mode capture position.  `start` is an integer and must be equal.  `frac_offset` comes out of parabolic_refine
# (oscilloscope/processor.rs:14-19) over three neighbouring f32 correlation scores: frac = (p - n) / (2 (p - 2 b + n)), so a
# score perturbation e moves it by ~ e / |p - 2 b + n|.  The scores of HIP and oracle differ at the f32 level (e ~ 1e-7: the
# period estimate goes through an FFT whose rounding is unpinnable, SURVEY §8c, and everything downstream — Gaussian widths,
# template, means — inherits its last bits), and the correlation peak of a 2-cycle template is flat (curvature 1e-2 ... 1e-3),
# hence |d frac| of 1e-5 ... 1e-4 samples; measured maximum 3.3e-5 (profiles/parity_r02.txt), bar 3e-4.
# The resampled trace is a linear interpolation of the history at pos_i = frac + i * span / (n - 1) (:788-803, all f32), so the
# whole trace difference must be EXPLAINED by the position error
#     |d pos| <= |d frac| + (n - 1) |d period| / period + 2 ulp(n)      (span = period * cycles; pos_i is rounded to f32 on
#                                                                        both sides: ulp(219) = 1.5e-5 samples)
# times the largest sample-to-sample step of the input:  |d trace| <= |d pos| * max_step + 2e-6.
SCOPE_FRAC_BAR = 1.2e-3   # measured 1.1e-4 (profiles/parity_r02.txt)
# ... for periods around 100 samples.  The curvature of the correlation peak is (2 pi / period)^2, so the same score noise moves frac_offset
# by e period^2 / (4 pi^2): a 2430-sample period (79 Hz at 192 kHz) measures 0.015 samples.  With e = 1.6e-6 (16 x the f32 level of a
# score summed over 4000 ... 8000 samples): 4e-8 period^2 — 1.2e-3 at 173 samples, 0.24 at 2430.
SCOPE_FRAC_PER_PERIOD2 = 4e-8


def check_stable_trace(name, g_samples, w_samples, g_cap, w_cap, max_step, spc, rel_rate, detail=None, period=None, span=None):
    assert g_cap is not None and w_cap is not None and g_cap[0] == w_cap[0], (g_cap, w_cap, detail)     # integer start: bit-exact
    dfrac = abs(g_cap[1] - w_cap[1])
    if period is None or SCOPE_FRAC_PER_PERIOD2 * period * period <= SCOPE_FRAC_BAR:
        bar(f"{name}: |d frac_offset| samples", dfrac, SCOPE_FRAC_BAR, detail)
    else:
        bar(f"{name}: |d frac_offset| / (4e-8 period^2), periods beyond 173 samples", dfrac / (SCOPE_FRAC_PER_PERIOD2 * period * period), 1.0, (detail, period))
    # (positions run up to the span = cycles x period, which exceeds the sample count once the snapshot is capped at 4096 samples per channel)
    reach = max(float(spc - 1), float(span)) if span is not None else float(spc - 1)
    ulp = 2.0 ** (np.ceil(np.log2(max(reach + 1.0, 2.0))) - 24)
    dpos = dfrac + reach * rel_rate + 2.0 * ulp
    err = float(np.abs(g_samples - w_samples).max())
    bar(f"{name}: |d trace| - |d pos| * max input step", max(err - dpos * max_step * 1.001, 0.0), 2e-6, (err, dfrac, dpos, max_step, detail))
    if period is None or period <= 173.0:
        bar(f"{name}: |d trace| (bounded by the line above)", err, 5e-4, detail)   # measured 4.9e-5
    return err


def cfg3_pcm(s, frames, channels=8):
    """channel c = 0.5*sin(2*pi*(997+10c+0.01s) n/fs), LFE (index 3) at 60 Hz."""
    n = np.arange(frames, dtype=np.float64)
    out = np.empty((frames, channels), np.float32)
    for c in range(channels):
        f = 60.0 if c == 3 else 997.0 + 10.0 * c + 0.01 * s
        out[:, c] = (0.5 * np.sin(2 * np.pi * f * n / FS)).astype(np.float32)
    return out


def snapshots_close(a, b, tol=1e-4):
    bar("loudness: |d short-term LUFS|", abs(a.short_term_loudness - b.short_term_loudness), tol)
    bar("loudness: |d momentary LUFS|", abs(a.momentary_loudness - b.momentary_loudness), tol)
    for f in ("rms_fast_db", "rms_slow_db", "true_peak_db"):
        bar(f"loudness: |d {f}|", np.abs(getattr(a, f) - getattr(b, f)).max(), tol)
    assert a.channel_count == b.channel_count and a.positions == b.positions


@pytest.mark.parametrize("channels,rate", [(8, 48000.0), (2, 44100.0), (6, 96000.0), (1, 192000.0)])
def test_loudness_blocks_match_oracle(omx, oracle, channels, rate):
    frames = 256 * 40
    pcm = cfg3_pcm(1, frames, channels)
    positions = capi.SURROUND if channels == 8 else capi.positions_fallback(channels)
    a = LoudnessProcessor(omx, LoudnessConfig(sample_rate=rate))
    b = LoudnessProcessor(oracle, LoudnessConfig(sample_rate=rate))
    for k in range(0, frames, 256):
        blk = pcm[k:k + 256].reshape(-1)
        snapshots_close(a.process_block(AudioBlock(blk, channels, rate, positions)),
                        b.process_block(AudioBlock(blk, channels, rate, positions)))


def test_loudness_bank_equals_per_block_oracle_with_full_windows(omx, oracle):
    """cfg3 shape, scaled: 16 streams x 8 ch, 4.1 s (all four windows full, ring wrapped), blocks of 256."""
    S, C, blocks = 16, 8, 770
    frames = 256 * blocks
    pcm = np.stack([cfg3_pcm(s, frames, C) for s in range(S)])
    bank = banks.LoudnessBank(omx, LoudnessConfig(), S, C)
    assert bank.process_host(pcm, 256, C, FS, capi.SURROUND) is not None
    for s in (0, 7, 15):
        p = LoudnessProcessor(oracle, LoudnessConfig())
        want = [p.process_block(AudioBlock(pcm[s, k:k + 256].reshape(-1), C, FS, capi.SURROUND)) for k in range(0, frames, 256)]
        for blk in (0, 1, 55, 56, 57, 187, 188, 562, 563, 769):  # around every window-fill / refresh boundary and the end
            snapshots_close(bank.fetch(s, blk), want[blk])
    # BS.1770 anchor through the bank: 997 Hz family at 0.5 amp, weights FL FR FC 1, LFE 0, surrounds 1.41
    last = bank.fetch(0, blocks - 1)
    ms = 0.125 * (3 * 1.0 + 4 * 1.41)
    assert abs(last.short_term_loudness - 10 * np.log10(ms)) < 0.15  # K gain over 997..1067 Hz is +0.69..0.78 dB, offset -0.691


def test_loudness_leading_silence_and_reset(omx, oracle):
    pcm = np.zeros((48001 + 4800, 2), np.float32)
    pcm[48001:, :] = (0.5 * np.sin(2 * np.pi * 1000.0 * np.arange(4800) / FS)).astype(np.float32)[:, None]
    a, b = LoudnessProcessor(omx, LoudnessConfig()), LoudnessProcessor(oracle, LoudnessConfig())
    snapshots_close(a.process_block(AudioBlock(pcm.reshape(-1), 2, FS)), b.process_block(AudioBlock(pcm.reshape(-1), 2, FS)))
    a.reset_audio()
    b.reset_audio()
    snapshots_close(a.process_block(AudioBlock(pcm[-2048:].reshape(-1), 2, FS)), b.process_block(AudioBlock(pcm[-2048:].reshape(-1), 2, FS)))


def cfg4_pcm(s, frames):
    """L = 440*2^((s mod 24)/12) Hz saw/sine/square by s mod 3, R = -0.7 L + noise -40 dBFS."""
    f = 440.0 * 2.0 ** ((s % 24) / 12.0)
    c = (f * np.arange(frames, dtype=np.float64) / FS)
    kind = s % 3
    left = (2.0 * (c - np.floor(c)) - 1.0) if kind == 0 else (np.sin(2 * np.pi * c) if kind == 1 else np.where((c - np.floor(c)) < 0.5, 1.0, -1.0))
    left = (0.8 * left).astype(np.float32)
    right = (-0.7 * left + xorshift32_noise(0x9E3779B9 ^ s, frames, 1e-2)).astype(np.float32)
    return np.stack([left, right], 1)


def test_stereometer_blocks_match_oracle(omx, oracle):
    cfg = StereometerConfig(analyze_bands=True, emit_band_points=True, correlation_window=0.05, segment_duration=0.02,
                            target_sample_count=2000)
    pcm = cfg4_pcm(4, 256 * 24)
    a, b = StereometerProcessor(omx, cfg), StereometerProcessor(oracle, cfg)
    for k in range(0, pcm.shape[0], 256):
        g = a.process_block(AudioBlock(pcm[k:k + 256].reshape(-1), 2, FS))
        w = b.process_block(AudioBlock(pcm[k:k + 256].reshape(-1), 2, FS))
        assert (g is None) == (w is None)
        if g is None:
            continue
        bar("stereometer: |d rho|", np.abs(g.correlations - w.correlations).max(), 1e-6)
        for band in range(4):
            assert g.points[band].shape == w.points[band].shape
            assert np.array_equal(g.points[band].view(np.uint32), w.points[band].view(np.uint32)), band  # bit-exact biquads


def test_stereometer_bank_and_surround_fold(omx, oracle):
    S, blocks = 12, 10
    cfg = StereometerConfig(analyze_bands=True, correlation_window=0.05, segment_duration=0.02, target_sample_count=2000)
    pcm = np.stack([cfg4_pcm(s, 256 * blocks) for s in range(S)])
    bank = banks.StereometerBank(omx, cfg, S)
    bank.set_option(capi.OPT_KERNEL_FORM, 1)   # the sequential kernels' bar below (by shape a 10-block call takes the chunk-parallel form)
    bank.process_host(pcm, 256, 2, FS)
    for s in (0, 5, 11):
        p = StereometerProcessor(oracle, cfg)
        for blk in range(blocks):
            w = p.process_block(AudioBlock(pcm[s, blk * 256:(blk + 1) * 256].reshape(-1), 2, FS))
            corr, produced = bank.fetch(s, blk)
            assert produced == (w is not None)
            if w is not None:
                bar("stereometer: |d rho|", np.abs(corr - w.correlations).max(), 1e-6)
    # 8-channel SURROUND fold feeds the same kernel (dsp.rs:135-176 weights)
    x = cfg3_pcm(2, 256 * 6, 8)
    a, b = StereometerProcessor(omx, cfg), StereometerProcessor(oracle, cfg)
    for k in range(0, x.shape[0], 256):
        g = a.process_block(AudioBlock(x[k:k + 256].reshape(-1), 8, FS, capi.SURROUND))
        w = b.process_block(AudioBlock(x[k:k + 256].reshape(-1), 8, FS, capi.SURROUND))
        assert (g is None) == (w is None)
        if g is not None:
            bar("stereometer: |d rho|", np.abs(g.correlations - w.correlations).max(), 1e-6)
            assert np.array_equal(g.points[0].view(np.uint32), w.points[0].view(np.uint32))


def scope_cfg():
    return OscilloscopeConfig(segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2, trigger_source=capi.CH_LEFT,
                              channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT)


@pytest.mark.parametrize("s", [0, 1, 2, 13])
def test_oscilloscope_blocks_match_oracle(omx, oracle, s):
    """cfg4 shape: linked trigger on Left, 256-frame blocks; lock state, period, snapshot geometry and the
    resampled traces must follow the oracle block by block."""
    pcm = cfg4_pcm(s, 256 * 120)
    a, b = OscilloscopeProcessor(omx, scope_cfg()), OscilloscopeProcessor(oracle, scope_cfg())
    period = FS / (440.0 * 2.0 ** ((s % 24) / 12.0))
    max_step = float(np.abs(np.diff(pcm, axis=0)).max())
    compared = 0
    for k in range(0, pcm.shape[0], 256):
        blk = pcm[k:k + 256].reshape(-1)
        g, w = a.process_block(AudioBlock(blk, 2, FS)), b.process_block(AudioBlock(blk, 2, FS))
        assert (g is None) == (w is None)
        ra, rb = a.last_cycle_rate(), b.last_cycle_rate()
        assert (ra is None) == (rb is None)
        if ra is not None:
            bar("oscilloscope: rel |d cycle rate|", abs(ra - rb) / rb, 1e-4)
        if g is None:
            continue
        assert (g.epoch, g.channels, g.slots[:g.channels], g.samples_per_channel) == (w.epoch, w.channels, w.slots[:w.channels],
                                                                                         w.samples_per_channel)
        if ra is not None and k > 256 * 60:
            # same capture (up to f32 noise) -> same resampled trace; a near-tie argmax flip would show up as a whole-sample
            # or whole-period shift, which the reference's own jitter test tolerates (< 3 samples, :933-955)
            check_stable_trace("oscilloscope (Stable)", g.samples, w.samples, a.last_capture(), b.last_capture(), max_step,
                               g.samples_per_channel, abs(ra - rb) / rb, k)
            compared += 1
    assert compared > 30
    assert abs(FS / a.last_cycle_rate() - period) < 0.02 * period


def test_oscilloscope_bank_matches_single_stream_handles(omx, oracle):
    S, blocks = 9, 60
    pcm = np.stack([cfg4_pcm(s, 256 * blocks) for s in range(S)])
    bank = banks.OscilloscopeBank(omx, scope_cfg(), S)
    up = bank.process_host(pcm, 256, 2, FS)
    assert up.n_streams == S and up.n_blocks == blocks and up.sample_stride == 4096
    for s in (0, 4, 8):
        p = OscilloscopeProcessor(oracle, scope_cfg())
        want = None
        for blk in range(blocks):
            w = p.process_block(AudioBlock(pcm[s, blk * 256:(blk + 1) * 256].reshape(-1), 2, FS))
            hdr, _ = bank.fetch(s, blk)
            assert bool(hdr.produced) == (w is not None)
            assert bool(hdr.locked) == (p.last_cycle_rate() is not None)
            if w is not None:
                assert (hdr.channels, hdr.samples_per_channel) == (w.channels, w.samples_per_channel)
                want = w
        hdr, samples = bank.fetch(s, blocks - 1, with_samples=True)
        n = hdr.samples_per_channel
        got = np.concatenate([samples[c, :n] for c in range(hdr.channels)])
        assert hdr.capture_start == p.last_capture()[0]
        check_stable_trace("oscilloscope (Stable)", got, want.samples, (hdr.capture_start, hdr.capture_frac), p.last_capture(),
                           float(np.abs(np.diff(pcm[s], axis=0)).max()), n, abs(hdr.period - FS / p.last_cycle_rate()) / hdr.period, s)


@pytest.mark.parametrize("rate,block", [(192000.0, 1024), (96000.0, 512)])
def test_oscilloscope_high_rates_wide_pass_hands_blocks_over(omx, oracle, rate, block):
    """88.2 ... 192 kHz: the wide trigger pass runs on 152 KiB of LDS — less than its worst case at these rates — and hands a stream's
    blocks over to the one-workgroup-per-stream kernel from the first block whose arrays do not fit (kernel = max(40 ms, two periods),
    `oscilloscope/processor.rs:184-189`: 5 len + 36 floats must fit 38 912).  Streams: 440 Hz (always fits), 22 Hz (never fits once
    the estimate is that long), a glide from 150 Hz down to 22 Hz (fits, then does not: hand-over in the middle of a call, and from
    block 0 of the next), near-silence.  Every block header and the newest trace against a per-stream oracle, three calls."""
    S, blocks, calls = 4, 20, 3
    n = block * blocks * calls
    t = np.arange(n) / rate
    glide = 150.0 * (22.0 / 150.0) ** (t / t[-1])
    sig = [0.6 * np.sin(2 * np.pi * 440.0 * t), 0.6 * np.sin(2 * np.pi * 22.0 * t), 0.6 * np.sin(2 * np.pi * np.cumsum(glide) / rate),
           1e-5 * np.sin(2 * np.pi * 300.0 * t)]
    rng = np.random.default_rng(11)
    pcm = np.stack([np.stack([x + 0.002 * rng.standard_normal(n), -0.7 * x], 1) for x in sig]).astype(np.float32)
    cfg = OscilloscopeConfig(segment_duration=0.05, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2, trigger_source=capi.CH_LEFT,
                             channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT)
    bank = banks.OscilloscopeBank(omx, cfg, S)
    refs = [OscilloscopeProcessor(oracle, cfg) for _ in range(S)]
    compared, handed = 0, []
    for call in range(calls):
        at = call * block * blocks
        up = bank.process_host(pcm[:, at:at + block * blocks], block, 2, rate)
        assert up.n_blocks == blocks
        handed.append([bank.resume_block(s) for s in range(S)])
        for s in range(S):
            want = None
            for blk in range(blocks):
                w = refs[s].process_block(AudioBlock(pcm[s, at + blk * block:at + (blk + 1) * block].reshape(-1), 2, rate))
                hdr, _ = bank.fetch(s, blk)
                assert bool(hdr.produced) == (w is not None), (call, s, blk)
                assert bool(hdr.locked) == (refs[s].last_cycle_rate() is not None), (call, s, blk)
                if w is not None:
                    assert hdr.channels == w.channels and abs(int(hdr.samples_per_channel) - int(w.samples_per_channel)) <= 1, (call, s, blk)
                    want = w
                if hdr.locked:
                    bar("oscilloscope: rel |d cycle rate|", abs(rate / hdr.period - refs[s].last_cycle_rate()) / refs[s].last_cycle_rate(), 1e-4, (call, s, blk))
            hdr, samples = bank.fetch(s, blocks - 1, with_samples=True)
            if want is not None and hdr.samples_per_channel == want.samples_per_channel:
                k = hdr.samples_per_channel
                got = np.concatenate([samples[c, :k] for c in range(hdr.channels)])
                if hdr.capture_start == refs[s].last_capture()[0]:
                    check_stable_trace("oscilloscope (Stable)", got, want.samples, (hdr.capture_start, hdr.capture_frac), refs[s].last_capture(),
                                       float(np.abs(np.diff(pcm[s, at:at + block * blocks], axis=0)).max()), k,
                                       abs(hdr.period - rate / refs[s].last_cycle_rate()) / hdr.period if hdr.locked else 0.0, (call, s),
                                       period=float(hdr.period) if hdr.locked else None, span=float(hdr.period) * cfg.num_cycles if hdr.locked else None)
                    compared += 1
    assert compared >= 6
    assert all(h[0] == blocks and h[3] == blocks for h in handed), handed            # 440 Hz, near-silence: the wide pass ran every block
    assert handed[-1][1] == 0, handed                                                 # 22 Hz: handed over from block 0 once locked there
    assert any(0 < h[2] < blocks for h in handed) or any(0 < h[1] < blocks for h in handed), handed   # a hand-over in the middle of a call


def _find_best(api, work, tmpl, search, period):
    import ctypes as C
    work, tmpl = np.ascontiguousarray(work, np.float32), np.ascontiguousarray(tmpl, np.float32)
    n = len(tmpl)
    assert len(work) == n + search
    off, frac, best, scores = C.c_uint32(), C.c_float(), C.c_float(), np.zeros(search + 1, np.float32)
    f = api.fn("debug_scope_find_best", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.POINTER(C.c_uint32),
                                                  C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p])
    api.check(f(work.ctypes.data, tmpl.ctypes.data, n, search, period, C.byref(off), C.byref(frac), C.byref(best), scores.ctypes.data))
    return off.value, frac.value, best.value, scores


def _exact_scores(work, tmpl, search):
    w, t = work.astype(np.float64), tmpl.astype(np.float64)
    n = len(t)
    out = np.zeros(search + 1)
    sy, syy = t.sum(), (t * t).sum()
    ey = max(syy - sy * sy / n, 0.0)
    for o in range(search + 1):
        x = w[o:o + n]
        sx = x.sum()
        ex = max((x * x).sum() - sx * sx / n, 0.0)
        den = np.sqrt(ex * ey)
        out[o] = np.clip(((x * t).sum() - sx * sy / n) / den, -1.0, 1.0) if den > np.finfo(np.float32).eps else 0.0
    return out


@pytest.mark.parametrize("case", ["periodic tie (period 101)", "periodic tie (period 64)", "perturbed ties", "noise", "short period, long window"])
def test_scope_find_best_scores_and_near_tie_argmax(omx, oracle, case):
    """StableTrigger::find_best (oscilloscope/processor.rs:441-484) of the trigger pass on fixed arrays, against the oracle's
    statement-for-statement restatement and against exact (f64) arithmetic:
      * the correlation score of EVERY offset: |HIP - exact| and |oracle - exact| at the f32 noise of a 2000-term sum (bar 1e-6; the HIP
        sums — packed FMA, 64-lane tree — must not be further from exact arithmetic than twice the oracle's 4-chain sums);
      * the argmax: a signal that repeats inside the search span puts EXACT ties into the score table (offsets a period apart see the
        same samples).  The strict-> scan order decides them in the reference; here the two offsets' sums are taken in differently
        aligned groups, so the last bit may fall the other way.  Whichever offset the walk ends on, its score — in the oracle's own
        table — must equal the oracle's winner within the f32 noise (2e-6): a flip may only happen between offsets that are equally
        good.  Without ties (noise) offset and frac_offset agree with the oracle's."""
    seed = {"periodic tie (period 101)": 1, "periodic tie (period 64)": 2, "perturbed ties": 3, "noise": 4, "short period, long window": 5}[case]
    rng = np.random.default_rng(seed)
    n = 1920
    if case.startswith("periodic tie"):
        period = 101.0 if "101" in case else 64.0
        search = int(round(1.5 * period))
        cyc = (0.7 * np.sin(2 * np.pi * np.arange(int(period)) / period) + 0.2 * rng.standard_normal(int(period))).astype(np.float32)
        work = np.tile(cyc, (n + search) // int(period) + 2)[:n + search]
    elif case == "perturbed ties":
        period, search = 109.09, 164
        t = np.arange(n + search)
        work = (0.8 * np.sin(2 * np.pi * t / 109.0) + 1e-6 * rng.standard_normal(n + search)).astype(np.float32)
    elif case == "noise":
        period, search = 300.0, 450
        work = rng.standard_normal(n + search).astype(np.float32) * 0.3
    else:
        period, search = 12.4, 19
        work = (0.5 * np.sin(2 * np.pi * np.arange(n + search) / 12.4)).astype(np.float32)
    i = np.arange(n)
    edge = np.exp(-0.5 * ((np.minimum(i, n - 1 - i) - (n - 1) * 0.5) / max(0.25 * period, 1.0)) ** 2)
    tmpl = (np.where(i < n // 2, -edge, edge) + 0.6 * np.roll(work[:n], 7) * np.exp(-0.5 * ((i - 960) / (0.5 * period * 4)) ** 2)).astype(np.float32)
    work = (work - work.mean(dtype=np.float32)).astype(np.float32)
    o_off, o_frac, o_best, o_scores = _find_best(oracle, work, tmpl, search, period)
    h_off, h_frac, h_best, h_scores = _find_best(omx, work, tmpl, search, period)
    exact = _exact_scores(work, tmpl, search)
    err_o, err_h = np.abs(o_scores - exact).max(), np.abs(h_scores - exact).max()
    bar("scope find_best: |score - exact f64| (HIP)", err_h, 1e-6)
    bar("scope find_best: |score - exact f64| (oracle)", err_o, 1e-6)
    bar("scope find_best: HIP score error / max(oracle score error, 1.2e-7)", err_h / max(err_o, 1.2e-7), 2.0)
    bar("scope find_best: oracle-table score of the HIP winner below the oracle winner's", float(o_scores[o_off] - o_scores[h_off]), 2e-6)
    assert h_best == h_scores[h_off]
    if case in ("noise", "short period, long window"):
        assert h_off == o_off
        bar("scope find_best: |d frac_offset| (no ties)", abs(h_frac - o_frac), 1.2e-3)
    if h_off != o_off:   # a tie decided the other way: the two winners are a whole number of periods apart
        k = abs(h_off - o_off) / period
        assert abs(k - round(k)) < 0.02 and round(k) >= 1, (h_off, o_off)


def test_oscilloscope_zero_crossing_mode_matches_oracle(omx, oracle):
    cfg = OscilloscopeConfig(segment_duration=0.01, trigger_mode=capi.TRIGGER_ZERO_CROSSING, channel_1=capi.CH_LEFT,
                             channel_2=capi.CH_MID, trigger_source=capi.CH_NONE)
    pcm = cfg4_pcm(1, 256 * 30)
    a, b = OscilloscopeProcessor(omx, cfg), OscilloscopeProcessor(oracle, cfg)
    for k in range(0, pcm.shape[0], 256):
        g = a.process_block(AudioBlock(pcm[k:k + 256].reshape(-1), 2, FS))
        w = b.process_block(AudioBlock(pcm[k:k + 256].reshape(-1), 2, FS))
        assert (g is None) == (w is None)
        if g is not None:
            assert (g.channels, g.samples_per_channel) == (w.channels, w.samples_per_channel)
            bar("oscilloscope (zero crossing): |d trace|", np.abs(g.samples - w.samples).max(), 1e-6)


def test_waveform_blocks_match_oracle(omx, oracle):
    """min/max bit-exact, colour bands / RMS history within 1e-5 relative (f32 biquads are evaluated in
    the reference's order without fusion; the KBN sums are sequential in both)."""
    from openmeters_amd.capi import WaveformConfig, WaveformProcessor
    cfg = WaveformConfig(scroll_speed=300.0, max_columns=512, analyze_bands=True, track_history=True)
    pcm = cfg4_pcm(7, 256 * 200)
    pcm[3000:3003, 0] = np.nan  # non-finite samples break column continuity (:275-291)
    a, b = WaveformProcessor(omx, cfg), WaveformProcessor(oracle, cfg)
    total = 0
    for k in range(0, pcm.shape[0], 256):
        g = a.process_block(AudioBlock(pcm[k:k + 256].reshape(-1), 2, FS))
        w = b.process_block(AudioBlock(pcm[k:k + 256].reshape(-1), 2, FS))
        assert g.reset == w.reset and g.columns.shape == w.columns.shape and g.preview_progress == w.preview_progress
        total += len(g.columns)
        assert np.array_equal(g.columns[:, :, :2].view(np.uint32), w.columns[:, :, :2].view(np.uint32))  # min / max
        if len(g.columns):
            bar("waveform: |d band colour| / max(1, max)", np.abs(g.columns[:, :, 2:5] - w.columns[:, :, 2:5]).max() /
                max(1.0, np.abs(w.columns[:, :, 2:5]).max()), 1e-6)
            bar("waveform: |d RMS history dB|", np.abs(g.columns[:, :, 5:] - w.columns[:, :, 5:]).max(), 2e-4)
        assert (g.preview is None) == (w.preview is None)
        if g.preview is not None:
            assert np.array_equal(g.preview[:, :2].view(np.uint32), w.preview[:, :2].view(np.uint32))
            bar("waveform: |d band colour| / max(1, max)", np.abs(g.preview[:, 2:5] - w.preview[:, 2:5]).max(), 1e-6)
    assert total in (319, 320)  # 0.00625 is not exact in f64: the reference phase accumulator lands one column short


def test_summary_reductions_on_device_resident_bank_outputs(omx, oracle):
    """spectrum peaks straight off the spectrum bank's d_traces, loudness bars + peak holds straight off
    the loudness bank's d_snapshots, compared with the oracle's reductions of the same (fetched) data."""
    import ctypes as C
    import torch
    from openmeters_amd.capi import SpectrumConfig
    from golden_inputs import cfg2_pcm
    S, frames = 5, 4096 + 256 * 15
    pcm = np.stack([cfg2_pcm(s, frames + 30000)[30000:] for s in range(S)])
    bank = banks.SpectrumBank(omx, SpectrumConfig(fft_size=4096, hop_size=256), S, emit_all_hops=True)
    up = bank.process_host(pcm, 2, FS)
    bins, hops = int(up.bins), int(up.n_hops_out)
    assert hops == 16
    rows = S * hops
    d_out = torch.empty((rows, 4), device="cuda:0", dtype=torch.int32)
    f = omx.fn("spectrum_peaks", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float,
                                           C.c_void_p, C.c_void_p])
    fbins = np.arange(bins, dtype=np.float32) * np.float32(FS / 4096)
    for weighting in (0, 1):   # trace 0, A-weighted / raw: row r starts at d_traces + (r * 4 + weighting) * bins
        omx.check(f(up.d_frequency_bins, up.d_traces + weighting * bins * 4, 1, bins, rows, 4 * bins, 20.0, float(fbins[-1]), None,
                    d_out.data_ptr()))
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(capi.SPECTRUM_PEAK_DTYPE).reshape(S, hops)
        for s in range(S):
            traces = np.stack([bank.fetch(s, h, bins)[0, weighting] for h in range(hops)])
            want = capi.spectrum_peaks(oracle, fbins, traces, 20.0, float(fbins[-1]))
            assert np.array_equal(got[s]["found"], want["found"]) and np.array_equal(got[s]["bin"], want["bin"])
            assert np.array_equal(got[s]["freq_hz"], want["freq_hz"]) and np.array_equal(got[s]["level_db"], want["level_db"])
            assert got[s]["found"].all() and (got[s]["freq_hz"] > 20.0).all()

    blocks = 40
    x = np.stack([cfg3_pcm(s, 256 * blocks, 6) * np.float32(1.0 if s % 2 else 0.05) for s in range(S)])
    x[:, 256 * 8:] *= np.float32(0.01)      # level drop: the holds keep the early peak
    lb = banks.LoudnessBank(omx, LoudnessConfig(), S, 6)
    ptr = lb.process_host(x, 256, 6, FS, capi.positions_fallback(6))
    holds = torch.empty(S * 3 * 16, device="cuda:0", dtype=torch.uint8)
    omx.check(omx.fn("peak_holds_reset", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_double, C.c_void_p])(holds.data_ptr(), 1, S * 3, 0.0, None))
    d_rows = torch.empty((S, blocks, 6), device="cuda:0", dtype=torch.float32)
    g = omx.fn("loudness_meters", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                            C.c_void_p, C.c_void_p, C.c_void_p])
    dt = 0.1                                 # coarse clock so that the 2 s hold expires inside the call
    omx.check(g(ptr, 1, S, blocks, capi.METER_TRUE_PEAK, capi.METER_RMS_FAST, 0.0, dt, holds.data_ptr(), None, d_rows.data_ptr()))
    torch.cuda.synchronize()
    got = d_rows.cpu().numpy()
    snaps = [lb.fetch(s, b) for s in range(S) for b in range(blocks)]
    oh = capi.peak_holds_reset(oracle, S * 3, 0.0)
    want = capi.loudness_meters(oracle, snaps, S, capi.METER_TRUE_PEAK, capi.METER_RMS_FAST, 0.0, dt, oh)
    assert np.array_equal(got[:, :, :3], want["values"]) and np.abs(got[:, :, 3:] - want["peaks"]).max() <= 1e-5
    assert np.array_equal(holds.cpu().numpy().view(capi.PEAK_HOLD_DTYPE)["db"], oh["db"])
    assert (want["peaks"][:, 20, 0] > want["values"][:, 20, 0] + 10.0).all() and (want["peaks"][:, -1, 0] < want["peaks"][:, 20, 0]).all()


@pytest.mark.parametrize("rate", [600.0, 1000.0, 3000.0])
def test_waveform_short_windows_match_oracle(omx, oracle, rate):
    """Low sample rates shrink the colour / history windows (2048 and 16384 samples at 44.1 kHz): 600 Hz -> 28 / 223 samples (the
    frame-at-a-time instantiation of the one-wavefront kernel), 1000 Hz -> 46 / 372 (its batched instantiation with the shortest
    windows it accepts; the role kernel too: >= 32), 3000 Hz -> 139 / 1115.  Windows wrap many times per call."""
    from openmeters_amd.capi import WaveformConfig, WaveformProcessor
    cfg = WaveformConfig(sample_rate=rate, scroll_speed=40.0, max_columns=512, analyze_bands=True, track_history=True)
    pcm = cfg4_pcm(11, 6000)
    a, b = WaveformProcessor(omx, cfg), WaveformProcessor(oracle, cfg)
    total = 0
    for lo, hi in [(0, 256), (256, 1300), (1300, 1301), (1301, 4000), (4000, 6000)]:
        g = a.process_block(AudioBlock(pcm[lo:hi].reshape(-1), 2, rate))
        w = b.process_block(AudioBlock(pcm[lo:hi].reshape(-1), 2, rate))
        assert g.reset == w.reset and g.columns.shape == w.columns.shape
        total += len(g.columns)
        assert np.array_equal(g.columns[:, :, :2].view(np.uint32), w.columns[:, :, :2].view(np.uint32))  # min / max
        if len(g.columns):
            bar("waveform (short windows): |d band colour| / max(1, max)", np.abs(g.columns[:, :, 2:5] - w.columns[:, :, 2:5]).max() /
                max(1.0, np.abs(w.columns[:, :, 2:5]).max()), 1e-6)
            bar("waveform (short windows): |d RMS history dB|", np.abs(g.columns[:, :, 5:] - w.columns[:, :, 5:]).max(), 2e-4)
    assert total > 50


def test_waveform_bank_matches_per_stream_oracle(omx, oracle):
    """bank of 7 streams, irregular block sizes: column counts / reset flags in lock-step, min / max bit-exact per stream"""
    from openmeters_amd.capi import WaveformConfig, WaveformProcessor
    S = 7
    cfg = WaveformConfig(scroll_speed=240.0, max_columns=256, analyze_bands=True, track_history=True)
    pcm = np.stack([cfg4_pcm(s, 256 * 90) for s in range(S)])
    bank = banks.WaveformBank(omx, cfg, S)
    bank.set_option(capi.OPT_KERNEL_FORM, 1)   # the sequential kernels' bars below (by shape the 2048 / 4096 / 9000-frame calls would go chunk-parallel)
    refs = [WaveformProcessor(oracle, cfg) for _ in range(S)]
    at, total = 0, 0
    for n in [256, 256, 1000, 37, 4096, 256, 2048, 9000, 256]:
        chunk = pcm[:, at:at + n]
        at += n
        up = bank.process_host(chunk, 2, FS)
        wants = [r.process_block(AudioBlock(chunk[s].reshape(-1), 2, FS)) for s, r in enumerate(refs)]
        assert (up is None) == (wants[0] is None)
        if up is None:
            continue
        for s in (0, 3, 6):
            w = wants[s]
            assert up.n_columns == len(w.columns) and bool(up.reset) == w.reset and bool(up.preview_some) == (w.preview is not None)
            got, prev = bank.fetch(s, int(up.n_columns), with_preview=True)
            assert np.array_equal(got[:, :, :2].view(np.uint32), w.columns[:, :, :2].view(np.uint32))
            if len(got):
                assert np.abs(got[:, :, 2:5] - w.columns[:, :, 2:5]).max() <= 1e-6 * max(1.0, np.abs(w.columns[:, :, 2:5]).max())
            if w.preview is not None:
                assert np.array_equal(prev[:, :2].view(np.uint32), w.preview[:, :2].view(np.uint32))
        total += int(up.n_columns)
    assert total > 50


def test_stereometer_non_finite_samples_reset_the_filters_like_the_reference(omx, oracle):
    """Biquad::process zeroes its state and output when the output is not finite (dsp.rs:428-431).  The HIP kernel runs 8-frame
    batches without that test and replays a batch frame by frame when its poison accumulator trips: inf / NaN / overflowing
    samples at scattered positions must give bit-identical band points and the same correlations as the frame-by-frame oracle."""
    cfg = StereometerConfig(analyze_bands=True, emit_band_points=True, correlation_window=0.05, segment_duration=0.01,
                            target_sample_count=480)
    pcm = cfg4_pcm(7, 256 * 16).copy()
    pcm[300, 0] = np.inf
    pcm[301, 1] = -np.inf
    pcm[777, 0] = np.nan
    pcm[1500:1503, :] = np.float32(3.0e38)    # finite input, overflows inside the cascades
    pcm[2300, 1] = np.float32(-3.4e38)
    pcm[3071, 0] = np.nan                     # last frame of a block
    pcm[3072, 1] = np.inf                     # first frame of the next
    a, b = StereometerProcessor(omx, cfg), StereometerProcessor(oracle, cfg)
    seen = 0
    for k in range(0, pcm.shape[0], 256):
        g = a.process_block(AudioBlock(pcm[k:k + 256].reshape(-1), 2, FS))
        w = b.process_block(AudioBlock(pcm[k:k + 256].reshape(-1), 2, FS))
        assert (g is None) == (w is None)
        if g is None:
            continue
        seen += 1
        assert np.array_equal(np.isnan(g.correlations), np.isnan(w.correlations))
        bar("stereometer: |d rho|", np.nanmax(np.abs(g.correlations - w.correlations), initial=0.0), 1e-6, k)
        for band in range(4):
            assert g.points[band].shape == w.points[band].shape
            gp, wp = g.points[band], w.points[band]
            both_nan = np.isnan(gp) & np.isnan(wp)
            assert np.array_equal(gp.view(np.uint32)[~both_nan], wp.view(np.uint32)[~both_nan]), (k, band)
    assert seen >= 10


def test_oscilloscope_two_pass_form_equals_block_by_block_calls(omx):
    """Calls with >= 4 blocks push every block first, estimate the periods of all (stream, block) in parallel and then run the
    stateful trigger pass; calls with fewer blocks do everything in one kernel.  Same functions on the same samples: headers and
    the newest snapshot must be BIT-identical, whatever mix of call shapes fed the bank (the ring grows on the first long call)."""
    S, blocks = 5, 48
    pcm = np.stack([cfg4_pcm(s + 3, 256 * blocks) for s in range(S)])
    a, b, c = (banks.OscilloscopeBank(omx, scope_cfg(), S) for _ in range(3))
    hdr_a, hdr_b, hdr_c = [], [], []

    def headers(bank, n):
        return [[tuple(bytes(bank.fetch(s, k)[0])) for s in range(S)] for k in range(n)]

    a.process_host(pcm, 256, 2, FS)                                   # one long call: two-pass
    hdr_a = headers(a, blocks)
    for k in range(blocks):                                           # block by block: single pass
        b.process_host(pcm[:, k * 256:(k + 1) * 256], 256, 2, FS)
        hdr_b += headers(b, 1)
    k = 0
    for n in (2, 10, 1, 3, 20, 12):                                   # mixed shapes
        c.process_host(pcm[:, k * 256:(k + n) * 256], 256, 2, FS)
        hdr_c += headers(c, n)
        k += n
    assert k == blocks and hdr_a == hdr_b == hdr_c
    for s in range(S):
        ha, sa = a.fetch(s, blocks - 1, with_samples=True)
        hb, sb = b.fetch(s, 0, with_samples=True)
        hc, sc = c.fetch(s, 11, with_samples=True)
        n = ha.samples_per_channel
        assert ha.produced and n == hb.samples_per_channel == hc.samples_per_channel
        for ch in range(ha.channels):   # what lies beyond samples_per_channel is not part of the snapshot
            assert np.array_equal(sa[ch, :n].view(np.uint32), sb[ch, :n].view(np.uint32))
            assert np.array_equal(sa[ch, :n].view(np.uint32), sc[ch, :n].view(np.uint32))


def test_stereometer_chunk_parallel_form_matches_oracle_and_sequential_form(omx, oracle):
    """stereometer_chunked.hip: every block of a bank call evaluated in parallel (zero-state pass, state scan, true-state pass,
    moment scan).  Against the oracle block by block (rho <= 1e-6, points <= 1e-6 of the full scale 1.0) and against the
    sequential kernels on the same bank input, over three calls: a long one, one whose state carries over, and a short one that
    falls back to the sequential kernels on the chunked path's state."""
    S = 6
    cfg = StereometerConfig(analyze_bands=True, emit_band_points=True, correlation_window=0.05, segment_duration=0.02,
                            target_sample_count=500)
    calls = [40, 24, 1]
    total = 256 * sum(calls)
    pcm = np.stack([cfg4_pcm(30 + s, total) for s in range(S)])
    pcm[2] *= np.float32(1e-3)        # a quiet stream
    pcm[3, :256 * 7] = 0.0            # leading silence: states and moments start from exact zeros
    chunk, seq = banks.StereometerBank(omx, cfg, S), banks.StereometerBank(omx, cfg, S)
    chunk.set_option(capi.OPT_KERNEL_FORM, 2)
    seq.set_option(capi.OPT_KERNEL_FORM, 1)
    refs = [StereometerProcessor(oracle, cfg) for _ in range(S)]
    at = 0
    for n_blocks in calls:
        part = pcm[:, at:at + 256 * n_blocks]
        chunk.process_host(part, 256, 2, FS)
        seq.process_host(part, 256, 2, FS)
        for s in range(S):
            w = None
            for blk in range(n_blocks):
                w = refs[s].process_block(AudioBlock(part[s, blk * 256:(blk + 1) * 256].reshape(-1), 2, FS))
                gc, pc = chunk.fetch(s, blk)
                gs, ps = seq.fetch(s, blk)
                assert pc == ps == (w is not None)
                if w is not None:
                    bar("stereometer: |d rho|", np.abs(gs - w.correlations).max(), 1e-6)
                    check_chunked_rho(gc, w.correlations, stereometer_band_rms(pcm[s, :at + 256 * (blk + 1)]), (s, blk))
            for band in range(4):
                got_c, got_s = chunk.fetch_points(s, band), seq.fetch_points(s, band)
                assert got_c.shape == got_s.shape == w.points[band].shape
                assert np.array_equal(got_s.view(np.uint32), w.points[band].view(np.uint32))       # sequential form: bit-exact biquads
                # points: the reference's f32 biquads are themselves 1e-5 ... 2e-5 away from exact arithmetic on these signals
                # (TDF-II sections with poles at 200 Hz / 48 kHz amplify every rounding error by ~fs / fc;
                # tests/test_kat_stereometer.py::test_band_filters_sit_on_an_f32_noise_floor measures it on the oracle), so
                # a second evaluation order differs from the first by that much: bar 1e-4 of full scale, measured 1.7e-5
                bar("stereometer (chunk-parallel): |d point| vs oracle", np.abs(got_c - w.points[band]).max(), 1e-4)
        at += 256 * n_blocks


def test_stereometer_chunk_parallel_form_hands_non_finite_input_to_the_sequential_kernels(omx, oracle):
    """Biquad::process zeroes a filter whose output is not finite (dsp.rs:428-431): not linear, so the chunk-parallel path
    detects it and the whole call is redone sequentially — the results must equal the sequential form bit for bit, and the oracle."""
    S, n_blocks = 4, 16
    cfg = StereometerConfig(analyze_bands=True, emit_band_points=True, correlation_window=0.05, segment_duration=0.02,
                            target_sample_count=300)
    pcm = np.stack([cfg4_pcm(50 + s, 256 * n_blocks) for s in range(S)])
    pcm[1, 777, 0] = np.inf
    pcm[2, 2000:2003, 1] = np.nan
    pcm[3, 1500, :] = np.float32(3e38)   # finite input, overflowing filter arithmetic
    chunk, seq = banks.StereometerBank(omx, cfg, S), banks.StereometerBank(omx, cfg, S)
    chunk.set_option(capi.OPT_KERNEL_FORM, 2)
    seq.set_option(capi.OPT_KERNEL_FORM, 1)
    chunk.process_host(pcm, 256, 2, FS)
    seq.process_host(pcm, 256, 2, FS)
    for s in range(S):
        p = StereometerProcessor(oracle, cfg)
        for blk in range(n_blocks):
            w = p.process_block(AudioBlock(pcm[s, blk * 256:(blk + 1) * 256].reshape(-1), 2, FS))
            gc, _ = chunk.fetch(s, blk)
            gs, _ = seq.fetch(s, blk)
            assert np.array_equal(gc.view(np.uint32), gs.view(np.uint32)), (s, blk)
            if w is not None:
                assert np.nanmax(np.abs(gs - w.correlations), initial=0.0) <= 1e-6
        for band in range(4):
            assert np.array_equal(chunk.fetch_points(s, band).view(np.uint32), seq.fetch_points(s, band).view(np.uint32))


def test_loudness_chunk_parallel_form_matches_oracle_and_alternates_with_the_sequential_form(omx, oracle):
    """loudness_chunked.hip: every block of a bank call in parallel (K-weighting by zero-state pass + scan + true-state pass,
    window sums from a prefix over 64-sample sub-block sums, true peak per block).  Against the oracle block by block (1e-4 dB)
    through a sequence of calls that alternates the two forms: chunked, chunked (windows fill, ring wraps), sequential (the
    chunked state must be usable), chunked again (running totals rebuilt from the ring), then a reset."""
    S, C = 5, 2
    calls = [(48, 2), (600, 2), (20, 1), (64, 2), (30, 2)]     # (blocks, form): 600 blocks = 3.2 s: the 3 s window fills and refreshes
    total = 256 * sum(n for n, _ in calls)
    pcm = np.stack([cfg3_pcm(70 + s, total, C) for s in range(S)])
    pcm[1, :256 * 30] = 0.0                                    # leading silence
    pcm[2] *= np.float32(1e-4)
    bank = banks.LoudnessBank(omx, LoudnessConfig(), S, C)
    refs = [LoudnessProcessor(oracle, LoudnessConfig()) for _ in range(S)]
    at = 0
    for n_blocks, form in calls:
        bank.set_option(capi.OPT_KERNEL_FORM, form)
        part = pcm[:, at:at + 256 * n_blocks]
        assert bank.process_host(part, 256, C, FS) is not None
        assert bank.last_form() == form
        check = sorted(set([0, 1, 2, n_blocks // 2, n_blocks - 2, n_blocks - 1]) & set(range(n_blocks)))
        for s in range(S):
            want = [refs[s].process_block(AudioBlock(part[s, k:k + 256].reshape(-1), C, FS)) for k in range(0, 256 * n_blocks, 256)]
            for blk in check:
                got = bank.fetch(s, blk)
                tag = "loudness (chunk-parallel)" if form == 2 else "loudness"
                bar(f"{tag}: |d short-term LUFS|", abs(got.short_term_loudness - want[blk].short_term_loudness), 1e-4, (s, blk))
                bar(f"{tag}: |d momentary LUFS|", abs(got.momentary_loudness - want[blk].momentary_loudness), 1e-4, (s, blk))
                for f in ("rms_fast_db", "rms_slow_db"):
                    bar(f"{tag}: |d {f}|", np.abs(getattr(got, f) - getattr(want[blk], f)).max(), 1e-4, (s, blk))
                bar(f"{tag}: |d true_peak_db|", np.abs(got.true_peak_db - want[blk].true_peak_db).max(), 1e-4, (s, blk))
                assert got.channel_count == want[blk].channel_count and got.positions == want[blk].positions
        at += 256 * n_blocks
    bank.reset_audio()
    refs = [LoudnessProcessor(oracle, LoudnessConfig()) for _ in range(S)]
    bank.set_option(capi.OPT_KERNEL_FORM, 2)
    part = pcm[:, :256 * 16]
    bank.process_host(part, 256, C, FS)
    for s in range(S):
        for k in range(0, 256 * 16, 256):
            w = refs[s].process_block(AudioBlock(part[s, k:k + 256].reshape(-1), C, FS))
        snapshots_close(bank.fetch(s, 15), w)


@pytest.mark.parametrize("channels,rate", [(8, 48000.0), (1, 96000.0), (4, 192000.0)] +
                         [(c, r) for r in (44100.0, 48000.0, 96000.0) for c in (2, 4, 5, 6)] + [(3, 44100.0), (7, 88200.0)])
def test_loudness_chunk_parallel_form_other_layouts_and_rates(omx, oracle, channels, rate):
    """Every channel count (1, 2, 4, 8 through LDS tiles; 3, 5, 6, 7 by per-lane reads: eight slots per stream as in the sequential
    kernels), SURROUND weights, 2x true-peak interpolation at 96 kHz, none at 192 kHz, and 44.1 / 88.2 kHz whose window lengths
    (17 640, 132 300 ... samples) are off the 64-sample sub-block grid (`tails`).  Long enough that the 0.4 s window is full and
    sliding; the chunked and the sequential bank must agree bit for bit on the true peak and within 1e-4 dB elsewhere."""
    S, block = 3, 1024 if rate > 48000.0 else 512
    n_blocks = int(0.55 * rate) // block + 1
    pcm = np.stack([cfg3_pcm(s, block * n_blocks, channels) for s in range(S)])
    positions = capi.SURROUND if channels == 8 else capi.positions_fallback(channels)
    a, b = banks.LoudnessBank(omx, LoudnessConfig(sample_rate=rate), S, channels), banks.LoudnessBank(omx, LoudnessConfig(sample_rate=rate), S, channels)
    a.set_option(capi.OPT_KERNEL_FORM, 2)
    b.set_option(capi.OPT_KERNEL_FORM, 1)
    a.process_host(pcm, block, channels, rate, positions)
    b.process_host(pcm, block, channels, rate, positions)
    assert (a.last_form(), b.last_form()) == (2, 1)
    for s in range(S):
        p = LoudnessProcessor(oracle, LoudnessConfig(sample_rate=rate))
        for blk in range(n_blocks):
            w = p.process_block(AudioBlock(pcm[s, blk * block:(blk + 1) * block].reshape(-1), channels, rate, positions))
            ga, gb = a.fetch(s, blk), b.fetch(s, blk)
            assert np.array_equal(ga.true_peak_db.view(np.uint32), gb.true_peak_db.view(np.uint32))
            snapshots_close(ga, w)
            snapshots_close(gb, w)


@pytest.mark.parametrize("channels", [2, 6])
def test_loudness_chunk_parallel_form_at_44100_alternates_with_the_sequential_form(omx, oracle, channels):
    """44.1 kHz, where no window length is a multiple of 64 samples and the squared-sample ring (132 300) is not either: chunked until
    the 3 s window is full and has refreshed, sequential (must continue from the chunked state: live / since-refresh sums taken at an
    off-grid refresh point), chunked again (running totals AND tails rebuilt from a ring whose oldest sub-block is partial), chunked."""
    S, block, rate = 2, 512, 44100.0
    calls = [(140, 2), (140, 2), (12, 1), (40, 2), (135, 2), (135, 2)]   # a call's sub-blocks + the ring's must fit the 4096-entry Q ring
    total = block * sum(n for n, _ in calls)
    pcm = np.stack([cfg3_pcm(170 + s, total, channels) for s in range(S)])
    pcm[1] *= np.float32(0.05)
    positions = capi.positions_fallback(channels)
    bank = banks.LoudnessBank(omx, LoudnessConfig(sample_rate=rate), S, channels)
    refs = [LoudnessProcessor(oracle, LoudnessConfig(sample_rate=rate)) for _ in range(S)]
    at = 0
    for n_blocks, form in calls:
        bank.set_option(capi.OPT_KERNEL_FORM, form)
        part = pcm[:, at:at + block * n_blocks]
        assert bank.process_host(part, block, channels, rate, positions) is not None
        assert bank.last_form() == form
        check = sorted(set([0, 1, 2, 34, 35, n_blocks // 2, 258, 259, n_blocks - 2, n_blocks - 1]) & set(range(n_blocks)))
        for s in range(S):
            want = [refs[s].process_block(AudioBlock(part[s, k:k + block].reshape(-1), channels, rate, positions))
                    for k in range(0, block * n_blocks, block)]
            for blk in check:
                snapshots_close(bank.fetch(s, blk), want[blk])
        at += block * n_blocks


@pytest.mark.parametrize("rebase", [4096, 1 << 22])
def test_loudness_chunk_parallel_form_keeps_its_accuracy_across_a_100_dB_drop_and_periodic_rebasing(omx, oracle, rebase):
    """Window sums of the chunk-parallel form are differences of running totals: a loud passage followed by a very quiet one is the
    case where the totals' size, not the window's, sets the error.  4 s at full scale, then a -100 dBFS tone: every snapshot of the
    quiet passage within 1e-4 dB of the oracle — with the totals taken afresh from the sample ring whenever a call finds them older
    than 4096 frames (the rebuild path runs at the start of every call after the first, OMX_OPT_LOUDNESS_REBASE_FRAMES) and with the
    default interval (no rebuild in this test: the double-double totals alone hold the bar)."""
    S, C, block = 3, 2, 256
    loud_blocks, quiet_blocks = 750, 300
    t = np.arange(block * (loud_blocks + quiet_blocks)) / FS
    pcm = np.empty((S, len(t), C), np.float32)
    for s in range(S):
        tone = np.sin(2 * np.pi * (500.0 + 37.0 * s) * t)
        gain = np.where(np.arange(len(t)) < block * loud_blocks, 0.9, 1e-5)
        pcm[s, :, 0] = (gain * tone).astype(np.float32)
        pcm[s, :, 1] = (0.8 * gain * tone).astype(np.float32)
    bank = banks.LoudnessBank(omx, LoudnessConfig(), S, C)
    bank.set_option(capi.OPT_KERNEL_FORM, 2)
    bank.set_option(capi.OPT_LOUDNESS_REBASE_FRAMES, rebase)
    refs = [LoudnessProcessor(oracle, LoudnessConfig()) for _ in range(S)]
    at = 0
    for n_blocks in (250, 250, 250, 150, 150):
        part = pcm[:, at:at + block * n_blocks]
        assert bank.process_host(part, block, C, FS) is not None and bank.last_form() == 2
        for s in range(S):
            want = [refs[s].process_block(AudioBlock(part[s, k:k + block].reshape(-1), C, FS)) for k in range(0, block * n_blocks, block)]
            for blk in sorted(set([0, 1, n_blocks // 3, n_blocks // 2, n_blocks - 2, n_blocks - 1])):
                snapshots_close(bank.fetch(s, blk), want[blk])
        at += block * n_blocks


def test_loudness_chunk_parallel_form_after_ninety_seconds_at_full_scale(omx, oracle):
    """as plain f64 running totals, 20 ... 90 s at full scale left 5e-4 ... 2e-3 dB in a -100 dBFS passage that followed (a
    window sum is the difference of two totals of ~4e6, each good to 9e-10).  The totals are double-double pairs now: 90 s at 0.9 of
    full scale with the periodic rebuild switched off, then the -100 dBFS tone, every compared snapshot within 1e-4 dB of the oracle."""
    S, C, block = 2, 2, 256
    loud_blocks, quiet_blocks = 16896, 512        # 90.1 s, 2.7 s
    n = block * (loud_blocks + quiet_blocks)
    t = np.arange(n) / FS
    pcm = np.empty((S, n, C), np.float32)
    for s in range(S):
        tone = np.sin(2 * np.pi * (500.0 + 37.0 * s) * t)
        gain = np.where(np.arange(n) < block * loud_blocks, 0.9, 1e-5)
        pcm[s, :, 0] = (gain * tone).astype(np.float32)
        pcm[s, :, 1] = (0.8 * gain * tone).astype(np.float32)
    bank = banks.LoudnessBank(omx, LoudnessConfig(), S, C)
    bank.set_option(capi.OPT_KERNEL_FORM, 2)
    bank.set_option(capi.OPT_LOUDNESS_REBASE_FRAMES, 1 << 40)
    refs = [LoudnessProcessor(oracle, LoudnessConfig()) for _ in range(S)]
    at = 0
    calls = [1024] * (loud_blocks // 1024) + [256, 256]
    for ci, n_blocks in enumerate(calls):
        part = pcm[:, at:at + block * n_blocks]
        assert bank.process_host(part, block, C, FS) is not None and bank.last_form() == 2
        quiet = at >= block * loud_blocks
        for s in range(S):
            want = [refs[s].process_block(AudioBlock(part[s, k:k + block].reshape(-1), C, FS)) for k in range(0, block * n_blocks, block)]
            if quiet or ci % 4 == 0:
                for blk in sorted(set([0, 1, n_blocks // 3, n_blocks // 2, n_blocks - 2, n_blocks - 1])):
                    snapshots_close(bank.fetch(s, blk), want[blk])
        at += block * n_blocks


def test_loudness_chunk_parallel_form_hands_non_finite_input_to_the_sequential_kernels(omx):
    S, C, n_blocks = 4, 2, 16
    pcm = np.stack([cfg3_pcm(90 + s, 256 * n_blocks * 2, C) for s in range(S)])
    pcm[1, 1000, 0] = np.nan
    pcm[3, 3333, 1] = np.inf
    a, b = banks.LoudnessBank(omx, LoudnessConfig(), S, C), banks.LoudnessBank(omx, LoudnessConfig(), S, C)
    a.set_option(capi.OPT_KERNEL_FORM, 2)
    b.set_option(capi.OPT_KERNEL_FORM, 1)
    for half in range(2):     # second call: finite again; the chunked form continues from the fallback's state and rebuilt totals
        part = pcm[:, half * 256 * n_blocks:(half + 1) * 256 * n_blocks]
        if half == 1:
            part = np.nan_to_num(part, nan=0.0, posinf=0.0)
        a.process_host(part, 256, C, FS)
        b.process_host(part, 256, C, FS)
        for s in range(S):
            for blk in (0, 5, n_blocks - 1):
                ga, gb = a.fetch(s, blk), b.fetch(s, blk)
                if half == 0:
                    for f in ("rms_fast_db", "rms_slow_db", "true_peak_db"):
                        assert np.array_equal(getattr(ga, f).view(np.uint32), getattr(gb, f).view(np.uint32)), (s, blk, f)
                    assert ga.short_term_loudness == gb.short_term_loudness or (np.isnan(ga.short_term_loudness) and np.isnan(gb.short_term_loudness))
                elif s in (0, 2):   # streams that never saw a non-finite sample
                    snapshots_close(ga, gb)


# ---- K10c: the waveform bank's chunk-parallel form (waveform_chunked.hip) ------------------------------------------------------
# Bars.  min / max: a reduction of the same samples — bit-identical.  Colour bands and RMS history: the band filters restart every
# 64 ... 256 frames from scanned states that carry one f32 rounding each (the low band's none).  A rounding of a DF2T state of a
# 200 Hz section excites the all-pole response 1 / A(z) — a near-double pole: (n + 1) r^n, peak fs / (2 pi e 0.707 fc) = 20 at
# 48 kHz — and the SAME mechanism acts on every step of the reference's own sequential f32 evaluation.  How far an f32 evaluation of
# these recurrences sits from the f64 recurrence therefore depends on the passage: ~1e-6 of the band's level on steady signal,
# and growing like eta t^1.5 ... eta t^2 / 2 through a free decay (t frames after a 120 dB drop the ringing that fills the fast
# window is 1e-3 away from exact in EITHER evaluation).  The bars are three-way, per call and per (field, band):
#     D_o  = |oracle - exact|, D_ho = |HIP - oracle|, D_h = |HIP - exact|     (exact = oracle/exact_f64.py::WaveformExact, f64 recurrence)
#     each as the maximum over the call's columns and channels of the difference relative to the loudest channel of that column;
#     D_ho <= FIX + 3 D_o   and   D_h <= FIX + 2 D_o,   FIX = 1e-5 (colour: the north star's tolerance), 2e-5 (power: its square), both x max(1, rate / 48 kHz).
# The sequential form's bars (1e-6, 2e-4 dB against the oracle; measured 0 and 1.1e-5) are unchanged.
WAVE_FIX_COLOUR, WAVE_FIX_POWER = 1e-5, 2e-5   # at <= 48 kHz; x rate / 48 kHz above (the response 1 / A(z) of the 200 Hz sections grows with fs / fc:
                                               # soak seed 9527360, 96 kHz, low band: HIP 2.9e-5 from exact in a call where the oracle sat at 4e-6)
# "Relative to" — the scale of a column's differences is the loudest channel of that band over the columns WITHIN REACH: the window's
# own length plus the memory of the 200 Hz sections (their free response falls by 200 dB in 26 ms; 50 ms are taken).  A window
# that has just lost a loud passage holds the passage's ringing, whose f32 error belongs to the passage's level (the 120 dB drop
# test: the fast window 170 frames after the drop is 1e-3 away from exact in EITHER evaluation, relative to its own -104 dB).
WAVE_MEMORY_S = 0.05


class WaveExact:
    """the f64 recurrence's columns of one stream from reset (oracle/exact_f64.py::WaveformExact) and the scales of the three-way bars"""

    def __init__(self, pcm_stream, rate, scroll=300.0, scroll_changes=()):
        sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
        import exact_f64 as ex
        model = ex.WaveformExact(rate, scroll)
        self.rate = float(rate)
        self.ends, self.colour, self.power = model.run(pcm_stream, scroll_changes)

        # Running maximum of the loudest channel's window mean over everything within `length` frames back of a column's end — sampled on the
        # model's 32-frame grid AND at the column ends, not at the column ends alone: at 10 columns per second (2205 frames apart at
        # 22.05 kHz, window 1024) a loud passage can begin and end between two columns, and the column right behind it — whose window
        # holds the passage's ringing — was then scaled by its own level (soak seeds 60000362 and 60038360, round 6: |HIP - exact| 4.5e-5
        # and 6.7e-5 of a column 33 frames behind a 37 dB drop, 1e-7 of the passage; `tools/debug/wave_seed.py`).
        def reach_max(top, grid_top, length):   # top [cols][1][3] at the column ends, grid_top [grid][1][3] at the grid frames
            out = np.empty_like(top)
            lo = np.searchsorted(self.ends, self.ends - length, side="left")
            glo = np.searchsorted(model.grid_ends, self.ends - length, side="left")
            ghi = np.searchsorted(model.grid_ends, self.ends, side="right")
            for c in range(len(self.ends)):
                out[c] = top[lo[c]:c + 1].max(axis=0)
                if ghi[c] > glo[c]:
                    out[c] = np.maximum(out[c], grid_top[glo[c]:ghi[c]].max(axis=0))
            return out
        memory = int(WAVE_MEMORY_S * rate)
        gmax = lambda x: x.max(axis=1, keepdims=True)
        self.top_colour = reach_max(gmax(self.colour), gmax(model.grid_colour), model.color_len + memory)
        self.top_power = np.stack([reach_max(gmax(self.power[:, :, 0]), gmax(model.grid_power[:, :, 0]), model.color_len + memory),
                                   reach_max(gmax(self.power[:, :, 1]), gmax(model.grid_power[:, :, 1]), model.slow_len + memory)], axis=2)   # [cols][1][2][3]

    def __len__(self):
        return len(self.ends)


def check_wave_three_way(tag, got, want, exact, cols, history, detail=None):
    """got / want: [n][4][11] f32 columns of the HIP bank and of the oracle = columns `cols` (a slice) of the stream `exact` describes"""
    rate_factor = max(1.0, exact.rate / 48000.0)

    def three(g, o, e, top, fix, name):
        fix = fix * rate_factor
        top = np.maximum(top, 1e-300)
        for band in range(3):
            sl = (..., band)
            d_o = float((np.abs(o - e) / top)[sl].max())
            d_ho = float((np.abs(g - o) / top)[sl].max())
            d_h = float((np.abs(g - e) / top)[sl].max())
            bar(f"{tag}: {name} |HIP - oracle| / (fix + 3 |oracle - exact|)", d_ho / (fix + 3.0 * d_o), 1.0, (detail, band, d_ho, d_o))
            bar(f"{tag}: {name} |HIP - exact| / (fix + 2 |oracle - exact|)", d_h / (fix + 2.0 * d_o), 1.0, (detail, band, d_h, d_o))
            bar(f"{tag}: {name} |oracle - exact| [recorded only]", d_o, 1e6, (detail, band))
            bar(f"{tag}: {name} |HIP - oracle| [recorded only]", d_ho, 1e6, (detail, band))
    g64, o64 = np.asarray(got, np.float64), np.asarray(want, np.float64)
    three(g64[:, :, 2:5], o64[:, :, 2:5], exact.colour[cols], exact.top_colour[cols], WAVE_FIX_COLOUR, "colour")
    if history:
        pg = 10.0 ** (g64[:, :, 5:].reshape(-1, 4, 2, 3) / 10.0)
        po = 10.0 ** (o64[:, :, 5:].reshape(-1, 4, 2, 3) / 10.0)
        e_power, e_top = np.maximum(exact.power[cols], 1e-14), np.maximum(exact.top_power[cols], 1e-14)   # power_to_db's floor: -140 dB
        live = e_power.max(axis=1, keepdims=True) > 1e-13
        for w, wname in enumerate(("fast", "slow")):
            keep = live[:, 0, w].any(axis=-1)
            if keep.any():
                three(pg[keep][:, :, w], po[keep][:, :, w], e_power[keep][:, :, w], e_top[keep][:, :, w], WAVE_FIX_POWER, f"{wname} history power")


@pytest.mark.parametrize("rate,history", [(48000.0, True), (48000.0, False), (44100.0, True), (96000.0, False)])
def test_waveform_chunk_parallel_form_matches_per_stream_oracle(omx, oracle, rate, history):
    """lock-step calls of mixed sizes with OMX_OPT_KERNEL_FORM = 2: calls of >= 1024 even frames take the chunk-parallel form, the
    others the sequential kernels — every hand-over of filter states, rings, compensated pairs and the open column goes both ways"""
    from openmeters_amd.capi import WaveformConfig, WaveformProcessor
    S = 5
    # (96 kHz: max_columns = 16 — the long calls emit more columns than that and keep the newest, cap_pending_columns :293-298)
    cfg = WaveformConfig(sample_rate=rate, scroll_speed=300.0, max_columns=16 if rate == 96000.0 else 1024, analyze_bands=True, track_history=history)
    sizes = [4096, 256, 2048, 1000, 8192, 1024, 3001, 1536, 16384, 512, 2050]
    pcm = np.stack([cfg4_pcm(40 + s, sum(sizes)) for s in range(S)])
    pcm[1, 9000:12000] *= np.float32(1e-4)   # a quiet passage inside one stream
    exact = [WaveExact(pcm[s], rate) for s in range(S)]
    bank = banks.WaveformBank(omx, cfg, S)
    bank.set_option(capi.OPT_KERNEL_FORM, 2)
    refs = [WaveformProcessor(oracle, cfg) for _ in range(S)]
    at, total, forms = 0, 0, []
    for n in sizes:
        chunk = pcm[:, at:at + n]
        at += n
        up = bank.process_host(chunk, 2, rate)
        forms.append(bank.last_form())
        for s, r in enumerate(refs):
            w = r.process_block(AudioBlock(chunk[s].reshape(-1), 2, rate))
            assert up.n_columns == len(w.columns) and bool(up.reset) == w.reset and bool(up.preview_some) == (w.preview is not None)
            got, prev = bank.fetch(s, int(up.n_columns), with_preview=True)
            assert np.array_equal(got[:, :, :2], w.columns[:, :, :2]), (n, s)   # min / max
            emitted = int(np.count_nonzero((exact[s].ends >= at - n) & (exact[s].ends < at)))   # columns ending inside this call
            assert len(got) == min(emitted, cfg.max_columns), (n, s)
            if len(got):
                cols = slice(total + emitted - len(got), total + emitted)                          # the newest max_columns of them are kept
                check_wave_three_way("waveform (chunk-parallel)", got, w.columns, exact[s], cols, history, (n, s))
            if w.preview is not None:
                assert np.array_equal(prev[:, :2], w.preview[:, :2]), (n, s)
                # (the preview column is compared with the oracle's under the widest column bar of the call: no exact twin is kept)
                assert np.abs(prev[:, 2:5] - w.preview[:, 2:5]).max() <= 1e-4 * max(1e-30, np.abs(w.preview[:, 2:5]).max()), (n, s)
        total += emitted
    assert forms == [2 if (n >= 1024 and n % 2 == 0) else 1 for n in sizes]
    assert total > 100 and total == len(exact[0])


def test_waveform_chunk_parallel_form_quiet_window_after_a_loud_passage(omx, oracle):
    """a window mean is a difference of two running totals: 1 s at full scale, then -120 dBFS — the history means fall by 120 dB
    and must still be those of the reference's compensated sums (a plain f64 difference would leave ~1e-13 of the loud total, 20 dB
    ABOVE the quiet passage's own mean power).  The calls around the drop also hold the free decay of the 200 Hz sections, where
    every f32 evaluation is ~1e-3 from exact: the three-way bar follows the oracle's own distance."""
    from openmeters_amd.capi import WaveformConfig, WaveformProcessor
    cfg = WaveformConfig(scroll_speed=300.0, max_columns=1024, analyze_bands=True, track_history=True)
    n = 16384
    rng = np.random.default_rng(5)
    loud = rng.uniform(-1.0, 1.0, (3 * n, 2)).astype(np.float32)
    quiet = (rng.uniform(-1.0, 1.0, (4 * n, 2)) * 1e-6).astype(np.float32)
    pcm = np.concatenate([loud, quiet])[None]
    exact = WaveExact(pcm[0], FS)
    bank = banks.WaveformBank(omx, cfg, 1)
    bank.set_option(capi.OPT_KERNEL_FORM, 2)
    ref = WaveformProcessor(oracle, cfg)
    floor_seen, total = 0.0, 0
    for k in range(0, pcm.shape[1], n):
        up = bank.process_host(pcm[:, k:k + n], 2, FS)
        assert bank.last_form() == 2
        w = ref.process_block(AudioBlock(pcm[0, k:k + n].reshape(-1), 2, FS))
        got, _ = bank.fetch(0, int(up.n_columns))
        assert np.array_equal(got[:, :, :2], w.columns[:, :, :2])
        cols = slice(total, total + len(got))
        check_wave_three_way("waveform (chunk-parallel, 120 dB drop)", got, w.columns, exact, cols, True, k)
        total += len(got)
        floor_seen = min(floor_seen, float(w.columns[:, :, 5:].min()))
    assert floor_seen < -110.0


@pytest.mark.parametrize("seed", [9001, 9002, 9003, 9004, 9005, 9006])
def test_waveform_chunk_parallel_random_sequences(omx, oracle, seed):
    """random rate / scroll speed / history flag / call sizes / level steps (tests/test_gpu_soak.py runs this on seeds nobody picked):
    three streams, one of them with nearly equal sides (a Side band 50 dB under its Left / Right bands), levels stepping over 80 dB"""
    from openmeters_amd.capi import WaveformConfig, WaveformProcessor
    rng = np.random.default_rng(seed)
    rate = float(rng.choice([22050.0, 32000.0, 44100.0, 48000.0, 96000.0]))
    scroll = float(rng.choice([10.0, 77.7, 300.0, 650.0, 1000.0]))
    history = bool(rng.integers(2))
    sizes = [int(x) for x in rng.choice([1024, 1536, 2048, 4096, 6000, 8192, 12288, 256, 1000, 3001], size=8)]
    S, total_frames = 3, sum(sizes)
    pcm = np.stack([cfg4_pcm(int(seed % 1000) * 3 + s, total_frames) for s in range(S)])
    pcm[2, :, 1] = pcm[2, :, 0] * np.float32(0.994) + pcm[2, :, 1] * np.float32(0.003)    # nearly mono
    at = 0
    while at < total_frames:   # level steps
        n = int(rng.integers(500, 6000))
        pcm[:, at:at + n] *= np.float32(10.0 ** float(rng.uniform(-4.0, 0.0)))
        at += n
    cfg = WaveformConfig(sample_rate=rate, scroll_speed=scroll, max_columns=4096, analyze_bands=True, track_history=history)
    exact = [WaveExact(pcm[s], rate, scroll) for s in range(S)]
    bank = banks.WaveformBank(omx, cfg, S)
    bank.set_option(capi.OPT_KERNEL_FORM, 2)
    refs = [WaveformProcessor(oracle, cfg) for _ in range(S)]
    at, total, chunked = 0, 0, 0
    for n in sizes:
        chunk = pcm[:, at:at + n]
        at += n
        up = bank.process_host(chunk, 2, rate)
        chunked += bank.last_form() == 2
        for s, r in enumerate(refs):
            w = r.process_block(AudioBlock(chunk[s].reshape(-1), 2, rate))
            assert up.n_columns == len(w.columns) and bool(up.reset) == w.reset and bool(up.preview_some) == (w.preview is not None)
            got, prev = bank.fetch(s, int(up.n_columns), with_preview=True)
            assert np.array_equal(got[:, :, :2], w.columns[:, :, :2]), (seed, n, s)
            if len(got):
                cols = slice(total, total + len(got))
                check_wave_three_way("waveform (chunk-parallel, random sequences)", got, w.columns, exact[s], cols, history, (seed, rate, scroll, n, s))
            if w.preview is not None:
                assert np.array_equal(prev[:, :2], w.preview[:, :2]), (seed, n, s)
        total += int(up.n_columns)
    assert total == len(exact[0])
    assert chunked >= sum(1 for n in sizes if n >= 1024 and n % 2 == 0) - 8 * (scroll >= 650.0)   # (thousands of columns per call: sequential)


@pytest.mark.parametrize("rate,history,sizes", [
    (48000.0, True, [2048] * 14 + [1024] * 5 + [256] + [2048] * 11 + [4096, 4096, 16384, 16384, 2048]),
    (48000.0, False, [2048] * 6 + [1000] + [4096] * 4 + [1024] * 4),
    (44100.0, True, [16384] * 2 + [4096] * 6 + [16384] + [6000] * 4),
])
def test_waveform_chunk_parallel_form_keeps_running_totals_between_calls(omx, oracle, rate, history, sizes):
    """lock-step calls of the chunk-parallel form leave the double-double running total at every push count a later call starts a
    window at (wave_keep_totals_kernel) and a later call takes its old segments from those instead of the rings.  Runs of equal calls
    (every start was foreseen), changes of the call length (the pseudo-column's starts were not: those segments come from the
    rings), a sequential call in between (the table empties and refills), a scroll-speed change (columns end where nobody foresaw),
    windows that reach back over 17 calls, a quiet passage after a loud one."""
    from openmeters_amd.capi import WaveformConfig, WaveformProcessor
    S, total_frames = 2, sum(sizes)
    change_at = sum(sizes[:len(sizes) // 2])
    pcm = np.stack([cfg4_pcm(70 + s, total_frames) for s in range(S)])
    pcm[1, total_frames // 3:total_frames // 3 + 30000] *= np.float32(1e-5)
    cfg = WaveformConfig(sample_rate=rate, scroll_speed=300.0, max_columns=1024, analyze_bands=True, track_history=history)
    exact = [WaveExact(pcm[s], rate, 300.0, [(change_at, 420.0)]) for s in range(S)]
    bank = banks.WaveformBank(omx, cfg, S)
    bank.set_option(capi.OPT_KERNEL_FORM, 2)
    refs = [WaveformProcessor(oracle, cfg) for _ in range(S)]
    at, total = 0, 0
    for n in sizes:
        if at == change_at:
            cfg.scroll_speed = 420.0
            bank.update_config(cfg)
            for r in refs:
                r.update_config(cfg)
        chunk = pcm[:, at:at + n]
        at += n
        up = bank.process_host(chunk, 2, rate)
        assert bank.last_form() == (2 if n >= 1024 and n % 2 == 0 else 1)
        for s, r in enumerate(refs):
            w = r.process_block(AudioBlock(chunk[s].reshape(-1), 2, rate))
            assert up.n_columns == len(w.columns)
            got, _ = bank.fetch(s, int(up.n_columns))
            assert np.array_equal(got[:, :, :2], w.columns[:, :, :2]), (n, s)
            if len(got):
                check_wave_three_way("waveform (chunk-parallel, kept totals)", got, w.columns, exact[s], slice(total, total + len(got)), history, (at, n, s))
        total += int(up.n_columns)
    assert total == len(exact[0])


@pytest.mark.parametrize("history", [False, True])
def test_waveform_chunk_parallel_form_in_ragged_calls(omx, oracle, history):
    """ragged calls whose streams fall into a few lock-step groups (the same frame count, push count and column phase) run one plan per
    group; a lock-step prefix, groups that split and re-merge, a stream that pauses, and a call the chunk form does not serve (an odd
    count) in between.  Every stream against its own oracle and its own f64 recurrence, counters (column counts, preview progress)
    exactly."""
    import torch
    from openmeters_amd.capi import WaveformConfig, WaveformProcessor
    S, cap = 6, 8192
    cfg = WaveformConfig(scroll_speed=300.0, max_columns=1024, analyze_bands=True, track_history=history)
    plan = [("lock", 4096),
            ("ragged", [4096, 4096, 2048, 4096, 2048, 0]),
            ("ragged", [2048, 2048, 4096, 2048, 4096, 0]),      # streams 0-4 meet again at 10240 frames; stream 5 paused
            ("ragged", [8192, 8192, 8192, 1024, 1024, 8192]),
            ("ragged", [1001, 2048, 2048, 2048, 2048, 2048]),   # an odd count: the sequential kernels do this call
            ("ragged", [3072, 2026, 2026, 2026, 2026, 2026]),
            ("ragged", [6144] * 6),
            ("ragged", [4096, 4096, 4096, 0, 4096, 4096], [0, 1, 0, 1, 0, 0]),   # reset_audio of stream 1 (with frames) and of stream 3 (without)
            ("ragged", [4096] * 6)]
    totals = [sum((p[1] if p[0] == "lock" else p[1][s]) for p in plan) for s in range(S)]
    feeds = [cfg4_pcm(70 + s, totals[s]) for s in range(S)]
    exact = [WaveExact(feeds[s], FS) for s in range(S)]
    bank = banks.WaveformBank(omx, cfg, S)
    bank.set_option(capi.OPT_KERNEL_FORM, 2)
    refs = [WaveformProcessor(oracle, cfg) for _ in range(S)]
    pos = capi.positions_fallback(2)
    at, cols_seen, forms = [0] * S, [0] * S, []
    for entry in plan:
        kind, counts = entry[0], entry[1]
        mask = entry[2] if len(entry) > 2 else None
        if kind == "lock":
            chunk = np.stack([feeds[s][at[s]:at[s] + counts] for s in range(S)])
            up = bank.process_host(chunk, 2, FS)
            forms.append(bank.last_form())
            for s in range(S):
                w = refs[s].process_block(AudioBlock(chunk[s].reshape(-1), 2, FS))
                got, _ = bank.fetch(s, int(up.n_columns))
                assert np.array_equal(got[:, :, :2], w.columns[:, :, :2])
                at[s] += counts
                cols_seen[s] += len(w.columns)
            continue
        pcm = np.zeros((S, cap, 2), np.float32)
        for s in range(S):
            pcm[s, :counts[s]] = feeds[s][at[s]:at[s] + counts[s]]
        d_pcm = torch.from_numpy(pcm).to("cuda:0")
        up = bank.process_ragged(d_pcm.data_ptr(), cap, counts, 2, FS, pos, mask)
        torch.cuda.synchronize()
        forms.append(bank.last_form())
        for s in range(S):
            if mask and mask[s]:   # the stream starts over: its oracle, and the f64 recurrence from here on
                refs[s].reset_audio()
                exact[s] = WaveExact(feeds[s][at[s]:], FS)
                cols_seen[s] = 0
        n_cols = torch.as_tensor(_DevView(up.d_n_columns, (S,), "<u4"), device="cuda:0").cpu().numpy()
        progress = torch.as_tensor(_DevView(up.d_preview_progress, (S,), "<f4"), device="cuda:0").cpu().numpy()
        M = int(up.max_columns)
        for s in range(S):
            if counts[s] == 0:
                assert int(n_cols[s]) == 0
                continue
            w = refs[s].process_block(AudioBlock(pcm[s, :counts[s]].reshape(-1), 2, FS))
            assert int(n_cols[s]) == len(w.columns) and progress[s] == np.float32(w.preview_progress), (kind, counts, s)
            got, prev = bank.fetch(s, M, with_preview=True)
            got = got[:len(w.columns)]
            assert np.array_equal(got[:, :, :2], w.columns[:, :, :2]), (counts, s)
            if len(got):
                cols = slice(cols_seen[s], cols_seen[s] + len(got))
                check_wave_three_way("waveform (chunk-parallel, ragged groups)", got, w.columns, exact[s], cols, history, (counts, s))
            if w.preview is not None:
                assert np.array_equal(prev[:, :2], w.preview[:, :2]), (counts, s)
                assert np.abs(prev[:, 2:5] - w.preview[:, 2:5]).max() <= 1e-4 * max(1e-30, np.abs(w.preview[:, 2:5]).max()), (counts, s)
            at[s] += counts[s]
            cols_seen[s] += len(w.columns)
    assert forms == [2, 2, 2, 2, 1, 2, 2, 2, 2]
    assert cols_seen == [len(e) for e in exact]


class _DevView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}


def test_waveform_chunk_parallel_form_hands_non_finite_input_to_the_sequential_kernels(omx, oracle):
    """NaN / inf / absurdly large samples: the chunk-parallel form raises its flag before it has written anything but scratch and
    the sequential kernels do the call — bit-identical to the sequential form, continuity rules of :275-291 included"""
    from openmeters_amd.capi import WaveformConfig
    cfg = WaveformConfig(scroll_speed=300.0, max_columns=1024, analyze_bands=True, track_history=True)
    S, n = 3, 4096
    pcm = np.stack([cfg4_pcm(60 + s, 4 * n) for s in range(S)])
    pcm[1, n + 100, 0] = np.nan
    pcm[2, n + 3000, 1] = np.inf
    pcm[0, 2 * n + 17, 0] = np.float32(1e30)
    a, b = banks.WaveformBank(omx, cfg, S), banks.WaveformBank(omx, cfg, S)
    a.set_option(capi.OPT_KERNEL_FORM, 2)
    b.set_option(capi.OPT_KERNEL_FORM, 1)
    for k in range(0, 4 * n, n):
        ua, ub = a.process_host(pcm[:, k:k + n], 2, FS), b.process_host(pcm[:, k:k + n], 2, FS)
        assert ua.n_columns == ub.n_columns
        for s in range(S):
            ga, pa = a.fetch(s, int(ua.n_columns), with_preview=True)
            gb, pb = b.fetch(s, int(ub.n_columns), with_preview=True)
            assert np.array_equal(ga[:, :, :2].view(np.uint32), gb[:, :, :2].view(np.uint32)), (k, s)
            fin = np.isfinite(gb[:, :, 2:5])
            assert np.array_equal(fin, np.isfinite(ga[:, :, 2:5]))
            assert np.abs(ga[:, :, 2:5][fin] - gb[:, :, 2:5][fin]).max(initial=0.0) <= 4.0 * WAVE_FIX_COLOUR * max(1.0, np.abs(gb[:, :, 2:5][fin]).max(initial=0.0))
