"""Long streaming runs: every ring wraps several times (spectrogram / spectrum pending-audio rings, the 3 s loudness ring,
refresh of the compensated sums), many calls with irregular sizes; the state after minutes of audio must still match the
oracle fed the same stream."""
import numpy as np
import pytest

from openmeters_amd import banks, capi
from openmeters_amd.capi import (AudioBlock, LoudnessConfig, LoudnessProcessor, SpectrogramConfig, SpectrogramProcessor,
                                 SpectrumConfig, SpectrumProcessor)
from parity import check_reassigned_columns, reassigned_column_metrics
from test_gpu_parity import check_trace

pytestmark = pytest.mark.gpu
FS = 48000.0


def long_signal(s, frames):
    t = np.arange(frames) / FS
    rng = np.random.default_rng(100 + s)
    x = 0.3 * np.sin(2 * np.pi * (220.0 * (s + 1)) * t + 3.0 * np.sin(2 * np.pi * 0.31 * t)) + 0.01 * rng.standard_normal(frames)
    return np.stack([x, 0.6 * x + 0.005 * rng.standard_normal(frames)], 1).astype(np.float32)


def test_loudness_ring_wraps_and_sums_refresh(omx, oracle):
    """12 s of audio: the 144 000-slot ring wraps twice, the 14 400 / 19 200 / 48 000-sample windows refresh many times"""
    S, frames = 3, 256 * 2250
    pcm = np.stack([long_signal(s, frames) for s in range(S)])
    bank = banks.LoudnessBank(omx, LoudnessConfig(), S, 2)
    refs = [LoudnessProcessor(oracle, LoudnessConfig()) for _ in range(S)]
    at = 0
    for n_blocks in [10, 1, 300, 77, 500, 12, 850, 500]:
        n = 256 * n_blocks
        chunk = pcm[:, at:at + n]
        at += n
        bank.process_host(chunk, 256, 2, FS)
        for s in range(S):
            for k in range(0, n, 256):
                w = refs[s].process_block(AudioBlock(chunk[s, k:k + 256].reshape(-1), 2, FS))
            g = bank.fetch(s, n_blocks - 1)
            assert abs(g.momentary_loudness - w.momentary_loudness) <= 1e-4 and abs(g.short_term_loudness - w.short_term_loudness) <= 1e-4
            assert np.abs(g.rms_fast_db - w.rms_fast_db).max() <= 1e-4 and np.abs(g.rms_slow_db - w.rms_slow_db).max() <= 1e-4
            assert np.abs(g.true_peak_db - w.true_peak_db).max() <= 1e-4
    assert at == frames


def test_spectrogram_and_spectrum_rings_wrap_over_many_calls(omx, oracle):
    """~13 s in 300 irregular calls: the pending-audio rings (a few thousand samples) wrap hundreds of times"""
    rng = np.random.default_rng(7)
    frames = 256 * 2500
    pcm = long_signal(1, frames)
    sg_cfg = SpectrogramConfig(fft_size=2048, hop_size=64, use_reassignment=True, history_length=16)
    sp_cfg = SpectrumConfig(fft_size=4096, hop_size=1024, averaging_mode=capi.AVG_EXPONENTIAL, averaging_param=0.8)
    a, b = SpectrogramProcessor(omx, sg_cfg), SpectrogramProcessor(oracle, sg_cfg)
    c, d = SpectrumProcessor(omx, sp_cfg), SpectrumProcessor(oracle, sp_cfg)
    at, calls, checked = 0, 0, 0
    while at < frames:
        n = int(rng.choice([256, 256, 256, 1024, 100, 3000, 5000]))
        n = min(n, frames - at)
        blk = AudioBlock(pcm[at:at + n].reshape(-1), 2, FS)
        at += n
        calls += 1
        g, w = a.process_block(blk), b.process_block(blk)
        sg, sw = c.process_block(blk), d.process_block(blk)
        assert (g is None) == (w is None) and (sg is None) == (sw is None)
        if w is not None:
            assert len(g.new_columns) == len(w.new_columns)
            if calls % 25 == 0 and w.new_columns:
                check_reassigned_columns([g.new_columns[-1]], [w.new_columns[-1]], FS, 64)
                checked += 1
        if sw is not None and calls % 25 == 0:
            check_trace(sg.traces[0][0], sw.traces[0][0])
    assert calls > 250 and checked >= 8


@pytest.mark.parametrize("W,hop,zp", [(8192, 2048, 1), (16384, 4096, 1), (4096, 1024, 4), (4096, 256, 1)])
def test_long_window_kernels_cross_the_ring_wrap(omx, oracle, W, hop, zp):
    """The 8192 / 16384-point kernels (and the pair kernel) read their window with buffer loads off one lane offset while the window lies
    in one piece of the pending-audio ring, and index by index through the wrap mask when it does not: irregular calls (odd sizes, so
    odd and even window starts) until the ring has wrapped several times, the newest column of every call against the oracle."""
    rng = np.random.default_rng(W + zp)
    frames = 40 * W
    pcm = long_signal(2, frames)
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, zero_padding_factor=zp, use_reassignment=True, history_length=8)
    a, b = SpectrogramProcessor(omx, cfg), SpectrogramProcessor(oracle, cfg)
    at, checked = 0, 0
    while at < frames:
        n = min(int(rng.choice([3 * hop + 1, 2 * hop, 5 * hop - 3, hop // 2 + 7, 4 * W + 5])), frames - at)
        blk = AudioBlock(pcm[at:at + n].reshape(-1), 2, FS)
        at += n
        g, w = a.process_block(blk), b.process_block(blk)
        assert (g is None) == (w is None)
        if w is not None:
            assert len(g.new_columns) == len(w.new_columns) and g.reset == w.reset
            if w.new_columns:
                check_reassigned_columns([g.new_columns[-1]], [w.new_columns[-1]], FS, hop)
                checked += 1
    assert checked >= 10

