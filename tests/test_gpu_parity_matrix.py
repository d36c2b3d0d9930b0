"""Configuration matrix for the spectrogram / spectrum parity (`-m gpu`): every WindowKind, several sample rates and
channel layouts, through the fused 4096 kernels and the generic ones.  Tolerances as in test_gpu_parity.py."""
import numpy as np
import pytest

from openmeters_amd import capi
from openmeters_amd.capi import (AudioBlock, SpectrogramConfig, SpectrogramProcessor, SpectrumConfig, SpectrumProcessor)
from test_gpu_parity import check_reassigned, check_trace, stream_pcm
from parity import check_reassigned_columns, check_classic

pytestmark = pytest.mark.gpu
WINDOWS = [capi.WINDOW_RECTANGULAR, capi.WINDOW_HANN, capi.WINDOW_HAMMING, capi.WINDOW_BLACKMAN, capi.WINDOW_BLACKMAN_HARRIS]


def multichannel(s, frames, channels):
    base = stream_pcm(s, frames)[:, 0]
    out = np.zeros((frames, channels), np.float32)
    for c in range(channels):
        out[:, c] = np.roll(base, 37 * c) * np.float32(1.0 - 0.1 * c)
    return out


@pytest.mark.parametrize("window", WINDOWS)
@pytest.mark.parametrize("W", [4096, 1024])
def test_reassigned_every_window_kind(omx, oracle, window, W):
    """derivative window (spectral derivative, on the device), time-weighted window (rebuilt in registers in the fused kernel)
    and the bin normalisation all depend on the window kind"""
    cfg = SpectrogramConfig(fft_size=W, hop_size=256, window=window, use_reassignment=True, history_length=64)
    pcm = stream_pcm(3, 2 * W + 256 * 5)
    blk = AudioBlock(pcm.reshape(-1), 2, 48000.0)
    g, w = SpectrogramProcessor(omx, cfg).process_block(blk), SpectrogramProcessor(oracle, cfg).process_block(blk)
    assert len(g.new_columns) == len(w.new_columns) == 6 and g.reassigned_power_scale == w.reassigned_power_scale
    check_reassigned(g.new_columns, w.new_columns, 256)


@pytest.mark.parametrize("window", WINDOWS)
@pytest.mark.parametrize("W,zp,hop", [(8192, 1, 1024), (16384, 1, 2048), (2048, 8, 256)])
def test_reassigned_big_transforms_every_window_kind(omx, oracle, window, W, zp, hop):
    """8192 (one dual 4096-point transform per transform, stft8192_kernels.hip) and 16384 (four, stft16384_kernels.hip; window 16384
    and a zero-padded one): Hann / Hamming run the kernels that window on the bins, the other kinds the table-driven ones"""
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, window=window, zero_padding_factor=zp, use_reassignment=True, history_length=16)
    pcm = stream_pcm(5, 2 * W + hop * 2)
    blk = AudioBlock(pcm.reshape(-1), 2, 48000.0)
    g, w = SpectrogramProcessor(omx, cfg).process_block(blk), SpectrogramProcessor(oracle, cfg).process_block(blk)
    assert len(g.new_columns) == len(w.new_columns) == 3 and g.reassigned_power_scale == w.reassigned_power_scale
    check_reassigned_columns(g.new_columns, w.new_columns, 48000.0, hop)


@pytest.mark.parametrize("W,zp,hop", [(1024, 32, 256), (2048, 16, 64), (2048, 32, 64), (4096, 8, 256), (8192, 4, 512), (16384, 2, 1024), (4096, 32, 256),
                                      (16384, 8, 2048)])
def test_reassigned_transforms_beyond_16384_points(omx, oracle, W, zp, hop):
    """zero padding to F = zp W > 16384 points (the GUI reaches 524288): zp W-point transforms of modulated slices per spectrum
    (stft_pow2_kernels.hip, windowed_residue_kernel) against the oracle's single F-point transforms"""
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, zero_padding_factor=zp, use_reassignment=True, history_length=16)
    pcm = stream_pcm(6, 2 * W + hop * 2)
    blk = AudioBlock(pcm.reshape(-1), 2, 48000.0)
    g, w = SpectrogramProcessor(omx, cfg).process_block(blk), SpectrogramProcessor(oracle, cfg).process_block(blk)
    assert g.fft_size == w.fft_size == W * zp
    assert len(g.new_columns) == len(w.new_columns) == 3 and g.reassigned_power_scale == w.reassigned_power_scale
    check_reassigned_columns(g.new_columns, w.new_columns, 48000.0, hop)


@pytest.mark.parametrize("W,zp,hop,window", [(1024, 32, 256, capi.WINDOW_HANN), (2048, 16, 512, capi.WINDOW_BLACKMAN_HARRIS), (4096, 8, 1024, capi.WINDOW_HAMMING),
                                             (16384, 4, 4096, capi.WINDOW_HANN), (8192, 32, 2048, capi.WINDOW_BLACKMAN)])
def test_classic_transforms_beyond_16384_points(omx, oracle, W, zp, hop, window):
    """classic (non-reassigned) columns zero-padded to F = zp W > 16384 points: classic_residue_kernel against the oracle's F-point real
    transform, at the fused classic kernels' bars (codes within 40 dB of the maximum at most one apart, weaker bins inside the f32
    transform noise budget)"""
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, window=window, zero_padding_factor=zp, use_reassignment=False, history_length=16)
    pcm = stream_pcm(8, W + hop * 3)
    blk = AudioBlock(pcm.reshape(-1), 2, 48000.0)
    g, w = SpectrogramProcessor(omx, cfg).process_block(blk), SpectrogramProcessor(oracle, cfg).process_block(blk)
    assert g.fft_size == w.fft_size == W * zp and len(g.new_columns) == len(w.new_columns) == 4
    assert all(len(c) == W * zp // 2 + 1 for c in g.new_columns)
    check_classic(g.new_columns, w.new_columns)


@pytest.mark.parametrize("window", WINDOWS)
def test_reassigned_transforms_beyond_16384_points_every_window_kind_in_a_bank(omx, oracle, window):
    """2048 x 16 in a bank of three streams (one of them silent: empty columns), every window kind (the residue form windows in the
    time domain from the reference's tables)"""
    from openmeters_amd import banks
    W, zp, hop, S, ncols = 2048, 16, 128, 3, 4
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, window=window, zero_padding_factor=zp, use_reassignment=True, history_length=16)
    pcm = np.stack([stream_pcm(7 + s, 2 * W + hop * (ncols - 1)) for s in range(S)])
    pcm[1] = 0.0
    bank = banks.SpectrogramBank(omx, cfg, S)
    up = bank.process_host(pcm, 2, 48000.0)
    assert up.n_columns == ncols
    for s in range(S):
        want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm[s].reshape(-1), 2, 48000.0)).new_columns
        got = [bank.fetch_column(s, c, capi.COLUMN_REASSIGNED, W * zp // 2 + 1) for c in range(ncols)]
        assert [len(c) for c in got] == [len(c) for c in want] or s != 1
        if s == 1:
            assert all(len(c) == 0 for c in got)
        else:
            check_reassigned_columns(got, want, 48000.0, hop)


@pytest.mark.parametrize("window", WINDOWS)
def test_classic_and_spectrum_every_window_kind(omx, oracle, window):
    pcm = stream_pcm(4, 4096 + 256 * 7)
    blk = AudioBlock(pcm.reshape(-1), 2, 48000.0)
    cfg = SpectrogramConfig(fft_size=2048, hop_size=256, window=window, use_reassignment=False, history_length=64)
    g, w = SpectrogramProcessor(omx, cfg).process_block(blk), SpectrogramProcessor(oracle, cfg).process_block(blk)
    assert len(g.new_columns) == len(w.new_columns) > 0
    check_classic(g.new_columns, w.new_columns)
    for N in (4096, 512):
        sc = SpectrumConfig(fft_size=N, hop_size=N // 4, window=window, source=capi.CH_LEFT, secondary_source=capi.CH_SIDE)
        gs, ws = SpectrumProcessor(omx, sc).process_block(blk), SpectrumProcessor(oracle, sc).process_block(blk)
        assert np.array_equal(gs.frequency_bins, ws.frequency_bins)
        for tr in range(2):
            for k in range(2):
                check_trace(gs.traces[tr][k], ws.traces[tr][k])


@pytest.mark.parametrize("rate,channels,positions", [(44100.0, 1, None), (96000.0, 6, None), (192000.0, 8, capi.SURROUND),
                                                     (48000.0, 4, [capi.POS_FC, capi.POS_LFE, capi.POS_SL, capi.POS_SR]),
                                                     (22050.0, 3, [capi.POS_AUX0, capi.POS_AUX0 + 1, capi.POS_UNKNOWN])])
def test_sample_rates_and_channel_layouts(omx, oracle, rate, channels, positions):
    """the stereo fold (dsp.rs:117-176) and every rate-derived constant (bin_hz, latency, max_hz) feed the fused kernel"""
    frames = 8192 + 256 * 4
    pcm = multichannel(6, frames, channels)
    if positions is not None:
        positions = list(positions) + [capi.POS_UNKNOWN] * (8 - len(positions))
    blk = AudioBlock(pcm.reshape(-1), channels, rate, positions)
    cfg = SpectrogramConfig(sample_rate=rate, fft_size=4096, hop_size=256, use_reassignment=True, history_length=64)
    g, w = SpectrogramProcessor(omx, cfg).process_block(blk), SpectrogramProcessor(oracle, cfg).process_block(blk)
    assert (g is None) == (w is None)
    if w is not None:
        assert len(g.new_columns) == len(w.new_columns) == 5
        for h, o in zip(g.new_columns, w.new_columns):
            from parity import reassigned_column_metrics
            check_reassigned_columns([h], [o], rate, 256)
    sc = SpectrumConfig(sample_rate=rate, fft_size=4096, hop_size=512, source=capi.CH_MID, secondary_source=capi.CH_RIGHT)
    gs, ws = SpectrumProcessor(omx, sc).process_block(blk), SpectrumProcessor(oracle, sc).process_block(blk)
    assert (gs is None) == (ws is None)
    if ws is not None:
        for tr in range(2):
            for k in range(2):
                check_trace(gs.traces[tr][k], ws.traces[tr][k])


@pytest.mark.parametrize("W,zp,hop", [(1024, 2, 256), (512, 4, 100), (2048, 2, 64), (256, 16, 64), (4096, 4, 1024), (64, 32, 16)])
def test_zero_padded_classic_columns(omx, oracle, W, zp, hop):
    """window W zero-padded to F = zp W in {1024 ... 16384}: fused two-columns-per-FFT kernel (mean over the W window samples,
    zeros beyond, window.rs:66-88 + processor.rs:350-368); 64 x 32 = 2048"""
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, zero_padding_factor=zp, use_reassignment=False, history_length=64)
    pcm = stream_pcm(8, W + hop * 10)
    blk = AudioBlock(pcm.reshape(-1), 2, 48000.0)
    g, w = SpectrogramProcessor(omx, cfg).process_block(blk), SpectrogramProcessor(oracle, cfg).process_block(blk)
    assert len(g.new_columns) == len(w.new_columns) == 11 and g.fft_size == w.fft_size == W * zp
    check_classic(g.new_columns, w.new_columns)
