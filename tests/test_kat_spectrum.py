"""Known-answer tests ported from reference src/visuals/spectrum/processor.rs:432-678."""
import ctypes as C

import numpy as np
import pytest

from openmeters_amd import capi
from openmeters_amd.capi import AudioBlock, SpectrumConfig, SpectrumProcessor
from oracle_kat import Kat
from signals import sine_wave


def test_normalization_bounds_runtime_values_without_enforcing_gui_ranges(backend):
    # :432-457
    p = SpectrumProcessor(backend, SpectrumConfig(sample_rate=float("nan"), fft_size=0, hop_size=0,
                                                  floor_db=float("inf")))
    c = p.config()
    assert c.fft_size == 1 and c.hop_size == 1 and c.floor_db == -100.0 and c.sample_rate == 48000.0
    for floor_db, expected in [(1.0, -100.0), (-280.0, -280.0)]:
        assert SpectrumProcessor(backend, SpectrumConfig(floor_db=floor_db)).config().floor_db == expected


def test_floor_change_reseeds_state_buffers_without_clearing_pending_audio(oracle):
    # :459-478 (reaches into private state: oracle only)
    p = SpectrumProcessor(oracle, SpectrumConfig())
    p.prepare()
    lib = oracle.lib
    pend = np.array([0.25, -0.25], np.float32)
    lib.omxo_spectrum_extend_pending(p._h, 0, pend.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(2))
    cfg = p.config()
    cfg.floor_db = -96.0
    p.update_config(cfg)
    buf = np.zeros(16, np.float32)
    lib.omxo_spectrum_pending.restype = C.c_uint64
    assert lib.omxo_spectrum_pending(p._h, 0, buf.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(16)) == 2
    snap = capi.CSpectrumSnapshot()
    lib.omxo_spectrum_peek_snapshot(p._h, C.byref(snap))
    s = SpectrumProcessor._snapshot(snap)
    bins = cfg.fft_size // 2 + 1
    for out in s.traces[0]:
        assert len(out) == bins and np.all(out == np.float32(-96.0))
    lib.omxo_spectrum_levels.restype = C.c_uint64
    lv = np.ones(bins, np.float32)
    assert lib.omxo_spectrum_levels(p._h, 0, 1, lv.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(bins)) == bins
    assert np.all(lv == 0.0)
    assert lib.omxo_spectrum_levels(p._h, 0, 0, lv.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(bins)) == 0
    assert lib.omxo_spectrum_levels(p._h, 1, 1, lv.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(bins)) == 0


def test_configured_sources_are_projected_before_fft(oracle):
    # :480-493
    p = SpectrumProcessor(oracle, SpectrumConfig(fft_size=8, source=capi.CH_LEFT, secondary_source=capi.CH_SIDE))
    p.process_block(AudioBlock([1.0, 0.0, 0.0, 1.0], 2, 48000.0))
    lib = oracle.lib
    lib.omxo_spectrum_pending.restype = C.c_uint64
    buf = np.zeros(8, np.float32)
    n = lib.omxo_spectrum_pending(p._h, 0, buf.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(8))
    assert buf[:n].tolist() == [1.0, 0.0]
    n = lib.omxo_spectrum_pending(p._h, 1, buf.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(8))
    assert buf[:n].tolist() == [0.5, -0.5]


def test_secondary_source_can_drive_processing_without_primary(backend):
    # :495-509
    p = SpectrumProcessor(backend, SpectrumConfig(fft_size=8, hop_size=8, source=capi.CH_NONE,
                                                  secondary_source=capi.CH_LEFT))
    assert p.process_block(AudioBlock(np.zeros(8, np.float32), 1, 48000.0)) is not None


def test_fft_size_update_resizes_scratch_before_processing(backend):
    # :511-536
    p = SpectrumProcessor(backend, SpectrumConfig(fft_size=128, hop_size=128))
    p.prepare()
    c = p.config()
    c.fft_size = 256
    c.hop_size = 256
    p.update_config(c)
    c = p.config()
    bins = c.fft_size // 2 + 1
    snap = p.process_block(AudioBlock(np.zeros(c.fft_size, np.float32), 1, c.sample_rate))
    assert snap is not None
    assert (len(snap.traces[0][0]), len(snap.traces[0][1])) == (bins, bins)
    assert len(snap.frequency_bins) == bins


def test_peak_hold_decays_for_each_audio_hop_in_large_batch(backend):
    # :538-563
    p = SpectrumProcessor(backend, SpectrumConfig(sample_rate=8.0, fft_size=8, hop_size=8,
                                                  window=capi.WINDOW_RECTANGULAR,
                                                  averaging_mode=capi.AVG_PEAK_HOLD, averaging_param=24.0,
                                                  floor_db=-100.0))
    samples = np.concatenate([sine_wave(1.0, 8.0, 8, 1.0), np.zeros(8, np.float32)])
    snap = p.process_block(AudioBlock(samples, 1, 8.0))
    assert snap is not None
    held_db = snap.traces[0][1][1]
    assert -24.1 < held_db < -23.9, f"held peak should decay once per hop, got {held_db} dB"


def test_changing_averaging_mode_clears_stale_state(oracle):
    # :565-581
    p = SpectrumProcessor(oracle, SpectrumConfig(averaging_mode=capi.AVG_PEAK_HOLD, averaging_param=12.0))
    p.prepare()
    lib = oracle.lib
    lib.omxo_spectrum_fill_smoothed(p._h, 0, C.c_float(1.0))
    c = p.config()
    c.averaging_mode = capi.AVG_EXPONENTIAL
    c.averaging_param = 0.5
    p.update_config(c)
    bins = c.fft_size // 2 + 1
    lv = np.ones(bins, np.float32)
    lib.omxo_spectrum_levels.restype = C.c_uint64
    assert lib.omxo_spectrum_levels(p._h, 0, 0, lv.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(bins)) == bins
    assert np.all(lv == 0.0)


def test_hops_larger_than_the_fft_are_block_partition_independent(backend):
    # :583-611
    c = SpectrumConfig(sample_rate=32.0, fft_size=8, hop_size=16, window=capi.WINDOW_RECTANGULAR,
                       source=capi.CH_LEFT)
    samples = np.sin((np.arange(29, dtype=np.float32) * np.float32(0.73)).astype(np.float32)).astype(np.float32)
    whole = SpectrumProcessor(backend, c).process_block(AudioBlock(samples, 1, 32.0))
    assert whole is not None
    p = SpectrumProcessor(backend, c)
    parted = None
    for k in range(0, 29, 8):
        s = p.process_block(AudioBlock(samples[k:k + 8], 1, 32.0))
        parted = s if s is not None else parted
    assert parted is not None
    for w in range(2):
        assert np.array_equal(whole.traces[0][w], parted.traces[0][w])


def test_averaged_power_is_zeroed_below_the_visible_floor(oracle):
    # :613-627
    kat = Kat(oracle)
    sf = kat.smoothing_state_floor([0.0], -100.0)
    r = kat.level_update(sf, kat.db_to_power(-101.0), 0.0, capi.AVG_EXPONENTIAL, 0.95, 0.0, 1.0, -100.0)
    assert r["smoothed"] == 0.0


def test_smoothing_retains_power_visible_after_weighting(oracle):
    # :629-651
    kat = Kat(oracle)
    for mode, param in [(capi.AVG_EXPONENTIAL, 0.95), (capi.AVG_PEAK_HOLD, 12.0)]:
        sf = kat.smoothing_state_floor([1.2], -100.0)
        r = kat.level_update(sf, 0.0, kat.db_to_power(-100.5), mode, param, 1.2, 1.0, -100.0)
        assert r["raw"] == -100.0
        assert -99.4 < r["weighted"] < -99.2


def test_a_weight_matches_iec_reference_points(backend):
    # :653-678
    for freq, expected in [(1.0, -148.6), (5.0, -93.1), (31.5, -39.4), (63.0, -26.2), (100.0, -19.1),
                           (200.0, -10.9), (500.0, -3.2), (1000.0, 0.0), (2000.0, 1.2), (4000.0, 1.0),
                           (8000.0, -1.1), (16000.0, -6.6)]:
        assert abs(backend.a_weight(freq) - expected) <= 0.15
    assert backend.a_weight(0.0) == float("-inf")


def test_snapshot_reflects_latest_hop_with_a_weighting(backend):
    # :215-253, :391-401: weighted = max(db + A[i], floor), raw = max(db, floor); bin 0 weighted = floor
    c = SpectrumConfig(fft_size=1024, hop_size=256, floor_db=-100.0)
    x = sine_wave(1000.0, 48000.0, 2048, 0.5)
    snap = SpectrumProcessor(backend, c).process_block(AudioBlock(np.stack([x, x], 1).reshape(-1), 2, 48000.0))
    assert snap is not None and len(snap.frequency_bins) == 513
    raw, weighted = snap.traces[0][1], snap.traces[0][0]
    k = int(np.argmax(raw))
    assert abs(snap.frequency_bins[k] - 1000.0) < 48000.0 / 1024
    assert abs(raw[k] - 20 * np.log10(0.5)) < 1.0
    assert weighted[0] == np.float32(-100.0)
    vis = raw > -99.0
    aw = np.array([backend.a_weight(float(f)) for f in snap.frequency_bins[vis]], np.float32)
    assert np.allclose(weighted[vis], np.maximum(raw[vis] + aw, -100.0), atol=1e-4)
    assert np.all(snap.traces[1][0] == np.float32(-100.0))  # inactive secondary trace stays at the floor
