"""Error behaviour of the C ABI on a machine WITH a device: the reference has no error path, so the only
negative statuses are invalid arguments and the shapes the HIP path does not implement; a failing call must leave the handle
usable."""
import ctypes as C

import numpy as np
import pytest

from openmeters_amd import banks, capi
from openmeters_amd.capi import AudioBlock, SpectrogramConfig, SpectrogramProcessor, SpectrumConfig, SpectrumProcessor

pytestmark = pytest.mark.gpu


def test_oversized_transforms_are_reported_not_computed(omx):
    """What is left of the unsupported set: padded transforms beyond 2^24 points, and reassigned windows beyond 65536 samples whose
    length is not a power of two (every other length the reference accepts is computed: tests/test_gpu_parity.py)."""
    p = SpectrogramProcessor(omx, SpectrogramConfig(fft_size=1 << 20, hop_size=500, zero_padding_factor=32))
    x = np.zeros(8000 * 2, np.float32)
    with pytest.raises(capi.OmxError) as e:
        p.process_block(AudioBlock(x, 2, 48000.0))
    assert e.value.status == capi.ERR_UNSUPPORTED and "unsupported" in str(e.value).lower()
    # the handle survives: a supported configuration works afterwards
    p.update_config(SpectrogramConfig(fft_size=1024, hop_size=256, history_length=64))
    t = np.arange(4096) / 48000.0
    pcm = np.stack([np.sin(2 * np.pi * 1000 * t), np.sin(2 * np.pi * 1000 * t)], 1).astype(np.float32)
    up = p.process_block(AudioBlock(pcm.reshape(-1), 2, 48000.0))
    assert up is not None and len(up.new_columns) == (4096 - 2048) // 256 + 1
    s = SpectrumProcessor(omx, SpectrumConfig(fft_size=(1 << 24) + 2, hop_size=250))
    with pytest.raises(capi.OmxError) as e:
        s.process_block(AudioBlock(pcm.reshape(-1), 2, 48000.0))
    assert e.value.status == capi.ERR_UNSUPPORTED


def test_rejected_update_config_leaves_a_prepared_handle_untouched(omx, oracle):
    """A prepared handle that is handed a shape the HIP path refuses must stay on its old configuration: same get_config,
    same tables, and the next block still yields the OLD configuration's columns (checked against the oracle that never saw
    the rejected update)."""
    from parity import check_reassigned_update
    cfg = SpectrogramConfig(fft_size=1024, hop_size=256, history_length=64, use_reassignment=True)
    t = np.arange(2048 + 256 * 11) / 48000.0
    left = (0.4 * np.sin(2 * np.pi * (300.0 + 3000.0 * t) * t)).astype(np.float32)
    pcm = np.stack([left, 0.7 * left], 1).reshape(-1)
    half = 2 * (2048 + 256 * 4)
    got, ref = SpectrogramProcessor(omx, cfg), SpectrogramProcessor(oracle, cfg)
    a0, b0 = got.process_block(AudioBlock(pcm[:half], 2, 48000.0)), ref.process_block(AudioBlock(pcm[:half], 2, 48000.0))
    assert len(a0.new_columns) == len(b0.new_columns) == 5
    for bad in (SpectrogramConfig(fft_size=100003, hop_size=256), SpectrogramConfig(fft_size=1 << 20, hop_size=64, zero_padding_factor=32)):
        with pytest.raises(capi.OmxError) as e:
            got.update_config(bad)
        assert e.value.status == capi.ERR_UNSUPPORTED
        kept = got.config()
        assert (kept.fft_size, kept.hop_size, kept.zero_padding_factor) == (1024, 256, 1)
    a1, b1 = got.process_block(AudioBlock(pcm[half:], 2, 48000.0)), ref.process_block(AudioBlock(pcm[half:], 2, 48000.0))
    assert a1 is not None and len(a1.new_columns) == len(b1.new_columns) == 7 and a1.reset == b1.reset
    check_reassigned_update(a1, b1, 48000.0, 256)
    # spectrum: same contract
    sg, sr = SpectrumProcessor(omx, SpectrumConfig(fft_size=1024, hop_size=256)), SpectrumProcessor(oracle, SpectrumConfig(fft_size=1024, hop_size=256))
    sg.process_block(AudioBlock(pcm[:half], 2, 48000.0)); sr.process_block(AudioBlock(pcm[:half], 2, 48000.0))
    with pytest.raises(capi.OmxError) as e:
        sg.update_config(SpectrumConfig(fft_size=(1 << 24) + 2, hop_size=250))
    assert e.value.status == capi.ERR_UNSUPPORTED and sg.config().fft_size == 1024
    x, y = sg.process_block(AudioBlock(pcm[half:], 2, 48000.0)), sr.process_block(AudioBlock(pcm[half:], 2, 48000.0))
    assert x is not None and y is not None
    np.testing.assert_allclose(10 ** (np.asarray(x.traces[0][1]) / 10), 10 ** (np.asarray(y.traces[0][1]) / 10), atol=1e-5 * (10 ** (np.max(y.traces[0][1]) / 10)))


def test_invalid_arguments_are_rejected(omx):
    f = omx.fn("spectrogram_bank_create", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)])
    h = C.c_void_p()
    cfg = SpectrogramConfig().to_c()
    assert f(None, 4, C.byref(h)) == capi.ERR_INVALID
    assert f(C.byref(cfg), 0, C.byref(h)) == capi.ERR_INVALID
    assert omx.fn("spectrogram_process_block", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p])(None, None, None) == capi.ERR_INVALID
    g = omx.fn("spectrum_peaks", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float,
                                           C.c_void_p, C.c_void_p])
    a = np.zeros(8, np.float32)
    out = np.zeros(1, capi.SPECTRUM_PEAK_DTYPE)
    assert g(a.ctypes.data, a.ctypes.data, 0, 8, 1, 4, 0.0, 1.0, None, out.ctypes.data) == capi.ERR_INVALID  # row_stride < n_bins
    bank = banks.SpectrogramBank(omx, SpectrogramConfig(fft_size=4096, hop_size=256), 2)
    with pytest.raises(capi.OmxError):
        bank.fetch_column(5, 0, capi.COLUMN_REASSIGNED, 2049)      # stream index out of range


def test_empty_and_short_blocks_return_none(omx):
    p = SpectrogramProcessor(omx, SpectrogramConfig(fft_size=4096, hop_size=256))
    assert p.process_block(AudioBlock(np.zeros(0, np.float32), 2, 48000.0)) is None   # AudioBlock::is_empty (dsp.rs:259-261)
    assert p.process_block(AudioBlock(np.zeros(1, np.float32), 2, 48000.0)) is None   # fewer samples than channels
    assert p.process_block(AudioBlock(np.zeros(2 * 100, np.float32), 2, 48000.0)) is None  # not yet a full window
