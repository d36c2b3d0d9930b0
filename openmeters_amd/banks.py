"""Batched "bank" handles: S independent processors advanced in lock-step on one GPU (include/omx.h
`omx_<visual>_bank_*`).  Inputs and outputs are device pointers; nothing here touches sample data."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import capi
from .capi import Api, CSpectrogramBankUpdate, CSpectrogramConfig, SpectrogramConfig

_u8x8 = C.c_uint8 * 8


class SpectrogramBank:
    """S lock-step SpectrogramProcessors (reference src/visuals/spectrogram/processor.rs:170-544)."""

    def __init__(self, api: Api, config: SpectrogramConfig, n_streams: int):
        self.api = api
        self.n_streams = n_streams
        self._h = C.c_void_p()
        c = config.to_c()
        api.check(api.fn("spectrogram_bank_create", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)])(
            C.byref(c), n_streams, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self.api.fn("spectrogram_bank_destroy", None, [C.c_void_p])(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update_config(self, config: SpectrogramConfig):
        c = config.to_c()
        self.api.check(self.api.fn("spectrogram_bank_update_config", C.c_int, [C.c_void_p, C.c_void_p])(self._h, C.byref(c)))

    def reset_audio(self):
        self.api.check(self.api.fn("spectrogram_bank_reset_audio", C.c_int, [C.c_void_p])(self._h))

    def set_option(self, option: int, value: int):
        self.api.check(self.api.fn("spectrogram_bank_set_option", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64])(
            self._h, option, value))

    def _process(self, ptr, on_device, frames, channels, sample_rate, positions, stream) -> Optional[CSpectrogramBankUpdate]:
        out = CSpectrogramBankUpdate()
        f = self.api.fn("spectrogram_bank_process", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        rc = self.api.check(f(self._h, C.c_void_p(ptr), int(on_device), frames, channels, sample_rate,
                              _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out if rc == capi.PRODUCED else None

    def process_device(self, device_ptr: int, frames: int, channels: int, sample_rate: float,
                       positions: Sequence[int], stream: int = 0):
        """pcm = device pointer to f32 [n_streams][frames][channels]; work is enqueued on `stream`."""
        return self._process(device_ptr, True, frames, channels, sample_rate, positions, stream)

    def process_ragged(self, device_ptr: int, frames_capacity: int, frames: Sequence[int], channels: int, sample_rate: float,
                       positions: Sequence[int], reset_mask: Optional[Sequence[int]] = None, stream: int = 0):
        """Streams advance independently: stream s receives frames[s] (<= frames_capacity) new frames, after reset_audio() when
        reset_mask[s]; pcm = device f32 [n_streams][frames_capacity][channels].  Returns the CSpectrogramRaggedUpdate."""
        out = capi.CSpectrogramRaggedUpdate()
        fr = _per_stream(frames, self.n_streams, np.uint32, "frames")
        mask = _per_stream(reset_mask, self.n_streams, np.uint8, "reset_mask") if reset_mask is not None else None
        f = self.api.fn("spectrogram_bank_process_ragged", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), frames_capacity, fr.ctypes.data, mask.ctypes.data if mask is not None else None,
                         channels, sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def process_host(self, pcm: np.ndarray, channels: int, sample_rate: float, positions: Optional[Sequence[int]] = None):
        """pcm = host f32 [n_streams][frames][channels] (copied to the device first)."""
        pcm = np.ascontiguousarray(pcm, np.float32).reshape(self.n_streams, -1, channels)
        positions = positions if positions is not None else capi.positions_fallback(channels)
        return self._process(pcm.ctypes.data, False, pcm.shape[1], channels, sample_rate, positions, 0)

    def fetch_column(self, stream_index: int, column: int, kind: int, stride: int) -> np.ndarray:
        n = C.c_uint64()
        if kind == capi.COLUMN_REASSIGNED:
            buf = np.zeros((stride, 3), np.float32)
        else:
            buf = np.zeros((stride,), np.uint16)
        self.api.check(self.api.fn("spectrogram_bank_fetch_column", C.c_int,
                                   [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)])(
            self._h, stream_index, column, buf.ctypes.data, stride, C.byref(n)))
        return buf[:n.value]

    def kernel_time(self):
        ms, n = C.c_double(), C.c_uint64()
        self.api.check(self.api.fn("spectrogram_bank_kernel_time", C.c_int,
                                   [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)])(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class SpectrumBank:
    """S lock-step SpectrumProcessors (reference src/visuals/spectrum/processor.rs:72-323).
    emit_all_hops=True materialises every hop's traces (what a caller feeding one-hop blocks would have
    seen); False keeps the reference's snapshot semantics (latest hop only)."""

    def __init__(self, api: Api, config: capi.SpectrumConfig, n_streams: int, emit_all_hops: bool = False):
        self.api = api
        self.n_streams = n_streams
        self._h = C.c_void_p()
        c = config.to_c()
        api.check(api.fn("spectrum_bank_create", C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)])(
            C.byref(c), n_streams, int(emit_all_hops), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self.api.fn("spectrum_bank_destroy", None, [C.c_void_p])(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset_audio(self):
        self.api.check(self.api.fn("spectrum_bank_reset_audio", C.c_int, [C.c_void_p])(self._h))

    def set_option(self, option: int, value: int):
        self.api.check(self.api.fn("spectrum_bank_set_option", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64])(
            self._h, option, value))

    def _process(self, ptr, on_device, frames, channels, sample_rate, positions, stream):
        out = capi.CSpectrumBankUpdate()
        f = self.api.fn("spectrum_bank_process", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        rc = self.api.check(f(self._h, C.c_void_p(ptr), int(on_device), frames, channels, sample_rate,
                              _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out if rc == capi.PRODUCED else None

    def process_device(self, device_ptr, frames, channels, sample_rate, positions, stream=0):
        return self._process(device_ptr, True, frames, channels, sample_rate, positions, stream)

    def process_ragged(self, device_ptr: int, frames_capacity: int, frames: Sequence[int], channels: int, sample_rate: float,
                       positions: Sequence[int], reset_mask: Optional[Sequence[int]] = None, stream: int = 0):
        """Streams advance independently: stream s receives frames[s] (<= frames_capacity) new frames, after reset_audio() when
        reset_mask[s]; pcm = device f32 [n_streams][frames_capacity][channels].  Returns the CSpectrumRaggedUpdate."""
        out = capi.CSpectrumRaggedUpdate()
        fr = _per_stream(frames, self.n_streams, np.uint32, "frames")
        mask = _per_stream(reset_mask, self.n_streams, np.uint8, "reset_mask") if reset_mask is not None else None
        f = self.api.fn("spectrum_bank_process_ragged", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), frames_capacity, fr.ctypes.data, mask.ctypes.data if mask is not None else None,
                         channels, sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def process_host(self, pcm, channels, sample_rate, positions=None):
        pcm = np.ascontiguousarray(pcm, np.float32).reshape(self.n_streams, -1, channels)
        positions = positions if positions is not None else capi.positions_fallback(channels)
        return self._process(pcm.ctypes.data, False, pcm.shape[1], channels, sample_rate, positions, 0)

    def fetch(self, stream_index: int, hop: int, bins: int) -> np.ndarray:
        """-> float32 [2 traces][2 = (A-weighted, raw)][bins]"""
        buf = np.zeros((2, 2, bins), np.float32)
        self.api.check(self.api.fn("spectrum_bank_fetch", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p])(
            self._h, stream_index, hop, buf.ctypes.data))
        return buf


def _per_stream(values, n_streams, dtype, what):
    """A per-stream argument of a ragged call as a contiguous array of exactly n_streams entries (the C side reads that many)."""
    arr = np.ascontiguousarray(values, dtype).reshape(-1)
    if arr.size != n_streams:
        raise ValueError(f"{what}: {arr.size} entries for a bank of {n_streams} streams")
    return arr


class _BlockBank:
    """Shared plumbing of the block-structured banks (loudness / stereometer / oscilloscope): one call =
    `n_blocks` consecutive blocks of `block_frames` frames for every stream."""
    _family = ""

    def __init__(self, api: Api, cconfig, n_streams: int, *extra):
        self.api = api
        self.n_streams = n_streams
        self._h = C.c_void_p()
        argtypes = [C.c_void_p, C.c_uint32] + [C.c_uint32] * len(extra) + [C.POINTER(C.c_void_p)]
        api.check(api.fn(f"{self._family}_bank_create", C.c_int, argtypes)(C.byref(cconfig), n_streams, *extra, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self.api.fn(f"{self._family}_bank_destroy", None, [C.c_void_p])(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset_audio(self):
        self.api.check(self.api.fn(f"{self._family}_bank_reset_audio", C.c_int, [C.c_void_p])(self._h))

    def update_config(self, config):
        c = config.to_c()
        self.api.check(self.api.fn(f"{self._family}_bank_update_config", C.c_int, [C.c_void_p, C.c_void_p])(self._h, C.byref(c)))

    def _pcm(self, pcm, channels):
        pcm = np.ascontiguousarray(pcm, np.float32).reshape(self.n_streams, -1, channels)
        return pcm


class LoudnessBank(_BlockBank):
    """reference src/visuals/loudness/processor.rs:218-312, S streams in lock-step."""
    _family = "loudness"

    def __init__(self, api: Api, config: capi.LoudnessConfig, n_streams: int, channels: int = 2):
        super().__init__(api, config.to_c(), n_streams, channels)

    def set_option(self, option, value):
        self.api.check(self.api.fn("loudness_bank_set_option", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64])(self._h, option, value))

    def last_form(self) -> int:
        """1 = the last call ran the sequential kernels, 2 = the chunk-parallel ones (omx_debug_loudness_bank_last_form)."""
        return self.api.fn("debug_loudness_bank_last_form", C.c_int, [C.c_void_p])(self._h)

    def _process(self, ptr, on_device, block_frames, n_blocks, channels, sample_rate, positions, stream):
        out = C.c_void_p()
        f = self.api.fn("loudness_bank_process", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32,
                                                           C.c_float, _u8x8, C.c_void_p, C.POINTER(C.c_void_p)])
        rc = self.api.check(f(self._h, C.c_void_p(ptr), int(on_device), block_frames, n_blocks, channels, sample_rate,
                              _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out.value if rc == capi.PRODUCED else None

    def process_device(self, device_ptr, block_frames, n_blocks, channels, sample_rate, positions, stream=0):
        return self._process(device_ptr, True, block_frames, n_blocks, channels, sample_rate, positions, stream)

    def process_host(self, pcm, block_frames, channels, sample_rate, positions=None):
        pcm = self._pcm(pcm, channels)
        positions = positions if positions is not None else capi.positions_fallback(channels)
        assert pcm.shape[1] % block_frames == 0
        return self._process(pcm.ctypes.data, False, block_frames, pcm.shape[1] // block_frames, channels, sample_rate, positions, 0)

    def process_ragged(self, device_ptr: int, block_frames: int, max_blocks: int, n_blocks: Sequence[int], channels: int, sample_rate: float,
                       positions: Sequence[int], reset_mask: Optional[Sequence[int]] = None, stream: int = 0):
        """Streams advance independently: stream s runs n_blocks[s] (<= max_blocks) blocks of block_frames frames, after reset_audio()
        when reset_mask[s]; pcm = device f32 [n_streams][block_frames * max_blocks][channels].  Returns the CLoudnessRaggedUpdate
        (snapshots of stream s, block k: fetch(s, k) for k < n_blocks[s])."""
        out = capi.CLoudnessRaggedUpdate()
        nb = _per_stream(n_blocks, self.n_streams, np.uint32, "n_blocks")
        mask = _per_stream(reset_mask, self.n_streams, np.uint8, "reset_mask") if reset_mask is not None else None
        f = self.api.fn("loudness_bank_process_ragged", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, _u8x8, C.c_void_p,
                         C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), block_frames, max_blocks, nb.ctypes.data, mask.ctypes.data if mask is not None else None,
                         channels, sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def process_chunks(self, device_ptr: int, frames_capacity: int, frames: Sequence[int], channels: int, sample_rate: float,
                       positions: Sequence[int], reset_mask: Optional[Sequence[int]] = None, stream: int = 0):
        """VisualManager::ingest_samples per capture (registry.rs:396-418): stream s delivers ONE block of frames[s] (<= frames_capacity)
        frames — a batcher chunk, whole; pcm = device f32 [n_streams][frames_capacity][channels].  fetch(s, 0) where frames[s] != 0."""
        out = capi.CLoudnessRaggedUpdate()
        fr = _per_stream(frames, self.n_streams, np.uint32, "frames")
        mask = _per_stream(reset_mask, self.n_streams, np.uint8, "reset_mask") if reset_mask is not None else None
        f = self.api.fn("loudness_bank_process_chunks", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), frames_capacity, fr.ctypes.data, mask.ctypes.data if mask is not None else None,
                         channels, sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def fetch(self, stream_index, block) -> capi.LoudnessSnapshot:
        out = capi.CLoudnessSnapshot()
        self.api.check(self.api.fn("loudness_bank_fetch", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p])(
            self._h, stream_index, block, C.byref(out)))
        return capi.LoudnessSnapshot(out.short_term_loudness, out.momentary_loudness, np.array(out.rms_fast_db[:], np.float32),
                                     np.array(out.rms_slow_db[:], np.float32), np.array(out.true_peak_db[:], np.float32),
                                     out.channel_count, list(out.positions))

    def kernel_time(self):
        ms, n = C.c_double(), C.c_uint64()
        self.api.check(self.api.fn("loudness_bank_kernel_time", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)])(
            self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class CStereometerRaggedUpdate(C.Structure):
    _fields_ = [("n_streams", C.c_uint64), ("max_blocks", C.c_uint64), ("target", C.c_uint64), ("d_n_blocks", C.c_void_p),
                ("d_correlations", C.c_void_p), ("d_produced", C.c_void_p), ("d_points", C.c_void_p), ("d_band_valid", C.c_void_p)]


class StereometerBank(_BlockBank):
    """reference src/visuals/stereometer/processor.rs:64-208, S streams in lock-step."""
    _family = "stereometer"

    def __init__(self, api: Api, config: capi.StereometerConfig, n_streams: int):
        super().__init__(api, config.to_c(), n_streams)

    def _process(self, ptr, on_device, block_frames, n_blocks, channels, sample_rate, positions, stream):
        out = capi.CStereometerBankUpdate()
        f = self.api.fn("stereometer_bank_process", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32,
                                                              C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(ptr), int(on_device), block_frames, n_blocks, channels, sample_rate,
                         _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def process_device(self, device_ptr, block_frames, n_blocks, channels, sample_rate, positions, stream=0):
        return self._process(device_ptr, True, block_frames, n_blocks, channels, sample_rate, positions, stream)

    def process_host(self, pcm, block_frames, channels, sample_rate, positions=None):
        pcm = self._pcm(pcm, channels)
        positions = positions if positions is not None else capi.positions_fallback(channels)
        assert pcm.shape[1] % block_frames == 0
        return self._process(pcm.ctypes.data, False, block_frames, pcm.shape[1] // block_frames, channels, sample_rate, positions, 0)

    def set_option(self, option, value):
        self.api.check(self.api.fn("stereometer_bank_set_option", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64])(self._h, option, value))

    def last_form(self) -> int:
        """1 = the last call ran the sequential kernels, 2 = the chunk-parallel ones (omx_debug_stereometer_bank_last_form)."""
        return self.api.fn("debug_stereometer_bank_last_form", C.c_int, [C.c_void_p])(self._h)

    def process_ragged(self, device_ptr: int, block_frames: int, max_blocks: int, n_blocks: Sequence[int], channels: int, sample_rate: float,
                       positions: Sequence[int], reset_mask: Optional[Sequence[int]] = None, stream: int = 0):
        """Streams advance independently: stream s runs n_blocks[s] (<= max_blocks) blocks of block_frames frames, after reset_audio()
        when reset_mask[s]; pcm = device f32 [n_streams][block_frames * max_blocks][channels].  fetch(s, k) for k < n_blocks[s];
        fetch_points(s, band) returns the points of the stream's last block when it produced a snapshot (else none)."""
        out = CStereometerRaggedUpdate()
        nb = _per_stream(n_blocks, self.n_streams, np.uint32, "n_blocks")
        mask = _per_stream(reset_mask, self.n_streams, np.uint8, "reset_mask") if reset_mask is not None else None
        f = self.api.fn("stereometer_bank_process_ragged", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, _u8x8, C.c_void_p,
                         C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), block_frames, max_blocks, nb.ctypes.data, mask.ctypes.data if mask is not None else None,
                         channels, sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def process_chunks(self, device_ptr: int, frames_capacity: int, frames: Sequence[int], channels: int, sample_rate: float,
                       positions: Sequence[int], reset_mask: Optional[Sequence[int]] = None, stream: int = 0):
        """VisualManager::ingest_samples per capture (registry.rs:396-418): stream s delivers ONE block of frames[s] (<= frames_capacity)
        frames — a batcher chunk, whole; pcm = device f32 [n_streams][frames_capacity][channels].  fetch(s, 0) where frames[s] != 0."""
        out = CStereometerRaggedUpdate()
        fr = _per_stream(frames, self.n_streams, np.uint32, "frames")
        mask = _per_stream(reset_mask, self.n_streams, np.uint8, "reset_mask") if reset_mask is not None else None
        f = self.api.fn("stereometer_bank_process_chunks", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), frames_capacity, fr.ctypes.data, mask.ctypes.data if mask is not None else None,
                         channels, sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def fetch_points(self, stream_index, band, capacity=4096):
        buf = np.zeros((capacity, 2), np.float32)
        n = C.c_uint64()
        self.api.check(self.api.fn("stereometer_bank_fetch_points", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                                                                              C.POINTER(C.c_uint64)])(
            self._h, stream_index, band, buf.ctypes.data, capacity, C.byref(n)))
        return buf[:n.value]

    def fetch(self, stream_index, block):
        corr = (C.c_float * 4)()
        produced = C.c_uint32()
        self.api.check(self.api.fn("stereometer_bank_fetch", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_float * 4,
                                                                       C.POINTER(C.c_uint32)])(
            self._h, stream_index, block, corr, C.byref(produced)))
        return np.array(corr[:], np.float32), bool(produced.value)


class COscilloscopeBlockHeader(C.Structure):
    _fields_ = [("produced", C.c_uint32), ("channels", C.c_uint32), ("slots", C.c_uint32 * 2),
                ("samples_per_channel", C.c_uint32), ("locked", C.c_uint32), ("period", C.c_float), ("capture_start", C.c_uint32),
                ("capture_frac", C.c_float), ("_pad", C.c_uint32)]


class COscilloscopeBankUpdate(C.Structure):
    _fields_ = [("n_streams", C.c_uint64), ("n_blocks", C.c_uint64), ("epoch", C.c_uint64), ("sample_stride", C.c_uint64),
                ("d_headers", C.c_void_p), ("d_samples", C.c_void_p)]


class COscilloscopeRaggedUpdate(C.Structure):
    _fields_ = [("n_streams", C.c_uint64), ("max_blocks", C.c_uint64), ("d_n_blocks", C.c_void_p), ("d_epochs", C.c_void_p),
                ("d_headers", C.c_void_p), ("d_samples", C.c_void_p)]


class OscilloscopeBank(_BlockBank):
    """reference src/visuals/oscilloscope/processor.rs:570-759, S streams in lock-step."""
    _family = "oscilloscope"

    def __init__(self, api: Api, config: capi.OscilloscopeConfig, n_streams: int):
        super().__init__(api, config.to_c(), n_streams)

    def _process(self, ptr, on_device, block_frames, n_blocks, channels, sample_rate, positions, stream):
        out = COscilloscopeBankUpdate()
        f = self.api.fn("oscilloscope_bank_process", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32,
                                                               C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(ptr), int(on_device), block_frames, n_blocks, channels, sample_rate,
                         _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def process_device(self, device_ptr, block_frames, n_blocks, channels, sample_rate, positions, stream=0):
        return self._process(device_ptr, True, block_frames, n_blocks, channels, sample_rate, positions, stream)

    def process_host(self, pcm, block_frames, channels, sample_rate, positions=None):
        pcm = self._pcm(pcm, channels)
        positions = positions if positions is not None else capi.positions_fallback(channels)
        assert pcm.shape[1] % block_frames == 0
        return self._process(pcm.ctypes.data, False, block_frames, pcm.shape[1] // block_frames, channels, sample_rate, positions, 0)

    def process_ragged(self, device_ptr: int, block_frames: int, max_blocks: int, n_blocks: Sequence[int], channels: int, sample_rate: float,
                       positions: Sequence[int], reset_mask: Optional[Sequence[int]] = None, stream: int = 0):
        """Streams advance independently: stream s runs n_blocks[s] (<= max_blocks) blocks of block_frames frames, after reset_audio()
        when reset_mask[s]; pcm = device f32 [n_streams][block_frames * max_blocks][channels].  Headers of stream s, block k:
        fetch(s, k) for k < n_blocks[s]; fetch(s, n_blocks[s] - 1, with_samples=True) adds the stream's newest snapshot."""
        out = COscilloscopeRaggedUpdate()
        nb = _per_stream(n_blocks, self.n_streams, np.uint32, "n_blocks")
        mask = _per_stream(reset_mask, self.n_streams, np.uint8, "reset_mask") if reset_mask is not None else None
        f = self.api.fn("oscilloscope_bank_process_ragged", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, _u8x8, C.c_void_p,
                         C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), block_frames, max_blocks, nb.ctypes.data, mask.ctypes.data if mask is not None else None,
                         channels, sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def process_chunks(self, device_ptr: int, frames_capacity: int, frames: Sequence[int], channels: int, sample_rate: float,
                       positions: Sequence[int], reset_mask: Optional[Sequence[int]] = None, stream: int = 0):
        """VisualManager::ingest_samples per capture (registry.rs:396-418): stream s delivers ONE block of frames[s] (<= frames_capacity)
        frames — a batcher chunk, whole; pcm = device f32 [n_streams][frames_capacity][channels].  fetch(s, 0) where frames[s] != 0."""
        out = COscilloscopeRaggedUpdate()
        fr = _per_stream(frames, self.n_streams, np.uint32, "frames")
        mask = _per_stream(reset_mask, self.n_streams, np.uint8, "reset_mask") if reset_mask is not None else None
        f = self.api.fn("oscilloscope_bank_process_chunks", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), frames_capacity, fr.ctypes.data, mask.ctypes.data if mask is not None else None,
                         channels, sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def resume_block(self, stream_index: int) -> int:
        """test hook: the block at which the capped wide trigger pass handed the stream over in the last call (omx_debug_oscilloscope_bank_resume_block)"""
        return int(self.api.fn("debug_oscilloscope_bank_resume_block", C.c_longlong, [C.c_void_p, C.c_uint32])(self._h, stream_index))

    def fetch(self, stream_index, block, with_samples=False):
        hdr = COscilloscopeBlockHeader()
        buf = np.zeros((2, 4096), np.float32) if with_samples else None
        self.api.check(self.api.fn("oscilloscope_bank_fetch", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p])(
            self._h, stream_index, block, C.byref(hdr), buf.ctypes.data if with_samples else None))
        return hdr, buf


class WaveformBank(_BlockBank):
    """reference src/visuals/waveform/processor.rs:135-353, S streams in lock-step; one block per call (the column phase is
    advanced on the host)."""
    _family = "waveform"

    def __init__(self, api: Api, config: capi.WaveformConfig, n_streams: int):
        super().__init__(api, config.to_c(), n_streams)

    def _process(self, ptr, on_device, frames, channels, sample_rate, positions, stream):
        out = capi.CWaveformBankUpdate()
        f = self.api.fn("waveform_bank_process", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_float, _u8x8,
                                                           C.c_void_p, C.c_void_p])
        rc = self.api.check(f(self._h, C.c_void_p(ptr), int(on_device), frames, channels, sample_rate, _u8x8(*positions),
                              C.c_void_p(stream or 0), C.byref(out)))
        return out if rc == capi.PRODUCED else None

    def process_device(self, device_ptr, frames, channels, sample_rate, positions, stream=0):
        return self._process(device_ptr, True, frames, channels, sample_rate, positions, stream)

    def process_host(self, pcm, channels, sample_rate, positions=None):
        pcm = self._pcm(pcm, channels)
        positions = positions if positions is not None else capi.positions_fallback(channels)
        return self._process(pcm.ctypes.data, False, pcm.shape[1], channels, sample_rate, positions, 0)

    def process_ragged(self, device_ptr: int, frames_capacity: int, frames: Sequence[int], channels: int, sample_rate: float,
                       positions: Sequence[int], reset_mask: Optional[Sequence[int]] = None, stream: int = 0):
        """Streams advance independently: stream s receives frames[s] (<= frames_capacity) new frames, after reset_audio() when
        reset_mask[s]; pcm = device f32 [n_streams][frames_capacity][channels].  Returns the CWaveformRaggedUpdate; columns of stream
        s: fetch(s, update.max_columns)[0][:n_columns[s]]."""
        out = capi.CWaveformRaggedUpdate()
        fr = _per_stream(frames, self.n_streams, np.uint32, "frames")
        mask = _per_stream(reset_mask, self.n_streams, np.uint8, "reset_mask") if reset_mask is not None else None
        f = self.api.fn("waveform_bank_process_ragged", C.c_int,
                        [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), frames_capacity, fr.ctypes.data, mask.ctypes.data if mask is not None else None,
                         channels, sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def set_option(self, option: int, value: int):
        """OMX_OPT_KERNEL_FORM: 0 by call shape, 1 sequential kernels, 2 chunk-parallel form where it applies"""
        self.api.check(self.api.fn("waveform_bank_set_option", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64])(self._h, option, value))

    def last_form(self) -> int:
        """1 = the last lock-step call ran the sequential kernels alone, 2 = the chunk-parallel form (omx_debug_waveform_bank_last_form)."""
        return self.api.fn("debug_waveform_bank_last_form", C.c_int, [C.c_void_p])(self._h)

    def fetch(self, stream_index, n_columns, with_preview=False):
        cols = np.zeros((max(n_columns, 1), 4, 11), np.float32)
        prev = np.zeros((4, 11), np.float32) if with_preview else None
        self.api.check(self.api.fn("waveform_bank_fetch", C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p])(
            self._h, stream_index, cols.ctypes.data, prev.ctypes.data if with_preview else None))
        return cols[:n_columns], prev
