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
