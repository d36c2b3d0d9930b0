"""ctypes mirror of the capture group (include/omx.h: omx_capture_group_*) — VisualManager::ingest_samples of the reference
(src/visuals/registry.rs:396-418): one block of every capture goes to every enabled visual.  The fan-out, the shared projection of
the block for the Spectrogram / Spectrum banks, the meter banks on side streams and the per-stream summary rows all live in
libomx_hip.so; nothing here computes anything.  `FullPipeline` is cfg5 of BASELINE.json (reassigned STFT + BS.1770 loudness + phase
correlation on one GPU's shard of streams, summary rows gathered over RCCL once per step: sharding.gather_stats)."""
from __future__ import annotations

import ctypes as C

import numpy as np
from typing import Optional, Sequence

from . import banks, capi
from .capi import (CLoudnessConfig, COscilloscopeConfig, CSpectrogramBankUpdate, CSpectrogramConfig, CSpectrumBankUpdate, CSpectrumConfig,
                   CStereometerBankUpdate, CStereometerConfig, CWaveformBankUpdate, CWaveformConfig)
from .sharding import STATS_COLUMNS, gather_stats, shard_streams

_u8x8 = C.c_uint8 * 8


class CCaptureGroupConfig(C.Structure):
    _fields_ = [("n_streams", C.c_uint32), ("visuals", C.c_uint32), ("block_frames", C.c_uint32), ("spectrum_emit_all_hops", C.c_uint32),
                ("spectrogram", CSpectrogramConfig), ("spectrum", CSpectrumConfig), ("loudness", CLoudnessConfig),
                ("stereometer", CStereometerConfig), ("oscilloscope", COscilloscopeConfig), ("waveform", CWaveformConfig)]


class CCaptureGroupUpdate(C.Structure):
    _fields_ = [("produced", C.c_uint32), ("ingest_launches", C.c_uint32), ("n_blocks", C.c_uint64), ("block_frames", C.c_uint64),
                ("spectrogram", CSpectrogramBankUpdate), ("spectrum", CSpectrumBankUpdate), ("d_loudness", C.c_void_p),
                ("stereometer", CStereometerBankUpdate), ("oscilloscope", banks.COscilloscopeBankUpdate), ("waveform", CWaveformBankUpdate),
                ("d_stats_rows", C.c_void_p)]


class CCaptureGroupRaggedUpdate(C.Structure):
    _fields_ = [("produced", C.c_uint32), ("ingest_launches", C.c_uint32), ("block_frames", C.c_uint64), ("max_blocks", C.c_uint64),
                ("spectrogram", capi.CSpectrogramRaggedUpdate), ("spectrum", capi.CSpectrumRaggedUpdate), ("loudness", capi.CLoudnessRaggedUpdate),
                ("stereometer", banks.CStereometerRaggedUpdate), ("oscilloscope", banks.COscilloscopeRaggedUpdate),
                ("waveform", capi.CWaveformRaggedUpdate), ("d_stats_rows", C.c_void_p)]


_CONFIG_OF = {capi.VISUAL_SPECTROGRAM: "spectrogram", capi.VISUAL_SPECTRUM: "spectrum", capi.VISUAL_LOUDNESS: "loudness",
              capi.VISUAL_STEREOMETER: "stereometer", capi.VISUAL_OSCILLOSCOPE: "oscilloscope", capi.VISUAL_WAVEFORM: "waveform"}


class CaptureGroup:
    """omx_capture_group: `n_streams` captures, one bank per enabled visual — VisualManager's ingest_samples, set_enabled,
    apply_module_settings -> update_config and reset_audio (registry.rs:266-277, :343-365, :396-418)."""

    def __init__(self, api: capi.Api, n_streams: int, *, spectrogram: Optional[capi.SpectrogramConfig] = None,
                 spectrum: Optional[capi.SpectrumConfig] = None, loudness: Optional[capi.LoudnessConfig] = None,
                 stereometer: Optional[capi.StereometerConfig] = None, oscilloscope: Optional[capi.OscilloscopeConfig] = None,
                 waveform: Optional[capi.WaveformConfig] = None, block_frames: int = 0, spectrum_emit_all_hops: bool = False,
                 stats: bool = False):
        self.api, self.n_streams = api, n_streams
        cfg = CCaptureGroupConfig()
        api.fn("capture_group_config_default", None, [C.c_void_p])(C.byref(cfg))
        cfg.n_streams = n_streams
        cfg.block_frames = block_frames
        cfg.spectrum_emit_all_hops = int(spectrum_emit_all_hops)
        for bit, name, value in ((capi.VISUAL_SPECTROGRAM, "spectrogram", spectrogram), (capi.VISUAL_SPECTRUM, "spectrum", spectrum),
                                 (capi.VISUAL_LOUDNESS, "loudness", loudness), (capi.VISUAL_STEREOMETER, "stereometer", stereometer),
                                 (capi.VISUAL_OSCILLOSCOPE, "oscilloscope", oscilloscope), (capi.VISUAL_WAVEFORM, "waveform", waveform)):
            if value is not None:
                cfg.visuals |= bit
                setattr(cfg, name, value.to_c())
        h = C.c_void_p()
        api.check(api.fn("capture_group_create", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)])(C.byref(cfg), C.byref(h)))
        self._h = h
        self.visuals = int(cfg.visuals)
        if stats:
            self.set_option(capi.OPT_GROUP_STATS, 1)

    def close(self):
        if getattr(self, "_h", None):
            self.api.fn("capture_group_destroy", None, [C.c_void_p])(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option: int, value: int):
        self.api.check(self.api.fn("capture_group_set_option", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64])(self._h, option, value))

    def reset_audio(self):
        self.api.check(self.api.fn("capture_group_reset_audio", C.c_int, [C.c_void_p])(self._h))

    def ingest(self, device_ptr: int, frames: int, channels: int, sample_rate: float, positions: Sequence[int], stream: int = 0) -> CCaptureGroupUpdate:
        """One block [n_streams][frames][channels] (device memory) to every enabled visual.  Returns the update (device pointers)."""
        out = CCaptureGroupUpdate()
        f = self.api.fn("capture_group_ingest", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, _u8x8, C.c_void_p, C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), frames, channels, sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def set_enabled(self, visual: int, on: bool):
        self.api.check(self.api.fn("capture_group_set_enabled", C.c_int, [C.c_void_p, C.c_uint32, C.c_int])(self._h, visual, int(on)))

    def enabled(self) -> int:
        return int(self.api.fn("capture_group_enabled", C.c_int, [C.c_void_p])(self._h))

    def update_config(self, visual: int, config, stream: int = 0):
        """`config`: the capi.<Visual>Config of that visual (update_config of its processor between two ingest calls)"""
        c = config.to_c()
        self.api.check(self.api.fn("capture_group_update_config", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p])(
            self._h, visual, C.byref(c), C.c_void_p(stream or 0)))

    def note_format(self, generation: int) -> bool:
        rc = self.api.fn("capture_group_note_format", C.c_int, [C.c_void_p, C.c_uint64])(self._h, generation)
        self.api.check(rc)
        return rc == 1

    def ingest_ragged(self, device_ptr: int, frames_capacity: int, frames: Sequence[int], channels: int, sample_rate: float,
                      positions: Sequence[int], reset_mask: Optional[Sequence[int]] = None, stream: int = 0) -> CCaptureGroupRaggedUpdate:
        """capture s delivers its first frames[s] frames of [n_streams][frames_capacity][channels] (device memory)"""
        out = CCaptureGroupRaggedUpdate()
        n = self.n_streams
        fr = np.ascontiguousarray(frames, dtype=np.uint32)   # (numpy arrays pass through without a per-element conversion)
        mk = np.ascontiguousarray(reset_mask, dtype=np.uint8) if reset_mask is not None else None
        assert fr.shape == (n,) and (mk is None or mk.shape == (n,))
        f = self.api.fn("capture_group_ingest_ragged", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float,
                                                                 _u8x8, C.c_void_p, C.c_void_p])
        self.api.check(f(self._h, C.c_void_p(device_ptr), frames_capacity, fr.ctypes.data, mk.ctypes.data if mk is not None else None, channels,
                         sample_rate, _u8x8(*positions), C.c_void_p(stream or 0), C.byref(out)))
        return out

    def kernel_time(self):
        ms, n = C.c_double(), C.c_uint64()
        self.api.check(self.api.fn("capture_group_kernel_time", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)])(
            self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class CAudioFormat(C.Structure):   # omx_audio_format (reference src/dsp.rs:79-85)
    _fields_ = [("generation", C.c_uint64), ("sample_rate", C.c_float), ("channels", C.c_uint32), ("positions", C.c_uint8 * 8)]


class BatcherBank:
    """omx_batcher_bank_*: one DspBatcher per capture (reference src/meter.rs:27-80) with the samples resident on the device.  `push`
    takes one packet per capture (device memory, host lengths) and returns the rounds it produced: (device pointer, chunk capacity in
    frames, host frame counts) per round — the arguments of CaptureGroup.ingest_ragged."""

    def __init__(self, api: capi.Api, n_captures: int, max_packet_frames: int):
        self.api, self.n_captures = api, n_captures
        self._h = C.c_void_p()
        api.check(api.fn("batcher_bank_create", C.c_int, [C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p)])(n_captures, max_packet_frames, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self.api.fn("batcher_bank_destroy", None, [C.c_void_p])(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, device_ptr: int, packet_stride: int, packet_frames: Sequence[int], channels: int, sample_rate: float,
             positions: Sequence[int], generation: int = 0, clear_mask: Optional[Sequence[int]] = None, stream: int = 0):
        fr = np.ascontiguousarray(packet_frames, dtype=np.uint32)
        mk = np.ascontiguousarray(clear_mask, dtype=np.uint8) if clear_mask is not None else None
        assert fr.shape == (self.n_captures,) and (mk is None or mk.shape == (self.n_captures,))
        fmt = CAudioFormat(generation, sample_rate, channels, (C.c_uint8 * 8)(*positions))
        n = C.c_uint32()
        f = self.api.fn("batcher_bank_push", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.POINTER(C.c_uint32)])
        self.api.check(f(self._h, C.c_void_p(device_ptr), packet_stride, fr.ctypes.data, mk.ctypes.data if mk is not None else None, C.byref(fmt),
                         C.c_void_p(stream or 0), C.byref(n)))
        return self._rounds(n.value)

    def _rounds(self, n: int):
        rounds = []
        g = self.api.fn("batcher_bank_round", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)])
        for r in range(n):
            ptr, cap, frames = C.c_void_p(), C.c_uint64(), C.c_void_p()
            self.api.check(g(self._h, r, C.byref(ptr), C.byref(cap), C.byref(frames)))
            counts = np.ctypeslib.as_array(C.cast(frames, C.POINTER(C.c_uint32)), shape=(self.n_captures,)).copy()
            rounds.append((int(ptr.value or 0), int(cap.value), counts))
        return rounds

    def push_silence(self, silence_frames: Sequence[int], channels: int, sample_rate: float, positions: Sequence[int], generation: int = 0,
                     stream: int = 0):
        """ingest_silence (meter.rs:145-166) per capture; returns (rounds, reset flags): a capture whose silence exceeds 2 s is reset"""
        fr = np.ascontiguousarray(silence_frames, dtype=np.uint64)
        assert fr.shape == (self.n_captures,)
        fmt = CAudioFormat(generation, sample_rate, channels, (C.c_uint8 * 8)(*positions))
        n = C.c_uint32()
        reset = np.zeros(self.n_captures, np.uint8)
        f = self.api.fn("batcher_bank_push_silence", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p])
        self.api.check(f(self._h, fr.ctypes.data, C.byref(fmt), C.c_void_p(stream or 0), C.byref(n), reset.ctypes.data))
        return self._rounds(n.value), reset

    def pending(self, capture: int, stream: int = 0):
        f = self.api.fn("batcher_bank_pending", C.c_uint64, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p])
        n = int(f(self._h, capture, None, 0, C.c_void_p(stream or 0)))
        buf = np.zeros(max(n, 1), np.float32)
        f(self._h, capture, buf.ctypes.data, n, C.c_void_p(stream or 0))
        return buf[:n]


class _DeviceView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}


def stats_rows_tensor(torch, device, update, n_streams: int):
    """The update's summary rows as a torch view [n_streams, 12] f32 of the library's device buffer (valid until the next ingest)."""
    if not update.d_stats_rows:
        return None
    return torch.as_tensor(_DeviceView(update.d_stats_rows, (n_streams, len(STATS_COLUMNS)), "<f4"), device=device)


class FullPipeline:
    """cfg5: one GPU's shard — `n_streams` streams, reassigned 4096 / 256 STFT + loudness + stereometer (band analysis), summary rows."""

    def __init__(self, api: capi.Api, n_streams: int, channels: int = 2, sample_rate: float = 48000.0):
        self.api, self.n_streams, self.channels, self.sample_rate = api, n_streams, channels, sample_rate
        self.positions = capi.positions_fallback(channels)
        self.group = CaptureGroup(
            api, n_streams,
            spectrogram=capi.SpectrogramConfig(sample_rate=sample_rate, fft_size=4096, hop_size=256, history_length=8192, use_reassignment=True),
            loudness=capi.LoudnessConfig(sample_rate=sample_rate),
            stereometer=capi.StereometerConfig(sample_rate=sample_rate, analyze_bands=True, correlation_window=0.05, segment_duration=0.02,
                                               target_sample_count=2000),
            block_frames=256, stats=True)

    def step(self, device_ptr: int, frames: int, stream: int = 0) -> CCaptureGroupUpdate:
        """`frames` (a multiple of 256) new frames per stream through the whole group; the update carries d_stats_rows."""
        assert frames % 256 == 0
        return self.group.ingest(device_ptr, frames, self.channels, self.sample_rate, self.positions, stream)

    def step_with_stats(self, torch, device, device_ptr: int, frames: int):
        """(update, summary rows [n_streams, 12] as a torch view) on torch's current stream."""
        up = self.step(device_ptr, frames, torch.cuda.current_stream().cuda_stream)
        return up, stats_rows_tensor(torch, device, up, self.n_streams)


__all__ = ["CaptureGroup", "CCaptureGroupUpdate", "CCaptureGroupRaggedUpdate", "FullPipeline", "stats_rows_tensor", "gather_stats", "shard_streams", "STATS_COLUMNS"]
