"""ctypes view of the C-ABI declared in ``include/omx.h``.

The classes here are the host-side mirror of the reference's per-visual processor API
(``new / config / update_config / reset_audio / prepare / process_block``; reference
``src/visuals/<visual>/processor.rs``), written once over an ``Api`` object that is bound
to a shared library + symbol prefix.  The product binds ``libomx_hip.so`` with prefix
``omx_`` (see ``openmeters_amd/__init__.py``); tests bind the CPU oracle with prefix
``omxo_`` through the very same classes, so parity tests read like the reference's own
unit tests with only the backend swapped.

Nothing in this module computes DSP: it marshals pointers and sizes.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

MAX_CHANNELS = 8

# status codes (omx_status)
PRODUCED, NONE = 1, 0
ERR_BACKEND, ERR_UNSUPPORTED, ERR_INVALID, ERR_NO_DEVICE = -1, -2, -3, -4

# ChannelPosition (reference src/dsp.rs:8-22)
POS_FL, POS_FR, POS_FC, POS_LFE, POS_RL, POS_RR, POS_SL, POS_SR, POS_MONO, POS_UNKNOWN = range(10)
POS_AUX0 = 16
SURROUND = [POS_FL, POS_FR, POS_FC, POS_LFE, POS_RL, POS_RR, POS_SL, POS_SR]

# WindowKind (reference src/util/audio/window.rs:9-18)
WINDOW_RECTANGULAR, WINDOW_HANN, WINDOW_HAMMING, WINDOW_BLACKMAN, WINDOW_BLACKMAN_HARRIS = range(5)
# Channel (reference src/util/audio/channel.rs:4-10)
CH_LEFT, CH_RIGHT, CH_MID, CH_SIDE, CH_NONE = range(5)
# AveragingMode (reference src/visuals/spectrum/processor.rs:64-70)
AVG_NONE, AVG_EXPONENTIAL, AVG_PEAK_HOLD = range(3)
# TriggerMode (reference src/visuals/oscilloscope/processor.rs:21-25)
TRIGGER_ZERO_CROSSING, TRIGGER_STABLE = range(2)
COLUMN_REASSIGNED, COLUMN_CLASSIC = range(2)
OPT_KERNEL_TIMING, OPT_FORCE_GENERIC, OPT_KERNEL_FORM, OPT_LOUDNESS_REBASE_FRAMES = 1, 2, 3, 4
OPT_GROUP_STATS, OPT_GROUP_SHARED_INGEST = 16, 17
VISUAL_SPECTROGRAM, VISUAL_SPECTRUM, VISUAL_LOUDNESS, VISUAL_STEREOMETER, VISUAL_OSCILLOSCOPE, VISUAL_WAVEFORM = 1, 2, 4, 8, 16, 32
STATS_COLUMN_COUNT = 12

DEFAULT_SAMPLE_RATE = 48_000.0

_f32p = C.POINTER(C.c_float)
_u8x8 = C.c_uint8 * 8


class OmxError(RuntimeError):
    """A negative omx_status from the backend (the reference has no such case)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"omx status {status}: {message}")
        self.status = status


class CBlock(C.Structure):
    _fields_ = [("samples", _f32p), ("n_samples", C.c_uint64), ("channels", C.c_uint32),
                ("sample_rate", C.c_float), ("positions", _u8x8)]


class CSpectrogramConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_float), ("window", C.c_uint32), ("fft_size", C.c_uint64),
                ("hop_size", C.c_uint64), ("history_length", C.c_uint64),
                ("zero_padding_factor", C.c_uint64), ("use_reassignment", C.c_uint32), ("_pad", C.c_uint32)]


class CSpectrogramPoint(C.Structure):
    _fields_ = [("time_offset", C.c_float), ("freq_hz", C.c_float), ("power", C.c_float)]


class CSpectrogramUpdate(C.Structure):
    _fields_ = [("fft_size", C.c_uint64), ("hop_size", C.c_uint64), ("history_length", C.c_uint64),
                ("n_columns", C.c_uint64), ("column_offsets", C.POINTER(C.c_uint64)),
                ("points", C.POINTER(CSpectrogramPoint)), ("codes", C.POINTER(C.c_uint16)),
                ("sample_rate", C.c_float), ("reassigned_power_scale", C.c_float),
                ("reset", C.c_uint32), ("kind", C.c_uint32)]


class CSpectrogramBankUpdate(C.Structure):
    _fields_ = [("fft_size", C.c_uint64), ("hop_size", C.c_uint64), ("history_length", C.c_uint64),
                ("n_streams", C.c_uint64), ("n_columns", C.c_uint64), ("column_stride", C.c_uint64),
                ("d_counts", C.c_void_p), ("d_points", C.c_void_p), ("d_codes", C.c_void_p),
                ("sample_rate", C.c_float), ("reassigned_power_scale", C.c_float),
                ("reset", C.c_uint32), ("kind", C.c_uint32)]


class CSpectrogramRaggedUpdate(C.Structure):
    _fields_ = [("fft_size", C.c_uint64), ("hop_size", C.c_uint64), ("history_length", C.c_uint64),
                ("n_streams", C.c_uint64), ("max_columns", C.c_uint64), ("column_stride", C.c_uint64),
                ("d_n_columns", C.c_void_p), ("d_reset", C.c_void_p), ("d_counts", C.c_void_p), ("d_points", C.c_void_p),
                ("d_codes", C.c_void_p), ("sample_rate", C.c_float), ("reassigned_power_scale", C.c_float),
                ("kind", C.c_uint32), ("_pad", C.c_uint32)]


class CSpectrumConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_float), ("window", C.c_uint32), ("fft_size", C.c_uint64),
                ("hop_size", C.c_uint64), ("averaging_mode", C.c_uint32), ("averaging_param", C.c_float),
                ("source", C.c_uint32), ("secondary_source", C.c_uint32), ("floor_db", C.c_float),
                ("_pad", C.c_uint32)]


class CSpectrumSnapshot(C.Structure):
    _fields_ = [("bins", C.c_uint64), ("frequency_bins", _f32p), ("traces", (_f32p * 2) * 2)]


class CSpectrumBankUpdate(C.Structure):
    _fields_ = [("bins", C.c_uint64), ("n_streams", C.c_uint64), ("n_hops", C.c_uint64),
                ("n_hops_out", C.c_uint64), ("d_traces", C.c_void_p), ("d_frequency_bins", C.c_void_p)]


class CWaveformRaggedUpdate(C.Structure):
    _fields_ = [("n_streams", C.c_uint64), ("max_columns", C.c_uint64), ("d_n_columns", C.c_void_p), ("d_columns", C.c_void_p),
                ("d_preview", C.c_void_p), ("d_preview_progress", C.c_void_p), ("d_reset", C.c_void_p)]


class CLoudnessRaggedUpdate(C.Structure):
    _fields_ = [("n_streams", C.c_uint64), ("max_blocks", C.c_uint64), ("d_n_blocks", C.c_void_p), ("d_snapshots", C.c_void_p),
                ("d_reset", C.c_void_p), ("d_block_frames", C.c_void_p)]


class CSpectrumRaggedUpdate(C.Structure):
    _fields_ = [("bins", C.c_uint64), ("n_streams", C.c_uint64), ("max_hops", C.c_uint64), ("n_hops_out", C.c_uint64),
                ("d_n_hops", C.c_void_p), ("d_traces", C.c_void_p), ("d_frequency_bins", C.c_void_p)]


class CLoudnessConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_float), ("floor_db", C.c_float)]


class CLoudnessSnapshot(C.Structure):
    _fields_ = [("short_term_loudness", C.c_float), ("momentary_loudness", C.c_float),
                ("rms_fast_db", C.c_float * 8), ("rms_slow_db", C.c_float * 8), ("true_peak_db", C.c_float * 8),
                ("channel_count", C.c_uint32), ("positions", _u8x8), ("_pad", C.c_uint32)]


class CStereometerConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_float), ("segment_duration", C.c_float), ("target_sample_count", C.c_uint64),
                ("correlation_window", C.c_float), ("analyze_bands", C.c_uint32), ("emit_band_points", C.c_uint32),
                ("_pad", C.c_uint32)]


class CStereometerSnapshot(C.Structure):
    _fields_ = [("points", _f32p * 4), ("n_points", C.c_uint64 * 4), ("correlations", C.c_float * 4)]


class CStereometerBankUpdate(C.Structure):
    _fields_ = [("n_streams", C.c_uint64), ("n_blocks", C.c_uint64), ("target", C.c_uint64),
                ("d_correlations", C.c_void_p), ("d_points", C.c_void_p), ("d_produced", C.c_void_p)]


class COscilloscopeConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_float), ("segment_duration", C.c_float), ("trigger_mode", C.c_uint32),
                ("trigger_source", C.c_uint32), ("num_cycles", C.c_uint64), ("channel_1", C.c_uint32),
                ("channel_2", C.c_uint32)]


class COscilloscopeSnapshot(C.Structure):
    _fields_ = [("epoch", C.c_uint64), ("channels", C.c_uint64), ("slots", C.c_uint64 * 2),
                ("samples_per_channel", C.c_uint64), ("n_samples", C.c_uint64), ("samples", _f32p)]


class CWaveformConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_float), ("scroll_speed", C.c_float), ("max_columns", C.c_uint64),
                ("analyze_bands", C.c_uint32), ("track_history", C.c_uint32)]


class CWaveColumn(C.Structure):
    _fields_ = [("min", C.c_float), ("max", C.c_float), ("color_bands", C.c_float * 3), ("rms_db", (C.c_float * 3) * 2)]


class CWaveformUpdate(C.Structure):
    _fields_ = [("n_columns", C.c_uint64), ("columns", C.POINTER(CWaveColumn)), ("reset", C.c_uint32),
                ("preview_some", C.c_uint32), ("preview_progress", C.c_float), ("_pad", C.c_uint32),
                ("preview", CWaveColumn * 4)]


class CWaveformBankUpdate(C.Structure):
    _fields_ = [("n_streams", C.c_uint64), ("n_columns", C.c_uint64), ("d_columns", C.c_void_p), ("d_preview", C.c_void_p),
                ("reset", C.c_uint32), ("preview_some", C.c_uint32), ("preview_progress", C.c_float), ("_pad", C.c_uint32)]


# ----------------------------------------------------------------------------- config dataclasses
@dataclass
class SpectrogramConfig:
    """reference src/visuals/spectrogram/processor.rs:45-56"""
    sample_rate: float = DEFAULT_SAMPLE_RATE
    fft_size: int = 2048
    hop_size: int = 64
    window: int = WINDOW_HANN
    history_length: int = 0
    use_reassignment: bool = True
    zero_padding_factor: int = 1

    def to_c(self) -> CSpectrogramConfig:
        return CSpectrogramConfig(self.sample_rate, self.window, self.fft_size, self.hop_size, self.history_length,
                                  self.zero_padding_factor, int(self.use_reassignment), 0)

    @staticmethod
    def from_c(c: CSpectrogramConfig) -> "SpectrogramConfig":
        return SpectrogramConfig(c.sample_rate, c.fft_size, c.hop_size, c.window, c.history_length,
                                 bool(c.use_reassignment), c.zero_padding_factor)


@dataclass
class SpectrumConfig:
    """reference src/visuals/spectrum/processor.rs:39-51"""
    sample_rate: float = DEFAULT_SAMPLE_RATE
    fft_size: int = 16_384
    hop_size: int = 16_384 // 16
    window: int = WINDOW_HANN
    averaging_mode: int = AVG_NONE
    averaging_param: float = 0.0
    source: int = CH_MID
    secondary_source: int = CH_NONE
    floor_db: float = -100.0

    def to_c(self) -> CSpectrumConfig:
        return CSpectrumConfig(self.sample_rate, self.window, self.fft_size, self.hop_size, self.averaging_mode,
                               self.averaging_param, self.source, self.secondary_source, self.floor_db, 0)

    @staticmethod
    def from_c(c: CSpectrumConfig) -> "SpectrumConfig":
        return SpectrumConfig(c.sample_rate, c.fft_size, c.hop_size, c.window, c.averaging_mode, c.averaging_param,
                              c.source, c.secondary_source, c.floor_db)


@dataclass
class LoudnessConfig:
    """reference src/visuals/loudness/processor.rs:210-216"""
    sample_rate: float = DEFAULT_SAMPLE_RATE
    floor_db: float = -99.9

    def to_c(self) -> CLoudnessConfig:
        return CLoudnessConfig(self.sample_rate, self.floor_db)


@dataclass
class StereometerConfig:
    """reference src/visuals/stereometer/processor.rs:11-21"""
    sample_rate: float = DEFAULT_SAMPLE_RATE
    segment_duration: float = 0.02
    target_sample_count: int = 2_000
    correlation_window: float = 0.05
    analyze_bands: bool = False
    emit_band_points: bool = False

    def to_c(self) -> CStereometerConfig:
        return CStereometerConfig(self.sample_rate, self.segment_duration, self.target_sample_count,
                                  self.correlation_window, int(self.analyze_bands), int(self.emit_band_points), 0)

    @staticmethod
    def from_c(c: CStereometerConfig) -> "StereometerConfig":
        return StereometerConfig(c.sample_rate, c.segment_duration, c.target_sample_count, c.correlation_window,
                                 bool(c.analyze_bands), bool(c.emit_band_points))


@dataclass
class OscilloscopeConfig:
    """reference src/visuals/oscilloscope/processor.rs:33-43 (TriggerMode flattened to mode + num_cycles)"""
    sample_rate: float = DEFAULT_SAMPLE_RATE
    segment_duration: float = 0.02
    trigger_mode: int = TRIGGER_STABLE
    num_cycles: int = 2
    trigger_source: int = CH_MID
    channel_1: int = CH_MID
    channel_2: int = CH_NONE

    def to_c(self) -> COscilloscopeConfig:
        return COscilloscopeConfig(self.sample_rate, self.segment_duration, self.trigger_mode, self.trigger_source,
                                   self.num_cycles, self.channel_1, self.channel_2)

    @staticmethod
    def from_c(c: COscilloscopeConfig) -> "OscilloscopeConfig":
        return OscilloscopeConfig(c.sample_rate, c.segment_duration, c.trigger_mode, c.num_cycles, c.trigger_source,
                                  c.channel_1, c.channel_2)


@dataclass
class WaveformConfig:
    """reference src/visuals/waveform/processor.rs:31-40"""
    sample_rate: float = DEFAULT_SAMPLE_RATE
    scroll_speed: float = 300.0
    max_columns: int = 8192
    analyze_bands: bool = True
    track_history: bool = False

    def to_c(self) -> CWaveformConfig:
        return CWaveformConfig(self.sample_rate, self.scroll_speed, self.max_columns, int(self.analyze_bands), int(self.track_history))

    @staticmethod
    def from_c(c: CWaveformConfig) -> "WaveformConfig":
        return WaveformConfig(c.sample_rate, c.scroll_speed, c.max_columns, bool(c.analyze_bands), bool(c.track_history))


# ----------------------------------------------------------------------------- snapshots
@dataclass
class WaveformUpdate:
    """reference src/visuals/waveform/processor.rs:66-76; columns: float32 [n, 4 channels, 11] with the WaveColumn
    fields flattened as (min, max, color_bands[3], rms_db[2][3])"""
    reset: bool
    columns: np.ndarray
    preview_progress: float
    preview: Optional[np.ndarray]  # [4, 11] or None


@dataclass
class SpectrogramUpdate:
    """reference src/visuals/spectrogram/processor.rs:160-168; columns are numpy arrays:
    reassigned -> float32 [n,3] (time_offset, freq_hz, power); classic -> uint16 [bins]"""
    fft_size: int
    hop_size: int
    sample_rate: float
    history_length: int
    reset: bool
    reassigned_power_scale: float
    kind: int
    new_columns: List[np.ndarray] = field(default_factory=list)


@dataclass
class SpectrumSnapshot:
    frequency_bins: np.ndarray
    traces: List[List[np.ndarray]]  # [trace][0 = weighted, 1 = raw]


@dataclass
class LoudnessSnapshot:
    short_term_loudness: float
    momentary_loudness: float
    rms_fast_db: np.ndarray
    rms_slow_db: np.ndarray
    true_peak_db: np.ndarray
    channel_count: int
    positions: List[int]


@dataclass
class StereometerSnapshot:
    points: List[np.ndarray]  # [band] float32 [n,2]
    correlations: np.ndarray


@dataclass
class OscilloscopeSnapshot:
    epoch: int
    channels: int
    slots: List[int]
    samples: np.ndarray
    samples_per_channel: int


def positions_fallback(channels: int) -> List[int]:
    """reference src/dsp.rs:36-47 (pure integer table; duplicated here so blocks can be built
    without a backend)"""
    channels = min(channels, MAX_CHANNELS)
    p = [POS_UNKNOWN] * MAX_CHANNELS
    p[:channels] = SURROUND[:channels]
    if channels == 1:
        p[0] = POS_MONO
    elif channels == 4:
        p[2:4] = [POS_RL, POS_RR]
    elif channels == 5:
        p[3:5] = [POS_RL, POS_RR]
    return p


class AudioBlock:
    """reference src/dsp.rs:108-115 — borrowed PCM + format for one call."""

    def __init__(self, samples, channels: int, sample_rate: float, positions: Optional[Sequence[int]] = None):
        self.samples = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
        self.channels = max(1, min(int(channels), MAX_CHANNELS))
        self.sample_rate = float(sample_rate)
        self.positions = list(positions) if positions is not None else positions_fallback(self.channels)

    def to_c(self) -> CBlock:
        return CBlock(self.samples.ctypes.data_as(_f32p), self.samples.size, self.channels,
                      self.sample_rate, _u8x8(*self.positions))


# ----------------------------------------------------------------------------- binding
class Api:
    """A shared library + symbol prefix exposing the omx.h entry points."""

    def __init__(self, path: str, prefix: str):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)

    def fn(self, name: str, restype=C.c_int, argtypes=None):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        if argtypes is not None:
            f.argtypes = argtypes
        return f

    def has(self, name: str) -> bool:
        return hasattr(self.lib, self.prefix + name)

    def check(self, status: int) -> int:
        if status < 0:
            msg = ""
            if self.has("last_error"):
                msg = self.fn("last_error", C.c_char_p, [])().decode("utf-8", "replace")
            raise OmxError(status, msg)
        return status

    # -- free functions
    def pack_classic_db(self, db: float) -> int:
        return int(self.fn("pack_classic_db", C.c_uint16, [C.c_float])(db))

    def history_columns(self, kind: int, points: int, requested: int) -> int:
        return int(self.fn("spectrogram_history_columns", C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint64])(
            kind, points, requested))

    def a_weight(self, freq_hz: float) -> float:
        return float(self.fn("a_weight", C.c_float, [C.c_float])(freq_hz))

    def k_weighting_coefficients(self, fs: float):
        b = (C.c_double * 5)()
        a = (C.c_double * 5)()
        self.fn("k_weighting_coefficients", None, [C.c_double, C.c_double * 5, C.c_double * 5])(fs, b, a)
        return np.array(b[:]), np.array(a[:])

    def positions_fallback(self, channels: int) -> List[int]:
        out = _u8x8()
        self.fn("positions_fallback", None, [C.c_uint32, _u8x8])(channels, out)
        return list(out)

    def positions_normalize(self, channels: int, positions: Sequence[int]) -> List[int]:
        out = _u8x8()
        self.fn("positions_normalize", None, [C.c_uint32, _u8x8, _u8x8])(channels, _u8x8(*positions), out)
        return list(out)


class _Handle:
    _family = ""

    def __init__(self, api: Api, cconfig):
        self.api = api
        self._h = C.c_void_p()
        api.check(api.fn(f"{self._family}_create", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)])(
            C.byref(cconfig), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.api.fn(f"{self._family}_destroy", None, [C.c_void_p])(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name: str, *args, argtypes=None) -> int:
        f = self.api.fn(f"{self._family}_{name}", C.c_int, [C.c_void_p] + list(argtypes or []))
        return self.api.check(f(self._h, *args))

    def reset_audio(self):
        self._call("reset_audio")


class SpectrogramProcessor(_Handle):
    """reference src/visuals/spectrogram/processor.rs:170-544"""
    _family = "spectrogram"

    def __init__(self, api: Api, config: SpectrogramConfig):
        super().__init__(api, config.to_c())

    def config(self) -> SpectrogramConfig:
        c = CSpectrogramConfig()
        self._call("get_config", C.byref(c), argtypes=[C.c_void_p])
        return SpectrogramConfig.from_c(c)

    def debug_capture(self, enable: bool = True):
        """ORACLE handles only (test hook): record the samples every column of the following updates is computed from."""
        self.api.fn("debug_spectrogram_capture", None, [C.c_void_p, C.c_int])(self._h, int(enable))

    def debug_captured(self, index: int) -> np.ndarray:
        """ORACLE handles only: the samples column `index` of the last update was computed from (empty when there is none)."""
        f = self.api.fn("debug_spectrogram_captured", C.c_uint64, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64])
        n = int(f(self._h, index, None, 0))
        buf = np.zeros(max(n, 1), np.float32)
        f(self._h, index, buf.ctypes.data, n)
        return buf[:n]

    def update_config(self, config: SpectrogramConfig):
        c = config.to_c()
        self._call("update_config", C.byref(c), argtypes=[C.c_void_p])

    def prepare(self):
        self._call("prepare")

    def process_block(self, block: AudioBlock) -> Optional[SpectrogramUpdate]:
        cb = block.to_c()
        out = CSpectrogramUpdate()
        if self._call("process_block", C.byref(cb), C.byref(out), argtypes=[C.c_void_p, C.c_void_p]) == NONE:
            return None
        n = out.n_columns
        offs = np.ctypeslib.as_array(out.column_offsets, shape=(n + 1,)).copy()
        cols: List[np.ndarray] = []
        total = int(offs[-1])
        if out.kind == COLUMN_REASSIGNED:
            flat = (np.ctypeslib.as_array(C.cast(out.points, _f32p), shape=(total * 3,)).copy().reshape(total, 3)
                    if total else np.zeros((0, 3), np.float32))
        else:
            flat = (np.ctypeslib.as_array(out.codes, shape=(total,)).copy() if total else np.zeros((0,), np.uint16))
        for c in range(n):
            cols.append(flat[int(offs[c]):int(offs[c + 1])])
        return SpectrogramUpdate(out.fft_size, out.hop_size, out.sample_rate, out.history_length, bool(out.reset),
                                 out.reassigned_power_scale, out.kind, cols)


class SpectrumProcessor(_Handle):
    """reference src/visuals/spectrum/processor.rs:72-323"""
    _family = "spectrum"

    def __init__(self, api: Api, config: SpectrumConfig):
        super().__init__(api, config.to_c())

    def config(self) -> SpectrumConfig:
        c = CSpectrumConfig()
        self._call("get_config", C.byref(c), argtypes=[C.c_void_p])
        return SpectrumConfig.from_c(c)

    def update_config(self, config: SpectrumConfig):
        c = config.to_c()
        self._call("update_config", C.byref(c), argtypes=[C.c_void_p])

    def prepare(self):
        self._call("prepare")

    @staticmethod
    def _snapshot(out: CSpectrumSnapshot) -> SpectrumSnapshot:
        bins = int(out.bins)
        grab = lambda p: np.ctypeslib.as_array(p, shape=(bins,)).copy() if bins and p else np.zeros((0,), np.float32)
        return SpectrumSnapshot(grab(out.frequency_bins),
                                [[grab(out.traces[t][w]) for w in range(2)] for t in range(2)])

    def process_block(self, block: AudioBlock) -> Optional[SpectrumSnapshot]:
        cb = block.to_c()
        out = CSpectrumSnapshot()
        if self._call("process_block", C.byref(cb), C.byref(out), argtypes=[C.c_void_p, C.c_void_p]) == NONE:
            return None
        return self._snapshot(out)


class LoudnessProcessor(_Handle):
    """reference src/visuals/loudness/processor.rs:218-312"""
    _family = "loudness"

    def __init__(self, api: Api, config: LoudnessConfig):
        super().__init__(api, config.to_c())

    def process_block(self, block: AudioBlock) -> Optional[LoudnessSnapshot]:
        cb = block.to_c()
        out = CLoudnessSnapshot()
        if self._call("process_block", C.byref(cb), C.byref(out), argtypes=[C.c_void_p, C.c_void_p]) == NONE:
            return None
        return LoudnessSnapshot(out.short_term_loudness, out.momentary_loudness,
                                np.array(out.rms_fast_db[:], np.float32), np.array(out.rms_slow_db[:], np.float32),
                                np.array(out.true_peak_db[:], np.float32), out.channel_count, list(out.positions))


class StereometerProcessor(_Handle):
    """reference src/visuals/stereometer/processor.rs:64-208"""
    _family = "stereometer"

    def __init__(self, api: Api, config: StereometerConfig):
        super().__init__(api, config.to_c())

    def config(self) -> StereometerConfig:
        c = CStereometerConfig()
        self._call("get_config", C.byref(c), argtypes=[C.c_void_p])
        return StereometerConfig.from_c(c)

    def update_config(self, config: StereometerConfig):
        c = config.to_c()
        self._call("update_config", C.byref(c), argtypes=[C.c_void_p])

    def process_block(self, block: AudioBlock) -> Optional[StereometerSnapshot]:
        cb = block.to_c()
        out = CStereometerSnapshot()
        if self._call("process_block", C.byref(cb), C.byref(out), argtypes=[C.c_void_p, C.c_void_p]) == NONE:
            return None
        pts = []
        for b in range(4):
            n = int(out.n_points[b])
            pts.append(np.ctypeslib.as_array(out.points[b], shape=(n * 2,)).copy().reshape(n, 2) if n
                       else np.zeros((0, 2), np.float32))
        return StereometerSnapshot(pts, np.array(out.correlations[:], np.float32))


class WaveformProcessor(_Handle):
    """reference src/visuals/waveform/processor.rs:135-353"""
    _family = "waveform"

    def __init__(self, api: Api, config: WaveformConfig):
        super().__init__(api, config.to_c())

    def config(self) -> WaveformConfig:
        c = CWaveformConfig()
        self._call("get_config", C.byref(c), argtypes=[C.c_void_p])
        return WaveformConfig.from_c(c)

    def update_config(self, config: WaveformConfig):
        c = config.to_c()
        self._call("update_config", C.byref(c), argtypes=[C.c_void_p])

    def prepare(self):
        self._call("prepare")

    def process_block(self, block: AudioBlock) -> Optional[WaveformUpdate]:
        cb = block.to_c()
        out = CWaveformUpdate()
        if self._call("process_block", C.byref(cb), C.byref(out), argtypes=[C.c_void_p, C.c_void_p]) == NONE:
            return None
        n = int(out.n_columns)
        cols = (np.ctypeslib.as_array(C.cast(out.columns, _f32p), shape=(n * 4 * 11,)).copy().reshape(n, 4, 11) if n
                else np.zeros((0, 4, 11), np.float32))
        prev = None
        if out.preview_some:
            prev = np.frombuffer(bytes(out.preview), dtype=np.float32).reshape(4, 11).copy()
        return WaveformUpdate(bool(out.reset), cols, out.preview_progress, prev)


class OscilloscopeProcessor(_Handle):
    """reference src/visuals/oscilloscope/processor.rs:570-759"""
    _family = "oscilloscope"

    def __init__(self, api: Api, config: OscilloscopeConfig):
        super().__init__(api, config.to_c())

    def config(self) -> OscilloscopeConfig:
        c = COscilloscopeConfig()
        self._call("get_config", C.byref(c), argtypes=[C.c_void_p])
        return OscilloscopeConfig.from_c(c)

    def update_config(self, config: OscilloscopeConfig):
        c = config.to_c()
        self._call("update_config", C.byref(c), argtypes=[C.c_void_p])

    def last_cycle_rate(self) -> Optional[float]:
        hz = C.c_float()
        f = self.api.fn("oscilloscope_last_cycle_rate", C.c_int, [C.c_void_p, C.POINTER(C.c_float)])
        return float(hz.value) if f(self._h, C.byref(hz)) == 1 else None

    def last_capture(self):
        """(start, frac_offset) of the Capture behind the newest snapshot (test-only view), or None"""
        start, frac = C.c_uint32(), C.c_float()
        f = self.api.fn("oscilloscope_last_capture", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float)])
        return (int(start.value), float(frac.value)) if f(self._h, C.byref(start), C.byref(frac)) == 1 else None

    def process_block(self, block: AudioBlock) -> Optional[OscilloscopeSnapshot]:
        cb = block.to_c()
        out = COscilloscopeSnapshot()
        if self._call("process_block", C.byref(cb), C.byref(out), argtypes=[C.c_void_p, C.c_void_p]) == NONE:
            return None
        n = int(out.n_samples)
        samples = np.ctypeslib.as_array(out.samples, shape=(n,)).copy() if n else np.zeros((0,), np.float32)
        return OscilloscopeSnapshot(out.epoch, out.channels, list(out.slots), samples, out.samples_per_channel)


# --------------------------------------------------------------------------- state-side summary reductions
METER_LUFS_SHORT_TERM, METER_LUFS_MOMENTARY, METER_RMS_FAST, METER_RMS_SLOW, METER_TRUE_PEAK = range(5)
SPECTRUM_PEAK_DTYPE = np.dtype([("found", "<u4"), ("bin", "<u4"), ("freq_hz", "<f4"), ("level_db", "<f4")])
PEAK_HOLD_DTYPE = np.dtype([("db", "<f4"), ("_pad", "<u4"), ("decay_from", "<f8")])
METER_ROW_DTYPE = np.dtype([("values", "<f4", (3,)), ("peaks", "<f4", (3,))])


def spectrum_peaks(api: Api, bins: np.ndarray, db: np.ndarray, min_f: float, max_f: float) -> np.ndarray:
    """peak_bin + interpolated_peak (reference src/visuals/spectrum/state.rs:320-356) for every row of `db` [rows, bins]."""
    bins = np.ascontiguousarray(bins, np.float32)
    db = np.ascontiguousarray(np.atleast_2d(db), np.float32)
    out = np.zeros(db.shape[0], SPECTRUM_PEAK_DTYPE)
    f = api.fn("spectrum_peaks", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float,
                                           C.c_void_p, C.c_void_p])
    api.check(f(bins.ctypes.data, db.ctypes.data, 0, bins.shape[0], db.shape[0], db.shape[1], min_f, max_f, None, out.ctypes.data))
    return out


def peak_holds_reset(api: Api, n: int, now: float) -> np.ndarray:
    """PeakHold::new(DB_RANGE.0, now) x n (reference src/visuals/loudness/state.rs:42-47)."""
    holds = np.zeros(n, PEAK_HOLD_DTYPE)
    api.check(api.fn("peak_holds_reset", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_double, C.c_void_p])(
        holds.ctypes.data, 0, n, now, None))
    return holds


def loudness_snapshots_to_c(snapshots: Sequence["LoudnessSnapshot"]):
    arr = (CLoudnessSnapshot * len(snapshots))()
    for c, s in zip(arr, snapshots):
        c.short_term_loudness, c.momentary_loudness = s.short_term_loudness, s.momentary_loudness
        for i in range(MAX_CHANNELS):
            c.rms_fast_db[i], c.rms_slow_db[i], c.true_peak_db[i] = s.rms_fast_db[i], s.rms_slow_db[i], s.true_peak_db[i]
            c.positions[i] = s.positions[i]
        c.channel_count = s.channel_count
    return arr


def loudness_meters(api: Api, snapshots: Sequence["LoudnessSnapshot"], n_streams: int, left_mode: int, right_mode: int,
                    t0: float, dt: float, holds: np.ndarray) -> np.ndarray:
    """visible_values + update_peak_holds (reference src/visuals/loudness/state.rs:178-217) for snapshots laid out
    [n_streams][n_blocks]; `holds` ([n_streams * 3] PEAK_HOLD_DTYPE) is updated in place; returns rows [n_streams, n_blocks]."""
    n_blocks = len(snapshots) // n_streams
    arr = loudness_snapshots_to_c(snapshots)
    rows = np.zeros((n_streams, n_blocks), METER_ROW_DTYPE)
    f = api.fn("loudness_meters", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double,
                                            C.c_double, C.c_void_p, C.c_void_p, C.c_void_p])
    api.check(f(C.byref(arr), 0, n_streams, n_blocks, left_mode, right_mode, t0, dt, holds.ctypes.data, None, rows.ctypes.data))
    return rows


# --------------------------------------------------------------------------- reassigned-splat accumulation
FREQ_SCALE_LINEAR, FREQ_SCALE_LOGARITHMIC, FREQ_SCALE_ERB = range(3)


class CSplatView(C.Structure):
    _fields_ = [("extent_x", C.c_float), ("extent_y", C.c_float), ("scale_factor", C.c_float), ("freq_scale", C.c_uint32),
                ("freq_min", C.c_float), ("freq_max", C.c_float), ("uv_lo", C.c_float), ("uv_hi", C.c_float), ("tilt_db", C.c_float),
                ("width", C.c_uint32), ("height", C.c_uint32)]


def splat_view(api: Api, extent_x: float, extent_y: float, sample_rate: float = DEFAULT_SAMPLE_RATE, scale_factor: float = 1.0,
               freq_scale: int = FREQ_SCALE_LOGARITHMIC, uv=(0.0, 1.0), tilt_db: float = 0.0) -> CSplatView:
    """display_axis (reference src/visuals/spectrogram/state.rs:48-51) + accumulation-target size (render.rs:523-529)."""
    nyq = max(sample_rate / 2.0, 1.0)
    v = CSplatView(extent_x, extent_y, scale_factor, freq_scale, min(1.0, nyq * 0.5), nyq, uv[0], uv[1], tilt_db, 0, 0)
    api.fn("splat_view_size", None, [C.c_void_p])(C.byref(v))
    return v


def spectrogram_splat(api: Api, columns: Sequence[np.ndarray], view: CSplatView, reassigned_power_scale: float, want_db: bool = True):
    """Host convenience for one stream: `columns` = reassigned columns oldest -> newest ([n, 3] float32 each).
    Returns (accum, db) as [height, width] float32."""
    stride = max(1, max((len(c) for c in columns), default=1))
    pts = np.zeros((len(columns), stride, 3), np.float32)
    counts = np.zeros(len(columns), np.uint32)
    for i, c in enumerate(columns):
        pts[i, :len(c)] = c
        counts[i] = len(c)
    accum = np.zeros((view.width, view.height), np.float32)   # C layout: [width][height]
    db = np.zeros((view.width, view.height), np.float32)
    f = api.fn("spectrogram_splat", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p])
    api.check(f(pts.ctypes.data, counts.ctypes.data, 0, 1, len(columns), stride, reassigned_power_scale, C.byref(view), None,
                accum.ctypes.data, db.ctypes.data if want_db else None))
    return accum.T, db.T


# --------------------------------------------------------------------------- column history ring
class CSpectrogramHistoryInfo(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("ring_capacity", C.c_uint32), ("write_slot", C.c_uint32), ("col_count", C.c_uint32),
                ("points_per_column", C.c_uint32), ("reassigned_points_per_slot", C.c_uint32), ("newest_slot", C.c_uint32),
                ("visible_slots", C.c_uint32)]


class SpectrogramHistory:
    """SpectrogramHistory::apply_update + the renderer's column ring (reference src/visuals/spectrogram/state.rs:53-175,
    render.rs:106-160, 457-597) for `n_streams` lock-step streams (the oracle models one)."""

    def __init__(self, api: Api, n_streams: int = 1):
        self.api, self.n_streams = api, n_streams
        self._h = C.c_void_p()
        api.check(api.fn("spectrogram_history_create", C.c_int, [C.c_uint32, C.POINTER(C.c_void_p)])(n_streams, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self.api.fn("spectrogram_history_destroy", None, [C.c_void_p])(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def apply(self, update: "SpectrogramUpdate"):
        """one single-stream update (host columns)"""
        cols = update.new_columns
        offs = np.zeros(len(cols) + 1, np.uint64)
        for i, c in enumerate(cols):
            offs[i + 1] = offs[i] + len(c)
        cu = CSpectrogramUpdate()
        cu.fft_size, cu.hop_size, cu.history_length, cu.n_columns = update.fft_size, update.hop_size, update.history_length, len(cols)
        cu.sample_rate, cu.reassigned_power_scale, cu.reset, cu.kind = update.sample_rate, update.reassigned_power_scale, int(update.reset), update.kind
        cu.column_offsets = offs.ctypes.data_as(C.POINTER(C.c_uint64))
        keep = None
        if update.kind == COLUMN_REASSIGNED:
            keep = (np.concatenate([np.asarray(c, np.float32).reshape(-1, 3) for c in cols]) if cols else np.zeros((0, 3), np.float32))
            keep = np.ascontiguousarray(keep, np.float32)
            cu.points = C.cast(keep.ctypes.data, C.POINTER(CSpectrogramPoint))
        else:
            keep = np.ascontiguousarray(np.concatenate([np.asarray(c, np.uint16) for c in cols]) if cols else np.zeros(0, np.uint16))
            cu.codes = keep.ctypes.data_as(C.POINTER(C.c_uint16))
        self.api.check(self.api.fn("spectrogram_history_apply", C.c_int, [C.c_void_p, C.c_void_p])(self._h, C.byref(cu)))

    def apply_bank(self, bank_update, stream: int = 0):
        """one bank update (device-resident columns): CSpectrogramBankUpdate as returned by banks.SpectrogramBank"""
        self.api.check(self.api.fn("spectrogram_bank_history_apply", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p])(
            self._h, C.byref(bank_update), C.c_void_p(stream or 0)))

    def info(self) -> CSpectrogramHistoryInfo:
        out = CSpectrogramHistoryInfo()
        self.api.check(self.api.fn("spectrogram_history_get_info", C.c_int, [C.c_void_p, C.c_void_p])(self._h, C.byref(out)))
        return out

    def slot_counts(self, stream_index: int = 0) -> np.ndarray:
        cap = self.info().ring_capacity
        out = np.zeros(max(cap, 1), np.uint32)
        f = self.api.fn("spectrogram_history_slot_counts", C.c_int64, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64])
        rc = f(self._h, stream_index, out.ctypes.data, cap)
        if rc < 0:
            self.api.check(int(rc))
        return out[:cap]

    def fetch_slot(self, slot: int, stream_index: int = 0) -> np.ndarray:
        i = self.info()
        n = C.c_uint64()
        buf = np.zeros((i.points_per_column, 3), np.float32) if i.kind == COLUMN_REASSIGNED else np.zeros(i.points_per_column, np.uint16)
        self.api.check(self.api.fn("spectrogram_history_fetch_slot", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                                                                               C.POINTER(C.c_uint64)])(
            self._h, stream_index, slot, buf.ctypes.data, i.points_per_column, C.byref(n)))
        return buf[:n.value]

    def splat(self, view: CSplatView, reassigned_power_scale: float):
        """host images of every stream: (accum, db) as [n_streams][height][width] float32"""
        accum = np.zeros((self.n_streams, view.width, view.height), np.float32)
        db = np.zeros_like(accum)
        self.api.check(self.api.fn("spectrogram_history_splat", C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                                          C.c_void_p])(
            self._h, reassigned_power_scale, C.byref(view), 0, None, accum.ctypes.data, db.ctypes.data))
        return accum.transpose(0, 2, 1), db.transpose(0, 2, 1)
