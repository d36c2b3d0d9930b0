"""ctypes wrappers for the oracle's `omxo_kat_*` exports (oracle/capi.cpp).  Test infra only."""
import ctypes as C

import numpy as np

from openmeters_amd.capi import AudioBlock, CBlock

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


def _p(a, typ=_f32p):
    return a.ctypes.data_as(typ)


class Kat:
    def __init__(self, api):
        self.api = api
        self.lib = api.lib

    def _fn(self, name, restype, argtypes):
        f = getattr(self.lib, "omxo_kat_" + name)
        f.restype = restype
        f.argtypes = argtypes
        return f

    # primitives -------------------------------------------------------------------
    def power_to_db(self, p, floor):
        return float(self._fn("power_to_db", C.c_float, [C.c_float, C.c_float])(p, floor))

    def db_to_power(self, db):
        return float(self._fn("db_to_power", C.c_float, [C.c_float])(db))

    def sanitize_sample_rate(self, r):
        return float(self._fn("sanitize_sample_rate", C.c_float, [C.c_float])(r))

    def window(self, kind, n):
        out = np.zeros(n, np.float32)
        self._fn("window", None, [C.c_uint32, C.c_uint64, _f32p])(kind, n, _p(out))
        return out

    def bin_normalization(self, window, fft_size):
        window = np.ascontiguousarray(window, np.float32)
        out = np.zeros(fft_size // 2 + 1, np.float32)
        self._fn("bin_normalization", None, [_f32p, C.c_uint64, C.c_uint64, _f32p])(
            _p(window), window.size, fft_size, _p(out))
        return out

    def stereo_frames(self, block: AudioBlock):
        cb = block.to_c()
        frames = block.samples.size // block.channels
        out = np.zeros((frames, 2), np.float32)
        matrix = np.zeros((8, 2), np.float32)
        sc = self._fn("stereo_frames", C.c_uint64, [C.POINTER(CBlock), _f32p, _f32p])(C.byref(cb), _p(out), _p(matrix))
        return out, matrix, int(sc)

    def windowed_means(self, capacities, values):
        caps = np.array(list(capacities) + [1] * (4 - len(capacities)), np.uint64)
        values = np.ascontiguousarray(values, np.float64)
        means = np.zeros(4, np.float64)
        self._fn("windowed_means", None, [C.POINTER(C.c_uint64), C.c_uint32, _f64p, C.c_uint64, _f64p])(
            caps.ctypes.data_as(C.POINTER(C.c_uint64)), len(capacities), _p(values, _f64p), values.size, _p(means, _f64p))
        return means[:len(capacities)]

    def biquad(self, highpass, sample_rate, frequency, x, clear_after=-1):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        coeffs = (C.c_float * 5)()
        self._fn("biquad", None, [C.c_int, C.c_float, C.c_float, _f32p, C.c_uint64, C.c_int64, _f32p, C.c_float * 5])(
            int(highpass), sample_rate, frequency, _p(x), x.size, clear_after, _p(out), coeffs)
        return out, np.array(coeffs[:], np.float32)

    def threeband_lr4(self, sample_rate, lr):
        lr = np.ascontiguousarray(lr, np.float32).reshape(-1, 2)
        out = np.zeros((lr.shape[0], 3, 2), np.float32)
        self._fn("threeband_lr4", None, [C.c_float, _f32p, C.c_uint64, _f32p])(sample_rate, _p(lr), lr.shape[0], _p(out))
        return out

    def fft_f32(self, z, inverse=False):
        buf = np.ascontiguousarray(z, np.complex64).copy()
        self._fn("fft_f32", None, [_f32p, C.c_uint64, C.c_int])(buf.view(np.float32).ctypes.data_as(_f32p), buf.size, int(inverse))
        return buf

    def fft_f64(self, z, inverse=False):
        buf = np.ascontiguousarray(z, np.complex128).copy()
        self._fn("fft_f64", None, [_f64p, C.c_uint64, C.c_int])(buf.view(np.float64).ctypes.data_as(_f64p), buf.size, int(inverse))
        return buf

    # spectrum ---------------------------------------------------------------------
    def smoothing_state_floor(self, weighting, floor):
        w = np.ascontiguousarray(weighting, np.float32)
        return float(self._fn("smoothing_state_floor", C.c_float, [_f32p, C.c_uint64, C.c_float])(_p(w), w.size, floor))

    def level_update(self, state_floor, smoothed_init, scratch_power, mode, param, weighting_db, dt, floor):
        out = (C.c_float * 3)()
        self._fn("level_update", None, [C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_float * 3])(
            state_floor, smoothed_init, scratch_power, mode, param, weighting_db, dt, floor, out)
        return {"weighted": out[0], "raw": out[1], "smoothed": out[2]}

    # loudness ---------------------------------------------------------------------
    def true_peak_coefficient(self, j, factor):
        return float(self._fn("true_peak_coefficient", C.c_float, [C.c_uint64, C.c_uint64])(j, factor))

    def true_peak_delay_len(self, rate):
        return int(self._fn("true_peak_delay_len", C.c_uint64, [C.c_double])(rate))

    def channel_weight(self, pos):
        return float(self._fn("channel_weight", C.c_double, [C.c_uint8])(pos))

    def window_length(self, rate, secs):
        return int(self._fn("window_length", C.c_uint64, [C.c_float, C.c_float])(rate, secs))

    # stereometer ------------------------------------------------------------------
    def correlation(self, pairs, alpha):
        p = np.ascontiguousarray(pairs, np.float32).reshape(-1, 2)
        return float(self._fn("correlation", C.c_float, [_f32p, C.c_uint64, C.c_double])(_p(p), p.shape[0], alpha))

    def ema_alpha(self, rate, window):
        return float(self._fn("ema_alpha", C.c_double, [C.c_float, C.c_float])(rate, window))

    # oscilloscope -----------------------------------------------------------------
    def estimate_period(self, samples, rate):
        s = np.ascontiguousarray(samples, np.float32)
        period, conf = C.c_float(), C.c_float()
        ok = self._fn("estimate_period", C.c_int, [_f32p, C.c_uint64, C.c_float, _f32p, _f32p])(
            _p(s), s.size, rate, C.byref(period), C.byref(conf))
        return (period.value, conf.value) if ok else None

    def stable_trigger_positions(self, signal, block, n_blocks, rate, segment_duration=0.02, cycles=2):
        s = np.ascontiguousarray(signal, np.float32)
        pos = np.zeros(n_blocks, np.float32)
        locked = np.zeros(n_blocks, np.uint8)
        self._fn("stable_trigger_positions", None, [_f32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float,
                                                    C.c_uint64, _f32p, C.POINTER(C.c_uint8)])(
            _p(s), s.size, block, n_blocks, rate, segment_duration, cycles, _p(pos), locked.ctypes.data_as(C.POINTER(C.c_uint8)))
        return pos, locked.astype(bool)

    def retune_reference(self, reference, old_period, new_period, len_out):
        r = np.ascontiguousarray(reference, np.float32).copy()
        out = np.zeros(len_out, np.float32)
        self._fn("retune_reference", None, [_f32p, C.c_uint64, C.c_float, C.c_float, C.c_uint64, _f32p])(
            _p(r), r.size, old_period, new_period, len_out, _p(out))
        return out

    def prepare_template(self, length, period):
        out = np.zeros(length, np.float32)
        self._fn("prepare_template", None, [C.c_uint64, C.c_float, _f32p])(length, period, _p(out))
        return out

    def write_candidate(self, reference, segment, period):
        r = np.ascontiguousarray(reference, np.float32)
        s = np.ascontiguousarray(segment, np.float32)
        cand = np.zeros(s.size, np.float32)
        v = self._fn("write_candidate", C.c_float, [_f32p, C.c_uint64, _f32p, C.c_uint64, C.c_float, _f32p])(
            _p(r), r.size, _p(s), s.size, period, _p(cand))
        return float(v), cand

    def find_best(self, candidate, work, search, period):
        c = np.ascontiguousarray(candidate, np.float32)
        w = np.ascontiguousarray(work, np.float32)
        frac = C.c_float()
        off = self._fn("find_best", C.c_uint64, [_f32p, C.c_uint64, _f32p, C.c_uint64, C.c_uint64, C.c_float, _f32p])(
            _p(c), c.size, _p(w), w.size, search, period, C.byref(frac))
        return int(off), frac.value

    def find_rising_zero_crossing(self, samples, lo, hi, reversed_=False):
        s = np.ascontiguousarray(samples, np.float32)
        idx = C.c_uint64()
        ok = self._fn("find_rising_zero_crossing", C.c_int, [_f32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                                             C.POINTER(C.c_uint64)])(
            _p(s), s.size, lo, hi, int(reversed_), C.byref(idx))
        return int(idx.value) if ok else None
