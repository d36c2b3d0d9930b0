"""ORACLE — test infrastructure only (never imported by the product).

Exact-arithmetic third leg through the FFT boundary: an f64 numpy restatement of ONE spectrogram column, reassigned and
classic, straight from the reference's formulas.  The reference's own FFTs live in un-vendored `rustfft 6.4.1` /
`realfft 3.5.0` whose f32 rounding cannot be reproduced here, so the honest pin for "1e-5 vs the reference" is the distance
of BOTH implementations (the C++ oracle and the HIP product) from exact arithmetic: `tests/test_exact_f64.py` asserts each is
within 1e-5 of the column maximum of this restatement and within 2x of the other.

Everything is computed in f64 from the f32 PCM samples; tables are the mathematical ones (no f32 rounding anywhere).
Reference = /root/reference/src (file:line below).
"""
import numpy as np

TAU = 2.0 * np.pi

# util/audio/window.rs:20-43 — periodic cosine-sum windows, coefficient sets per kind (0 rect, 1 Hann, 2 Hamming, 3 Blackman,
# 4 Blackman-Harris): w[n] = sum_k c_k cos(k * n * 2 pi / len)
WINDOW_COEFFS = {
    0: [1.0],
    1: [0.5, -0.5],
    2: [25.0 / 46.0, -21.0 / 46.0],
    3: [0.42, -0.5, 0.08],
    4: [0.35875, -0.48829, 0.14128, -0.01168],
}


def window(kind, n):
    phi = np.arange(n, dtype=np.float64) * (TAU / n)
    return sum(c * np.cos(phi * k) for k, c in enumerate(WINDOW_COEFFS[kind]))


def bin_normalization(w, fft_size):
    """window.rs:90-109: 4 / (sum w)^2, DC and Nyquist 1 / (sum w)^2"""
    s = w.sum()
    inv = 1.0 / (s * s) if s > 0 else 0.0
    norm = np.full(fft_size // 2 + 1, 4.0 * inv)
    norm[0] = inv
    if fft_size % 2 == 0:
        norm[-1] = inv
    return norm


def derivative_window(w):
    """spectrogram/processor.rs:569-599: FFT(w), DC and Nyquist zeroed, times i omega_k, inverse, / W, real part"""
    n = len(w)
    spec = np.fft.fft(w)
    k = np.arange(n)
    omega = TAU / n * np.where(k > n // 2, k - n, k)
    spec = spec * (1j * omega)
    spec[0] = 0.0
    if n % 2 == 0:
        spec[n // 2] = 0.0
    return np.fft.ifft(spec).real     # numpy's ifft already carries the 1/W


def reassigned_column(x, window_kind=1, window_size=4096, zero_padding=1, hop=256, sample_rate=48000.0):
    """One reassigned column from the first H = next_pow2(2 W) samples of `x` (mono, the projected ring content).
    spectrogram/processor.rs:318-348 (driver), :546-557 (Hilbert), :559-567 (windows), :439-488 (points).
    Returns float64 [n][3] = (time_offset, freq_hz, power), ascending bin, plus the bin index of each point."""
    W, F = window_size, window_size * zero_padding
    H = max(int(2 ** np.ceil(np.log2(2 * W))), 2)
    x = np.asarray(x[:H], dtype=np.float64)
    assert len(x) == H
    spec = np.fft.fft(x)
    spec[0] = 0.0                      # :554
    spec[H // 2 + 1:] = 0.0            # :555 (Nyquist bin kept, no doubling)
    analytic = np.fft.ifft(spec) * H   # unnormalised inverse; the 1/H^2 lives in bin_norm (:263-266)
    center = (H - W) // 2
    s = analytic[center:center + W]
    w = window(window_kind, W)
    dw = derivative_window(w)
    tw = (np.arange(W) - (W - 1) * 0.5) * w          # :601-608
    B = np.fft.fft(s * w, F)
    D = np.fft.fft(s * dw, F)
    T = np.fft.fft(s * tw, F)
    norm = bin_normalization(w, F) / (float(H) * float(H))
    bins = np.arange(F // 2 + 1)
    B, D, T = B[:F // 2 + 1], D[:F // 2 + 1], T[:F // 2 + 1]
    pw = B.real ** 2 + B.imag ** 2
    scaled = pw * norm
    keep = scaled >= 1e-14                             # ANALYSIS_FLOOR_POWER (:69, :462)
    with np.errstate(divide="ignore", invalid="ignore"):
        d_omega = -(D.imag * B.real - D.real * B.imag) / pw
        freq = bins * (sample_rate / F) + d_omega * (sample_rate / TAU)
        t = (T.real * B.real + T.imag * B.imag) / pw / hop - center / hop
    keep &= (freq > 0.0) & (freq < sample_rate * 0.5)  # :471
    pts = np.stack([t[keep], freq[keep], scaled[keep]], 1)
    return pts, bins[keep]


def classic_column_power(x, window_kind=1, window_size=1024, zero_padding=1):
    """Linear power per bin of one classic column from the first W samples of `x`: window.rs:66-88 (mean removal + window),
    spectrogram/processor.rs:350-380 (real FFT, |X|^2 * bin_norm).  dB / u16 packing is left to the caller."""
    W, F = window_size, window_size * zero_padding
    x = np.asarray(x[:W], dtype=np.float64)
    w = window(window_kind, W)
    y = (x - x.sum() / W) * w
    X = np.fft.rfft(y, F)
    return (X.real ** 2 + X.imag ** 2) * bin_normalization(w, F)


def codes_to_power(codes):
    """inverse of pack_classic_db (:103-108): code -> dB -> linear power"""
    db = np.asarray(codes, dtype=np.float64) * (156.0 / 65535.0) - 144.0
    return 10.0 ** (db / 10.0)


# ------------------------------------------------------------------------------------------------------------------------
# Oscilloscope, Stable trigger — an f64 restatement of the capture of ONE trace (the linked-trigger shape: trigger_source ==
# channel_1), statement by statement from reference src/visuals/oscilloscope/processor.rs: PeriodEstimator (:85-182),
# StableTrigger (:272-528) and the helpers (:14-19, :184-263).  All arithmetic in f64 on the f32 PCM samples; integer decisions
# (lengths, strides, the coarse-to-fine walk) as in the reference.  `frac_offset` divides a score difference by the curvature of
# a flat correlation peak (parabolic_refine), so it is the quantity whose f32 error is largest: tests/test_exact_f64.py compares
# the C++ oracle's and the HIP product's frac_offset with this one.
class ScopeTraceExact:
    MIN_HZ, MAX_HZ, PROBE_SECONDS, MIN_SIGNAL_PEAK, MIN_PERIODICITY, PEAK_CUTOFF = 20.0, 8000.0, 0.1, 0.001, 0.5, 0.93   # :86-91
    WINDOW_SECONDS, MIN_CYCLES, SEARCH_PERIODS, NORMALIZE_FLOOR, MEAN_RESPONSIVENESS = 0.04, 2.0, 1.5, 0.01, 0.25       # :285-296
    BUFFER_RESPONSIVENESS, BUFFER_FALLOFF_PERIODS, BUFFER_RETUNE_SEMITONES, SLOPE_WIDTH_PERIODS = 0.5, 0.5, 1.0, 0.25
    RESET_BELOW_MATCH, MAX_MISSED_PERIODS = 0.3, 4
    EPS32 = float(np.finfo(np.float32).eps)   # the reference's f32::EPSILON thresholds keep their f32 value

    def __init__(self, sample_rate=48000.0, segment_duration=0.02, num_cycles=2):
        self.rate, self.cycles = float(sample_rate), int(num_cycles)
        r32 = np.float32(sample_rate)
        self.base_frames = int(max(np.round(r32 * np.float32(segment_duration)), 1.0))                       # :631-633
        self.max_period = int(np.ceil(r32 / np.float32(self.MIN_HZ)))                                          # :634
        self.probe_frames = max(int(np.round(r32 * np.float32(self.PROBE_SECONDS))), self.max_period * 2)      # :635-636
        max_kernel = self._kernel_len(float(self.max_period))
        max_tail = max(self.max_period * max(self.cycles, 1) + 1, -(-max_kernel // 2))                          # :761-767
        self.history = max(self.probe_frames, self.base_frames,
                           max_kernel // 2 + max_tail + int(np.ceil(self.max_period * self.SEARCH_PERIODS)) + 2)
        self.trace = np.zeros(0, np.float64)
        self.period = None
        self.missed = 0
        self.reference = np.zeros(0)
        self.reference_period = 0.0
        self.mean = 0.0
        self.last_peak = 0.0

    @staticmethod
    def _round_half_away(x):
        return float(np.sign(x) * np.floor(abs(x) + 0.5))

    def _kernel_len(self, period):  # :184-189
        return int(max(self._round_half_away(max(self.rate * self.WINDOW_SECONDS, period * self.MIN_CYCLES)), 2.0))

    @staticmethod
    def _parabolic(yp, yc, yn, tau, eps):  # :14-19
        denom = yp - 2.0 * yc + yn
        if abs(denom) < eps:
            return float(tau)
        return max(tau + min(max(0.5 * (yp - yn) / denom, -1.0), 1.0), 1.0)

    def _estimate(self, samples):  # :93-131 + compute_periodicity :133-181
        self.last_peak = 0.0
        n = len(samples)
        if n < 3:
            return None
        mean = samples.sum() / n
        self.last_peak = np.abs(samples - mean).max()
        if self.last_peak < self.MIN_SIGNAL_PEAK:
            return None
        min_period = int(max(self._round_half_away(self.rate / self.MAX_HZ), 2.0))
        max_period = min(int(self._round_half_away(self.rate / self.MIN_HZ)), n // 2)
        if max_period <= min_period + 1:
            return None
        size = 1 << int(np.ceil(np.log2(n + max_period)))
        c = samples - mean
        energy = np.concatenate([[0.0], np.cumsum(c * c)])
        spec = np.fft.rfft(c, size)
        acf = np.fft.irfft(spec.real ** 2 + spec.imag ** 2, size) * size   # the reference's inverse is unnormalised ...
        if energy[n] <= self.EPS32:
            return None
        tau = np.arange(max_period + 1)
        denom = energy[n - tau] + (energy[n] - energy[tau])
        nsdf = np.where(denom > self.EPS32, 2.0 * acf[:max_period + 1] * (1.0 / size) / np.where(denom > self.EPS32, denom, 1.0), 0.0)  # ... * norm
        zc = np.nonzero(nsdf[1:] <= 0.0)[0]
        if len(zc) == 0:
            return None
        first_tau = max(min_period, int(zc[0]) + 1)
        if first_tau >= max_period:
            return None
        cand = [t for t in range(first_tau, max_period) if nsdf[t] >= self.MIN_PERIODICITY and nsdf[t] >= nsdf[t - 1] and nsdf[t] >= nsdf[t + 1]]
        if not cand:
            return None
        best = max(cand, key=lambda t: (nsdf[t], t))          # max_by keeps the last maximum
        cutoff = nsdf[best] * self.PEAK_CUTOFF
        peak = next((t for t in cand if t <= best and nsdf[t] >= cutoff), best)
        return self._parabolic(nsdf[peak - 1], nsdf[peak], nsdf[peak + 1], peak, self.EPS32), min(max(nsdf[peak], 0.0), 1.0)

    def _unlock(self):  # :298-304
        self.period, self.missed, self.reference, self.reference_period, self.mean = None, 0, np.zeros(0), 0.0, 0.0

    @staticmethod
    def _gauss(n, i, std, eps):  # :199-204
        if n <= 1 or std <= eps:
            return np.zeros_like(np.asarray(i, np.float64))
        return np.exp(-0.5 * ((np.asarray(i, np.float64) - (n - 1) * 0.5) / std) ** 2)

    def _corr(self, x, y):  # :206-236
        n = len(x)
        if n == 0:
            return 0.0
        sx, sxx, sxy, sy, syy = x.sum(), (x * x).sum(), (x * y).sum(), y.sum(), (y * y).sum()
        dot = sxy - sx * sy / n
        ex, ey = max(sxx - sx * sx / n, 0.0), max(syy - sy * sy / n, 0.0)
        den = np.sqrt(ex * ey)
        return min(max(dot / den, -1.0), 1.0) if den > self.EPS32 else 0.0

    def _template(self, n, period, use_reference):  # :422-439
        width = min(max(self.SLOPE_WIDTH_PERIODS * period, 1.0), max(max(n // 2, 1) / 3.0, 1.0))
        t = np.zeros(n)
        half = -(-n // 2)
        i = np.arange(half)
        w = self._gauss(n, i, width, self.EPS32)
        t[i] = -w
        t[n - 1 - i] = w          # written second: the middle element of an odd length ends up +w
        return t + self.reference if use_reference else t

    def _write_candidate(self, seg, period):  # :509-527
        c = seg - seg.sum() / max(len(seg), 1)
        c = c * (1.0 / max(np.abs(c).max() if len(c) else 0.0, self.NORMALIZE_FLOOR))
        n = len(c)
        std = max(period * self.BUFFER_FALLOFF_PERIODS, 1.0)
        i = np.arange(n)
        c = c * self._gauss(n, np.minimum(i, n - 1 - i), std, self.EPS32)
        return c, self._corr(self.reference, c)

    def _find_best(self, work, tmpl, search, period):  # :441-484
        n = len(tmpl)
        scores = {}

        def score_at(o):
            if o not in scores:
                scores[o] = self._corr(work[o:o + n], tmpl)
            return scores[o]
        stride = min(min(max(int(self._round_half_away(period / 16.0)), 1), 128), max(search, 1))
        best = (search // 2, -np.inf)
        for o in list(range(search, -1, -stride)) + [0]:
            s = score_at(o)
            if s > best[1]:
                best = (o, s)
        step = stride
        while step > 1:
            nxt = max(step // 4, 1)
            for o in range(min(best[0] + step, search), max(best[0] - step, 0) - 1, -nxt):
                s = score_at(o)
                if s > best[1]:
                    best = (o, s)
            step = nxt
        frac = 0.0
        if 0 < best[0] < search:
            frac = min(max(self._parabolic(score_at(best[0] - 1), best[1], score_at(best[0] + 1), best[0], self.EPS32) - best[0], -0.5), 0.5)
        return best[0], frac

    def _locate(self, trace, period_in, confidence):  # :358-411
        period = max(period_in, 1.0)
        span = period * max(self.cycles, 1)
        frames = int(np.ceil(span)) + 1
        n = self._kernel_len(period)
        before = n // 2
        after = n - before
        if len(trace) < max(frames, after):
            return None
        right = len(trace) - max(frames, after)
        if right < before:
            return None
        search = min(max(int(self._round_half_away(period * self.SEARCH_PERIODS)), 1), n // 2, right - before)
        left = right - search
        data = trace[left - before:right + after]
        # prepare (:413-420) with retune_reference (:486-498, :249-263)
        if len(self.reference) == 0:
            self.reference, self.reference_period = np.zeros(n), period
        else:
            semis = np.log2(period / self.reference_period) * 12.0
            if len(self.reference) != n or abs(semis) >= self.BUFFER_RETUNE_SEMITONES:
                ratio = period / self.reference_period
                old = self.reference
                if not np.isfinite(ratio) or ratio <= self.EPS32:
                    self.reference = np.zeros(n)
                else:
                    pos = (len(old) - 1) * 0.5 + (np.arange(n) - (n - 1) * 0.5) / ratio
                    ok = (pos >= 0) & (pos <= len(old) - 1)
                    idx = np.clip(np.floor(pos).astype(int), 0, len(old) - 1)
                    fr = pos - idx
                    nxt = np.clip(idx + 1, 0, len(old) - 1)
                    lin = np.where((fr > self.EPS32) & (idx + 1 < len(old)), old[idx] + (old[nxt] - old[idx]) * fr, old[idx])
                    self.reference = np.where(ok, lin, 0.0)
                self.reference_period = period
        self.mean += self.MEAN_RESPONSIVENESS * (data.sum() / max(len(data), 1) - self.mean)
        work = data - self.mean
        use_reference = bool(np.any(np.abs(self.reference) > 1.0e-3))
        tmpl = self._template(n, period, use_reference)
        offset, frac = self._find_best(work, tmpl, search, period)
        confident = confidence >= self.MIN_PERIODICITY
        seg = lambda o: trace[left + o - before:left + o - before + n]
        cand, reset = None, False
        if confident and use_reference:
            cand, match = self._write_candidate(seg(offset), period)
            reset = match < self.RESET_BELOW_MATCH
        if reset:
            self.reference = np.zeros(n)
            tmpl = self._template(n, period, False)
            offset, frac = self._find_best(work, tmpl, search, period)
        if confident:
            if not use_reference or reset:
                cand, _ = self._write_candidate(seg(offset), period)
            ref = self.reference * (1.0 / max(np.abs(self.reference).max(), self.NORMALIZE_FLOOR))   # update_reference :500-507
            self.reference = ref + self.BUFFER_RESPONSIVENESS * (cand - ref)
            self.reference_period += self.BUFFER_RESPONSIVENESS * (period - self.reference_period)
        start = left + offset
        if frac < 0.0 and start > 0:
            start -= 1
            frac += 1.0
        return span, start, frac

    def process_block(self, samples):
        """Push one block of the trace's (projected) samples; returns (span, start, frac_offset) of the capture, or None."""
        self.trace = np.concatenate([self.trace, np.asarray(samples, np.float64)])[-self.history:]
        trace = self.trace
        if len(trace) < self.base_frames:
            return None
        probe = trace[len(trace) - min(self.probe_frames, len(trace)):]
        detected = self._estimate(probe) if len(probe) >= 3 else None       # capture :306-334
        if len(probe) > 0 and self.last_peak < self.MIN_SIGNAL_PEAK:
            self._unlock()
        est = None                                                          # stabilize :336-356
        if detected is None:
            if self.period is not None:
                self.missed = min(self.missed + 1, 255)
                if self.missed > self.MAX_MISSED_PERIODS:
                    self._unlock()
                else:
                    est = (self.period, 0.0)
        else:
            self.missed = 0
            p, conf = detected
            if self.period is not None and 0.9 <= p / self.period <= 1.1:
                p = self.period + 0.35 * (p - self.period)
            self.period = p
            est = (p, conf)
        cap = self._locate(trace, *est) if est is not None else None
        if cap is None:
            return float(max(max(self.base_frames - 1, 0), 1)), max(len(trace) - self.base_frames, 0), 0.0
        return cap


# ------------------------------------------------------------------------------------------------------------------------
# Stereometer — an f64 restatement of the band split and the correlators (reference src/visuals/stereometer/processor.rs:34-61,
# :99-182; src/dsp.rs:399-495).  The filter COEFFICIENTS are the reference's f32 values (Biquad::new evaluates them in f32, and a
# pole pair at 200 Hz / 48 kHz turns a coefficient difference of 6e-8 into 1e-4 of response): what is exact here is the
# RECURRENCE — scipy's f64 lfilter on the f32 samples — and the f64 moving averages of the correlators.  Third leg of the
# chunk-parallel stereometer form (tests/test_exact_f64.py): the sequential f32 order of the reference and the re-ordered, fused
# order of the kernels are two f32 evaluations of the same recurrence, each 1e-5 ... 2e-5 of full scale away from this one.
class StereometerExact:
    BAND_DISPLAY_GAIN = 0.8   # :8

    def __init__(self, sample_rate=48000.0, segment_duration=0.02, target_sample_count=2000, correlation_window=0.05):
        self.rate = float(sample_rate)
        r32 = np.float32(sample_rate)
        self.frames = int(max(np.round(r32 * np.float32(segment_duration)), 1.0))                    # :163
        self.target = min(max(int(target_sample_count), 1), self.frames)
        self.alpha = 1.0 - np.exp(-1.0 / max(float(r32) * float(np.float32(correlation_window)), 1.0))   # :210-212

    def _biquad(self, highpass, f):   # Biquad::new (dsp.rs:402-420), every step in f32
        f32 = np.float32
        ratio = np.clip(f32(f) / f32(self.rate), f32(1e-6), f32(0.49))
        ang = f32(6.2831855) * ratio
        sin, cos = f32(np.sin(ang)), f32(np.cos(ang))
        alpha = sin * f32(0.70710677)
        gain, sign = (f32(1) + cos, f32(-1)) if highpass else (f32(1) - cos, f32(1))
        inv = f32(1) / (f32(1) + alpha)
        return (np.array([gain * f32(0.5) * inv, gain * inv * sign, gain * f32(0.5) * inv], np.float64),
                np.array([1.0, f32(-2) * cos * inv, (f32(1) - alpha) * inv], np.float64))

    def run(self, pcm_lr):
        """pcm_lr [frames][2] f32, the whole stream from reset.  Returns (points [4][target][2], correlations [4]) as the snapshot
        after the last frame would hold them: band order full, low, mid, high (ThreeBand<_, true>: the high band is cut from `above_low`)."""
        from scipy.signal import lfilter

        def lr4(c, v):
            return lfilter(c[0], c[1], lfilter(c[0], c[1], v, axis=0), axis=0)
        x = np.asarray(pcm_lr, np.float64)
        above = lr4(self._biquad(True, 200.0), x)
        bands = [x, lr4(self._biquad(False, 200.0), x), lr4(self._biquad(False, 2000.0), above), lr4(self._biquad(True, 2000.0), above)]
        idx = (np.arange(self.target) * self.frames) // self.target                                  # :171 hist[i * frames / target]
        points, rho = [], []
        for b, v in enumerate(bands):
            tail = v[-self.frames:]
            points.append(tail[idx] * (1.0 if b == 0 else self.BAND_DISPLAY_GAIN))
            l, r = v[:, 0], v[:, 1]
            ema = [lfilter([self.alpha], [1.0, -(1.0 - self.alpha)], q)[-1] for q in (l * r, l * l, r * r)]   # :36-40 from zero moments
            denom = np.sqrt(ema[1] * ema[2])
            rho.append(0.0 if denom <= 1e-12 else float(np.clip(ema[0] / denom, -1.0, 1.0)))
        return np.array(points), np.array(rho)


# ---- the waveform's band colours and RMS history in f64 (reference src/visuals/waveform/processor.rs:78-121, :213-291;
# src/dsp.rs:298-371, :399-495).  As for the stereometer above: the filter COEFFICIENTS are the reference's f32 values, the
# RECURRENCE runs in f64 (scipy lfilter), the window means are plain f64 means of the last `len` tracker inputs (the reference keeps
# them compensated: exact).  Third leg of the waveform bank's chunk-parallel form (tests/test_exact_f64.py): the reference's
# sequential f32 order and the chunk form — f32 filters restarted every 64 ... 256 frames from scanned states — are two f32
# evaluations of this recurrence.  Finite two-channel (FL, FR) input only; min / max fields are not modelled (bit-exact elsewhere).
class WaveformExact:
    GAINS = (1.0, 0.7, 2.0)   # BAND_COLOR_GAINS (:22), as f32 values
    GRID = 32                 # frames between the window means kept for the scales of the parity bars (run: grid_*)

    def __init__(self, sample_rate=48000.0, scroll_speed=300.0):
        self.rate = float(np.float32(sample_rate))
        self.step = min(max(float(np.float32(scroll_speed)) / float(np.float32(sample_rate)), 0.0), 1.0)   # :253-254
        f32 = np.float32
        ref = min(f32(sample_rate), f32(1.0e6))
        self.color_len = max(int(np.round(f32(2048) * ref / f32(44100.0))), 1)    # window_len (:78-82)
        self.slow_len = max(int(np.round(f32(16384) * ref / f32(44100.0))), 1)

    _biquad = StereometerExact._biquad

    def run(self, pcm_lr, scroll_changes=()):
        """pcm_lr [frames][2] f32, the whole stream from reset.  Returns (column end frames, colour [cols][4][3], power [cols][4][2][3]):
        the colour bands and the mean band powers (fast, slow window) of every column the stream emits.  scroll_changes: (frame,
        scroll_speed) pairs — update_config between two blocks changes the step and nothing else (:336-352)."""
        from scipy.signal import lfilter
        x = np.asarray(pcm_lr, np.float64)
        f = lambda c, v: lfilter(c[0], c[1], v, axis=0)
        above = f(self._biquad(True, 200.0), x)
        sides = np.stack([f(self._biquad(False, 200.0), x), f(self._biquad(False, 2000.0), above), f(self._biquad(True, 2000.0), x)], axis=-1)  # [frames][L/R][band]
        l, r = sides[:, 0], sides[:, 1]
        bands = np.stack([l, r, (l + r) * 0.5, (l - r) * 0.5], axis=1)               # [frames][channel][band] (:262-268)
        gains = np.array([float(np.float32(g)) for g in self.GAINS])
        colour = np.abs(bands) * gains
        power = bands * bands
        zero = np.zeros((1,) + colour.shape[1:])
        # running totals in extended precision (x87, 64-bit mantissa): a window 120 dB below the passage before it is still resolved to
        # ~1e-6 of its own sum (the product keeps double-double totals for the same reason)
        ld = np.longdouble
        cc = np.concatenate([zero.astype(ld), np.cumsum(colour.astype(ld), axis=0)])
        cp = np.concatenate([zero.astype(ld), np.cumsum(power.astype(ld), axis=0)])
        ends, phase = [], 0.0
        steps = np.full(x.shape[0], self.step)
        for frame, scroll in scroll_changes:
            steps[frame:] = min(max(float(np.float32(scroll)) / self.rate, 0.0), 1.0)
        for k in range(x.shape[0]):
            phase += steps[k]
            if phase >= 1.0:
                ends.append(k)
                phase -= 1.0
        ends = np.array(ends, np.int64)

        def mean(c, cap, at=None):
            hi = (ends if at is None else at) + 1
            lo = np.maximum(hi - cap, 0)
            return ((c[hi] - c[lo]) / np.maximum(np.minimum(hi, cap), 1)[:, None, None]).astype(np.float64)
        # the same window means on a grid of frames (every GRID-th frame, whether a column ends there or not): the scale "the loudest the
        # windows have been within reach" must not depend on where the columns happen to end (at 10 columns per second a 40 ms passage
        # can lie between two of them)
        self.grid_ends = np.arange(self.GRID - 1, x.shape[0], self.GRID, dtype=np.int64)
        self.grid_colour = mean(cc, self.color_len, self.grid_ends)
        self.grid_power = np.stack([mean(cp, self.color_len, self.grid_ends), mean(cp, self.slow_len, self.grid_ends)], axis=2)
        return ends, mean(cc, self.color_len), np.stack([mean(cp, self.color_len), mean(cp, self.slow_len)], axis=2)
