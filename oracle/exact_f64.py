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
