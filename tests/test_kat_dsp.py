"""Known-answer tests ported from the reference's in-file unit tests for the shared primitives:
src/dsp.rs:510-665, src/util/audio/window.rs:115-122, src/util/audio/level.rs:45-48.
These pin the ORACLE (CPU restatement); the HIP path is then compared with the oracle in test_gpu_*."""
import numpy as np
import pytest

from openmeters_amd import capi
from openmeters_amd.capi import AudioBlock
from oracle_kat import Kat

P = capi


@pytest.fixture(scope="module")
def kat(oracle):
    return Kat(oracle)


def test_channel_layouts_fill_unknown_and_duplicate_positions_without_collisions(oracle):
    # dsp.rs:510-556
    unknown = [P.POS_UNKNOWN] * 8
    for channels, expected in [
        (1, [P.POS_MONO]),
        (4, [P.POS_FL, P.POS_FR, P.POS_RL, P.POS_RR]),
        (6, [P.POS_FL, P.POS_FR, P.POS_FC, P.POS_LFE, P.POS_RL, P.POS_RR]),
        (8, P.SURROUND),
    ]:
        assert oracle.positions_normalize(channels, unknown)[:channels] == expected
        assert oracle.positions_fallback(channels)[:channels] == expected
        assert capi.positions_fallback(channels) == oracle.positions_fallback(channels)
    partial = list(unknown)
    partial[:2] = [P.POS_FR, P.POS_UNKNOWN]
    assert oracle.positions_normalize(2, partial)[:2] == [P.POS_FR, P.POS_FL]
    partial[:3] = [P.POS_FL, P.POS_FL, P.POS_FR]
    got = oracle.positions_normalize(3, partial)
    assert got[0] == P.POS_FL and got[2] == P.POS_FR and len(set(got[:3])) == 3


def test_stereo_matrix_folds_semantic_channels_and_ignores_lfe(kat):
    # dsp.rs:558-589
    samples = np.array([1.0, 2.0, 3.0, 100.0, 4.0, 5.0, 6.0, 7.0], np.float32)
    lr, _, _ = kat.stereo_frames(AudioBlock(samples, 8, 48000.0, P.SURROUND))
    gain = np.float32(0.70710678)
    # the reference asserts exact equality with these f32 expressions
    assert lr[0, 0] == np.float32(1.0) + gain * np.float32(13.0)
    assert lr[0, 1] == np.float32(2.0) + gain * np.float32(15.0)
    _, matrix, _ = kat.stereo_frames(AudioBlock([0.25], 1, 48000.0, [P.POS_MONO] + [P.POS_UNKNOWN] * 7))
    assert matrix[0].tolist() == [1.0, 1.0]
    unsupported = [P.POS_LFE, P.POS_AUX0] + [P.POS_UNKNOWN] * 6
    _, matrix, sc = kat.stereo_frames(AudioBlock(np.zeros(0, np.float32), 8, 48000.0, unsupported))
    assert sc == 2 and matrix[:2].tolist() == [[1.0, 0.0], [0.0, 1.0]]


def test_common_stereo_paths_preserve_general_fold_bits(kat):
    # dsp.rs:591-624: the 1/2-channel specialisations equal the general fold bit for bit
    nan = np.array([0x7FC01234], np.uint32).view(np.float32)[0]
    for samples, channels in [
        (np.array([0.0, -0.0, nan, np.inf], np.float32), 1),
        (np.array([0.0, -0.0, 0.25, -0.5, nan, np.inf], np.float32), 2),
    ]:
        lr, matrix, sc = kat.stereo_frames(AudioBlock(samples, channels, 48000.0))
        frames = samples.reshape(-1, channels)
        with np.errstate(invalid="ignore"):
            for f in range(frames.shape[0]):
                left = np.float32(0.0)
                right = np.float32(0.0)
                for c in range(sc):
                    left = np.float32(left + frames[f, c] * matrix[c, 0])
                    right = np.float32(right + frames[f, c] * matrix[c, 1])
                exp = np.array([left, right], np.float32)
                a, e = lr[f].view(np.uint32), exp.view(np.uint32)
                for k in range(2):
                    if np.isnan(exp[k]):
                        assert np.isnan(lr[f, k])  # NaN payload propagation is platform-defined
                    else:
                        assert a[k] == e[k]


def test_running_means_sanitize_non_finite_values_without_poisoning_state(kat):
    # dsp.rs:626-635
    for prefix in ([np.nan], [np.nan, np.inf], [np.nan, np.inf, -np.inf]):
        assert kat.windowed_means([1], prefix)[0] == 0.0
    assert kat.windowed_means([1], [np.nan, np.inf, -np.inf, 1.0])[0] == 1.0


def test_running_means_preserve_small_values_after_a_large_value_expires(kat):
    # dsp.rs:637-656 (exact equality)
    assert kat.windowed_means([4], [1.0, 1.0e100, 1.0, -1.0e100])[0] == 0.5
    assert kat.windowed_means([2], [2.0 ** 53, 1.0, 1.0])[0] == 1.0
    assert kat.windowed_means([2], [1.0e100, 2.0, 1.0e-100, 1.0e-100])[0] == 1.0e-100


def test_rolling_mean_square_tracks_average(kat):
    # loudness/processor.rs:323-336
    eps = np.finfo(np.float64).eps
    m = kat.windowed_means([4, 2, 1, 4], [1.0, 9.0])
    assert abs(m[0] - 5.0) < eps
    m = kat.windowed_means([4, 2, 1, 4], [1.0, 9.0, 16.0, 25.0, 36.0])
    assert abs(m[0] - 21.5) < eps and abs(m[1] - 30.5) < eps and abs(m[2] - 36.0) < eps


def test_biquad_clear_matches_fresh_filter_state(kat):
    # dsp.rs:658-665
    used, _ = kat.biquad(False, 48000.0, 1000.0, [1.0, 0.25], clear_after=1)
    fresh, _ = kat.biquad(False, 48000.0, 1000.0, [0.25])
    assert used[1] == fresh[0]


def test_lr4_bands_sum_to_allpass(kat):
    # SURVEY §8c reading (6): with the dsp.rs:402-420 coefficients LP^2 + HP^2 is all-pass
    # (Linkwitz-Riley), checked at three frequencies around the 200 Hz split.
    n = 48000
    for freq in (60.0, 200.0, 1000.0):
        x = np.sin(2 * np.pi * freq * np.arange(n) / 48000.0).astype(np.float32)
        lp, _ = kat.biquad(False, 48000.0, 200.0, kat.biquad(False, 48000.0, 200.0, x)[0])
        hp, _ = kat.biquad(True, 48000.0, 200.0, kat.biquad(True, 48000.0, 200.0, x)[0])
        tail = slice(n // 2, n)
        ratio = np.sqrt(np.mean((lp + hp)[tail] ** 2)) / np.sqrt(np.mean(x[tail] ** 2))
        assert abs(ratio - 1.0) < 1e-3
    # and the LR4 three-band export produces three finite bands per lane
    bands = kat.threeband_lr4(48000.0, np.stack([x, -x], 1))
    assert bands.shape == (n, 3, 2) and np.isfinite(bands).all()
    assert np.allclose(bands[:, :, 0], -bands[:, :, 1])


def test_fft_windows_are_periodic(kat):
    # window.rs:115-122
    hann = kat.window(P.WINDOW_HANN, 8)
    assert hann[0] == 0.0
    assert abs(hann[4] - 1.0) < 1.0e-6
    assert abs(hann[7] - 0.1464465) < 1.0e-6


def test_power_conversion_preserves_deep_levels(kat):
    # level.rs:45-48
    assert abs(kat.power_to_db(1.0e-21, -300.0) + 210.0) < 1.0e-4


def test_sanitize_sample_rate(kat):
    # rate.rs:9-13
    assert kat.sanitize_sample_rate(float("nan")) == 48000.0
    assert kat.sanitize_sample_rate(-5.0) == 48000.0
    assert kat.sanitize_sample_rate(1e9) == 768000.0
    assert kat.sanitize_sample_rate(0.5) == 1.0


def test_oracle_fft_matches_f64_dft(kat):
    # The oracle's f32 FFT against its own f64 path and numpy's f64 FFT (definitional check,
    # rustfft's own rounding is unpinned).
    rng = np.random.default_rng(7)
    for n in (8, 64, 1024, 4096, 8192, 12):
        z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        ref = np.fft.fft(z.astype(np.complex128))
        got32 = kat.fft_f32(z)
        got64 = kat.fft_f64(z.astype(np.complex128))
        scale = np.abs(ref).max()
        assert np.abs(got64 - ref).max() <= 1e-12 * scale
        assert np.abs(got32 - ref).max() <= 2e-6 * scale
        inv = kat.fft_f32(got32, inverse=True) / n
        assert np.abs(inv - z).max() <= 2e-6 * np.abs(z).max()
