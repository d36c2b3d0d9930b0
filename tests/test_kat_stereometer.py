"""Known-answer tests ported from reference src/visuals/stereometer/processor.rs:230-256."""
import numpy as np

from openmeters_amd import capi
from openmeters_amd.capi import AudioBlock, StereometerConfig, StereometerProcessor
from oracle_kat import Kat


def test_snapshot_downsampling_preserves_stereo_pairs(backend):
    # :230-244
    p = StereometerProcessor(backend, StereometerConfig(sample_rate=4.0, segment_duration=1.0, target_sample_count=2))
    snap = p.process_block(AudioBlock([1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0], 2, 4.0))
    assert snap is not None
    assert snap.points[0].tolist() == [[1.0, 2.0], [5.0, 6.0]]


def test_correlator_matches_reference_points(oracle):
    # :246-256
    kat = Kat(oracle)
    close = lambda a, b: abs(a - b) <= 1e-6
    assert close(kat.correlation([(1.0, 1.0), (-1.0, -1.0)], 0.5), 1.0)
    assert close(kat.correlation([(1.0, -1.0), (-1.0, 1.0)], 0.5), -1.0)
    assert close(kat.correlation([(1.0, 0.25), (-1.0, -0.25)], 0.5), 1.0)
    assert close(kat.correlation([(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0)], 0.5), 0.0)
    assert close(kat.correlation([(0.0, 0.0)], 0.5), 0.0)
    assert abs(kat.ema_alpha(48000.0, 0.05) - 4.16580e-4) < 1e-8  # SURVEY §8c reading 6


def test_none_until_segment_is_full_then_band_correlations(backend):
    # :142-181: None until round(fs*segment)=960 frames; inverted stereo -> rho = -1 in every band
    cfg = StereometerConfig(analyze_bands=True, correlation_window=0.05, segment_duration=0.02,
                            target_sample_count=2000)
    p = StereometerProcessor(backend, cfg)
    n = 256
    t = np.arange(n * 8, dtype=np.float32)
    x = (0.5 * np.sin(2 * np.pi * 440.0 * t / 48000.0) + 0.2 * np.sin(2 * np.pi * 5000.0 * t / 48000.0)
         + 0.3 * np.sin(2 * np.pi * 80.0 * t / 48000.0)).astype(np.float32)
    outs = []
    for b in range(8):
        blk = np.stack([x[b * n:(b + 1) * n], -x[b * n:(b + 1) * n]], 1).reshape(-1)
        outs.append(p.process_block(AudioBlock(blk, 2, 48000.0)))
    assert [o is None for o in outs] == [True, True, True, False, False, False, False, False]
    last = outs[-1]
    assert last.points[0].shape == (960, 2)
    assert all(len(last.points[b]) == 0 for b in (1, 2, 3))  # emit_band_points is off
    assert np.all(last.correlations < -0.999)


def test_band_filters_sit_on_an_f32_noise_floor(oracle):
    """Not a reference test: the evidence behind the bars of the chunk-parallel stereometer form.  The reference evaluates its
    LR4 band split with f32 TDF-II biquads (src/dsp.rs:399-437); with poles at 200 Hz / 48 kHz every rounding error is amplified
    by ~fs / fc, so its band samples are 1e-5 ... 2e-5 (of a +-0.8 full scale) away from EXACT arithmetic on the cfg4 signals —
    measured here on the oracle against scipy's f64 `lfilter` with the same f32 coefficients.  Any other evaluation order of the
    same f32 arithmetic (the chunk-parallel kernels) differs from the reference's by the same order: hence |d point| <= 1e-4,
    not 1e-6, for that form (the sequential kernels keep the reference's order and stay bit-exact)."""
    from scipy.signal import lfilter
    from golden_inputs import cfg4_pcm
    FS = 48000.0
    f32 = np.float32

    def biquad(highpass, f):     # Biquad::new (dsp.rs:402-420) in f32
        ratio = np.clip(f32(f) / f32(FS), f32(1e-6), f32(0.49))
        ang = f32(6.2831855) * ratio
        sin, cos = f32(np.sin(ang)), f32(np.cos(ang))
        alpha = sin * f32(0.70710677)
        gain, sign = (f32(1) + cos, f32(-1)) if highpass else (f32(1) - cos, f32(1))
        inv = f32(1) / (f32(1) + alpha)
        return (np.array([gain * f32(0.5) * inv, gain * inv * sign, gain * f32(0.5) * inv], np.float64),
                np.array([1.0, f32(-2) * cos * inv, (f32(1) - alpha) * inv], np.float64))

    cfg = StereometerConfig(analyze_bands=True, emit_band_points=True, correlation_window=0.05, segment_duration=0.02, target_sample_count=960)
    worst = 0.0
    for s in (30, 32):
        n = 256 * 40
        pcm = cfg4_pcm(s, n)
        p = StereometerProcessor(oracle, cfg)
        for k in range(0, n, 256):
            w = p.process_block(AudioBlock(pcm[k:k + 256].reshape(-1), 2, FS))
        hp_lo, lp_hi = biquad(True, 200.0), biquad(False, 2000.0)
        x = pcm.astype(np.float64)
        above = lfilter(*hp_lo, lfilter(*hp_lo, x, axis=0), axis=0)
        mid = lfilter(*lp_hi, lfilter(*lp_hi, above, axis=0), axis=0)
        got = w.points[2].astype(np.float64) / 0.8
        err = float(np.abs(got - mid[-960:]).max())
        assert 3e-6 < err < 1e-4, err      # 1.7e-5 / 2.1e-5 measured: the reference's own distance from exact arithmetic
        worst = max(worst, err)
    assert worst > 1e-5
