"""Reassigned-splat accumulation + dB resolve (reference spectrogram.wgsl:66-76, 126-147, 215-237).
The reference has no test for its shaders (PARITY UNPINNED, see oracle/splat.hpp): these cases pin the restated behaviour —
pixel-centre coverage, the column-age / time-offset mapping, cull rules, tilt, frequency scales, the resolve — on the CPU
oracle, and on the HIP product with `-m gpu`."""
import numpy as np
import pytest

from openmeters_amd import capi


def one(api, points, view, scale=1.0, ages=1):
    cols = [np.zeros((0, 3), np.float32)] * (ages - 1) + [np.array(points, np.float32).reshape(-1, 3)]
    return capi.spectrogram_splat(api, cols, view, scale)


def test_single_point_lands_in_one_pixel_and_resolves_to_db(backend):
    v = capi.splat_view(backend, 8.0, 10.0, freq_scale=capi.FREQ_SCALE_LINEAR)
    assert (v.width, v.height) == (8, 10) and v.freq_max == 24000.0 and v.freq_min == 1.0
    # newest column (age 0), time_offset -0.5 -> x = 8 - 0.5 = 7.5 -> pixel 7; 12 kHz -> y = (1 - 0.49998) * 10 -> pixel 5
    acc, db = one(backend, [(-0.5, 12000.0, 0.25)], v, scale=0.5)
    assert acc[5, 7] == 0.25 and acc.sum() == 0.25
    assert abs(db[5, 7] - 10.0 * np.log10(0.125)) < 1e-4
    assert np.isneginf(db[0, 0]) and np.isneginf(np.delete(db.ravel(), 5 * 8 + 7)).all()


def test_age_and_time_offset_move_along_time_axis_and_points_accumulate(backend):
    v = capi.splat_view(backend, 16.0, 4.0, freq_scale=capi.FREQ_SCALE_LINEAR)
    older = np.array([(-0.5, 6000.0, 1.0), (-2.5, 6000.0, 2.0)], np.float32)     # age 3: x = 16 - 3.5 = 12.5, 16 - 5.5 = 10.5
    newest = np.array([(-3.5, 6000.0, 4.0), (-0.5, 6000.0, 8.0)], np.float32)    # age 0: x = 12.5 (same pixel), 15.5
    cols = [older, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), newest]
    acc, _ = capi.spectrogram_splat(backend, cols, v, 1.0)
    row = acc[3]                                                                   # 6 kHz of 24 kHz: y = 0.75 * 4 = 3.0 -> pixel 3
    assert row[12] == 5.0 and row[10] == 2.0 and row[15] == 8.0 and acc.sum() == 15.0


def test_scale_factor_two_covers_a_two_by_two_block_and_edges_clip(backend):
    v = capi.splat_view(backend, 8.0, 8.0, scale_factor=2.0, freq_scale=capi.FREQ_SCALE_LINEAR)
    acc, _ = one(backend, [(-1.0, 12000.0, 1.0)], v)           # x = 8 - 2 = 6 -> [5, 7): pixels 5, 6; y = 4.0 -> pixels 3, 4
    assert acc[3:5, 5:7].tolist() == [[1.0, 1.0], [1.0, 1.0]] and acc.sum() == 4.0
    acc, _ = one(backend, [(0.0, 12000.0, 1.0)], v)            # x = 8 -> [7, 9): only pixel 7 is inside the target
    assert acc[3:5, 7].tolist() == [1.0, 1.0] and acc.sum() == 2.0
    acc, _ = one(backend, [(-0.25, 12000.0, 1.0)], v)          # x = 7.5 -> centres 6.5 and 7.5 in [6.5, 8.5): top-left rule keeps 6
    assert acc[3:5, 6:8].tolist() == [[1.0, 1.0], [1.0, 1.0]]


def test_cull_rules(backend):
    v = capi.splat_view(backend, 4.0, 4.0, freq_scale=capi.FREQ_SCALE_LINEAR, uv=(0.25, 0.75))
    pts = [(-0.5, 12000.0, 0.0), (-0.5, 12000.0, -1.0), (-0.5, 12000.0, np.nan),   # !(power > 0)
           (-0.5, 2000.0, 1.0), (-0.5, 22000.0, 1.0),                                # outside the zoom window by > 1 %
           (-0.5, 12000.0, 3.0)]
    acc, db = one(backend, pts, v)
    assert acc.sum() == 3.0 and acc[2, 3] == 3.0                                     # zoomed = 0.5 -> y = 2.0 -> pixel 2
    assert np.isfinite(db).sum() == 1


def test_tilt_scales_power_around_one_kilohertz_and_skips_floor_bins(backend):
    v = capi.splat_view(backend, 4.0, 64.0, freq_scale=capi.FREQ_SCALE_LOGARITHMIC, tilt_db=3.0)
    acc, _ = one(backend, [(-0.5, 1000.0, 1.0), (-1.5, 4000.0, 1.0), (-2.5, 250.0, 1.0), (-3.5, 8000.0, 1e-14)], v)
    got = [acc[:, x].sum() for x in (3, 2, 1, 0)]
    assert abs(got[0] - 1.0) < 1e-6 and abs(got[1] - 10 ** 0.6) < 1e-4 and abs(got[2] - 10 ** -0.6) < 1e-5 and got[3] == 0.0


@pytest.mark.parametrize("scale", [capi.FREQ_SCALE_LINEAR, capi.FREQ_SCALE_LOGARITHMIC, capi.FREQ_SCALE_ERB])
def test_frequency_scales_follow_pos_of(backend, scale):
    """row of a tone = (1 - FrequencyScale::pos_of(min, max, f)) * height (util/audio/frequency.rs:20-31)"""
    H = 200.0
    v = capi.splat_view(backend, 2.0, H, freq_scale=scale)
    fn = {capi.FREQ_SCALE_LINEAR: lambda f: f, capi.FREQ_SCALE_LOGARITHMIC: lambda f: np.arcsinh(f / 20.0),
          capi.FREQ_SCALE_ERB: lambda f: 21.4 * np.log10(1.0 + f / 228.8)}[scale]
    for f in (30.0, 440.0, 1000.0, 9000.0, 23000.0):
        acc, _ = one(backend, [(-0.5, f, 1.0)], v)
        pos = (fn(f) - fn(1.0)) / (fn(24000.0) - fn(1.0))
        rows = np.flatnonzero(acc[:, 1])
        assert len(rows) == 1 and abs(rows[0] + 0.5 - (1.0 - pos) * H) <= 0.5 + 1e-3


def test_resolve_floor_and_many_points(backend):
    v = capi.splat_view(backend, 4.0, 4.0, freq_scale=capi.FREQ_SCALE_LINEAR)
    pts = [(-0.5, 12000.0, 1e-18)] + [(-1.5, 12000.0, 0.001)] * 1000
    acc, db = one(backend, pts, v, scale=2.0 / 3.0)
    assert db[2, 3] == -140.0                                   # 6.7e-19 -> -181.7 dB, clamped to the analysis floor
    assert abs(acc[2, 2] - 1.0) < 1e-4 and abs(db[2, 2] - 10.0 * np.log10(2.0 / 3.0)) < 1e-3
