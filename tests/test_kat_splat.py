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


# ---- the column history ring (reference src/visuals/spectrogram/state.rs:53-175; its tests :803-861 ported) ---------------
def _update(history_length, reset, columns, kind, ppc=2, scale=0.25):
    return capi.SpectrogramUpdate(fft_size=(ppc - 1) * 2, hop_size=1, sample_rate=48000.0, history_length=history_length, reset=reset,
                                  reassigned_power_scale=scale, kind=kind, new_columns=columns)


def classic_update(history_length, reset, values):        # state.rs:771-775
    return _update(history_length, reset, [np.full(2, int(v), np.uint16) for v in values], capi.COLUMN_CLASSIC)


def reassigned_update(history_length, reset, counts, ppc=3):   # state.rs:777-786
    point = np.array([0.0, 100.0, 0.01], np.float32)
    return _update(history_length, reset, [np.tile(point, (c, 1)) for c in counts], capi.COLUMN_REASSIGNED, ppc=ppc)


def ring_values(h):
    """the classic test values as they sit in the slots"""
    i = h.info()
    return [int(h.fetch_slot(s)[0]) for s in range(i.ring_capacity)]


def test_resize_copy_plans_preserve_visible_columns(backend):
    # state.rs:810-834: the copy plans [2, 3, 0, 1] and [MAX, MAX, 0, 1] are observed through the slots' contents
    h = capi.SpectrogramHistory(backend)
    h.apply(classic_update(4, True, [10, 11, 12, 13]))
    assert ring_values(h) == [10, 11, 12, 13]
    h.apply(classic_update(4, False, [14, 15]))                     # uploads to slots 0, 1
    i = h.info()
    assert (i.ring_capacity, i.col_count, i.write_slot) == (4, 4, 2) and ring_values(h) == [14, 15, 12, 13]
    h.apply(classic_update(6, False, [16]))                         # grow while full: remap_retained(write_slot, col_count)
    i = h.info()
    assert (i.ring_capacity, i.col_count, i.write_slot) == (6, 5, 5)
    assert ring_values(h)[:5] == [12, 13, 14, 15, 16]               # plan [2, 3, 0, 1], then the new column in slot 4
    assert (i.newest_slot, i.visible_slots) == (4, 5)

    h = capi.SpectrogramHistory(backend)
    h.apply(classic_update(4, True, [10, 11, 12, 13]))
    h.apply(classic_update(2, False, [14]))                         # shrink: keep the newest two, then upload to slot 0
    i = h.info()
    assert (i.ring_capacity, i.col_count, i.write_slot) == (2, 2, 1)
    assert ring_values(h) == [14, 13]                               # plan [MAX, MAX, 0, 1] -> [12, 13], slot 0 overwritten


def test_reassigned_params_track_sparse_slot_counts(backend):
    # state.rs:836-861
    h = capi.SpectrogramHistory(backend)
    h.apply(reassigned_update(4, True, [0, 2, 1]))
    i = h.info()
    assert i.reassigned_points_per_slot == 2 and i.kind == capi.COLUMN_REASSIGNED
    assert h.slot_counts().tolist() == [0, 2, 1, 0]
    assert [len(h.fetch_slot(s)) for s in range(4)] == [0, 2, 1, 0]
    # hysteresis of fit_reassigned_slot_capacity (state.rs:131-148): shrinks only when more than 4x oversized
    h.apply(reassigned_update(4, False, [1, 1, 1, 1], ppc=3))
    assert h.info().reassigned_points_per_slot == 2
    h.apply(reassigned_update(4, True, [3], ppc=20))
    h.apply(reassigned_update(4, False, [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 13], ppc=20))
    assert h.info().reassigned_points_per_slot == 13 and h.slot_counts().tolist() == [13, 0, 0, 0]   # column 27 lands in slot (1 + 27) % 4


def test_history_ring_age_arithmetic_matches_time_ordered_splat(backend):
    """age = (newest_col + hl - slot) % hl (spectrogram.wgsl:141-142): the image splatted from the ring equals the image of
    the same columns handed over oldest -> newest, through wrap-around, growth and shrink of the ring."""
    rng = np.random.default_rng(7)
    view = capi.splat_view(backend, 24.0, 32.0, freq_scale=capi.FREQ_SCALE_LINEAR)
    ppc = 9
    h = capi.SpectrogramHistory(backend)
    kept = []                                                       # the columns a time-ordered consumer would still hold

    def column():
        n = int(rng.integers(0, ppc + 1))
        pts = np.zeros((n, 3), np.float32)
        pts[:, 0] = rng.uniform(-3.0, 0.0, n)
        pts[:, 1] = np.sort(rng.uniform(100.0, 23000.0, n))
        pts[:, 2] = rng.uniform(0.1, 1.0, n)
        return pts

    for step, (hist, n_new, reset) in enumerate([(6, 4, True), (6, 5, False), (10, 3, False), (10, 9, False), (4, 2, False), (4, 0, False),
                                                 (4, 11, False), (7, 1, False), (7, 3, True), (3, 1, False)]):
        cols = [column() for _ in range(n_new)]
        h.apply(_update(hist, reset, cols, capi.COLUMN_REASSIGNED, ppc=ppc, scale=1.0))
        if reset:
            kept = []
        kept = (kept + cols)[-hist:]
        i = h.info()
        assert i.ring_capacity == hist and i.visible_slots == min(len(kept), hist) == i.col_count, (step, i.col_count, len(kept))
        acc, db = h.splat(view, 1.0)
        want_acc, want_db = capi.spectrogram_splat(backend, kept, view, 1.0) if kept else (np.zeros_like(acc[0]), None)
        assert np.array_equal(acc[0], want_acc), step
        if want_db is not None:
            assert np.array_equal(db[0], want_db, equal_nan=True), step
        # slot -> age bookkeeping: the newest column sits in newest_slot
        if kept:
            assert np.array_equal(h.fetch_slot(i.newest_slot), kept[-1]), step
