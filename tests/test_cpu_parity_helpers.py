"""The column-alignment helper of the GPU parity tests is test infrastructure that every reassigned-column verdict rests on:
pin its behaviour on synthetic columns (no GPU)."""
import numpy as np

from parity import align_points, reassigned_column_metrics


def column(rng, n):
    power = 10.0 ** rng.uniform(-14.0, -2.0, n)
    power[n // 3] = 1.4e-2
    return np.stack([rng.normal(size=n), np.sort(rng.uniform(10.0, 23990.0, n)), power], 1).astype(np.float32)


def test_alignment_survives_floor_level_orphans_next_to_a_decaying_skirt():
    """The case a greedy merge got wrong (found by smoke()): one side carries an extra floor-level point just before a run of
    points whose powers halve from one to the next — pairing the run shifted by one keeps every |dP| under a 1e-5 tolerance."""
    rng = np.random.default_rng(3)
    ora = column(rng, 400)
    k = 200
    ora[k:k + 12, 2] = (2.6e-7 * 0.5 ** np.arange(12)).astype(np.float32)       # skirt: each point half of the previous one
    hip = np.insert(ora, k, np.array([0.1, ora[k, 1] - 1.0, 4.6e-12], np.float32), axis=0)  # extra floor-level point on one side
    hip[:, 2] *= (1.0 + 1e-7 * rng.normal(size=len(hip))).astype(np.float32)
    pairs, oa, ob = align_points(hip, ora, float(ora[:, 2].max()))
    assert oa == [k] and ob == [] and len(pairs) == len(ora)
    assert all(i == j if j < k else i == j + 1 for i, j in pairs)
    m = reassigned_column_metrics(hip, ora, 48000.0, 256)
    assert m["power"] < 1e-6 and m["orphan"] < 1e-9 and m["orphans"] == 1


def test_alignment_keeps_floor_level_points_with_noisy_frequencies_paired():
    rng = np.random.default_rng(4)
    ora = column(rng, 600)
    hip = ora.copy()
    weak = ora[:, 2] < 1e-10
    hip[weak, 1] += rng.normal(scale=40.0, size=int(weak.sum())).astype(np.float32)   # f-hat of a floor-level bin is noise
    pairs, oa, ob = align_points(hip, ora, float(ora[:, 2].max()))
    assert len(pairs) == len(ora) and not oa and not ob


def test_alignment_of_empty_and_one_sided_columns():
    rng = np.random.default_rng(5)
    c = column(rng, 5)
    assert align_points(c[:0], c, 1.0) == ([], [], [0, 1, 2, 3, 4])
    assert align_points(c, c[:0], 1.0) == ([], [0, 1, 2, 3, 4], [])
    pairs, oa, ob = align_points(c, c[1:], float(c[:, 2].max()))
    assert oa == [0] and not ob and pairs == [(i + 1, i) for i in range(4)]


def test_alignment_gaps_the_strong_point_kept_by_one_side_only():
    """soak seed 16016002: the last bins below fs/2 of a quiet column; one side keeps a point at fs/2 - 0.07 Hz (the other side's
    f-hat sits on the far side of the keep test) next to a neighbour 38 dB weaker whose power differs by parity-level rounding.  The
    power costs of "gap the strong point" and "pair it with the weak neighbour, gap that one" tie by construction."""
    rng = np.random.default_rng(6)
    ora = column(rng, 300)
    top = float(ora[:, 2].max())
    ora[-3:] = np.array([[-58.4, 22016.7, 1.4144e-11 * top / 1.137e-6], [-76.5, 22039.8, 1.1000e-11 * top / 1.137e-6],
                         [-32.46, 22046.7, 2.126e-10 * top / 1.137e-6]], np.float32)
    extra = np.array([-38.6, 22049.93, 4.147e-10 * top / 1.137e-6], np.float32)
    for sign in (1.0, -1.0):
        hip = ora.copy()
        hip[-2, 2] *= np.float32(1.0 + sign * 1.5e-3)     # the weak neighbour, off by what a quiet column's bars allow
        with_extra = np.insert(ora, len(ora) - 1, extra, axis=0)
        pairs, oa, ob = align_points(hip, with_extra, top)
        assert oa == [] and ob == [len(ora) - 1], (sign, oa, ob)
        assert all(i == j if j < len(ora) - 1 else i == j - 1 for i, j in pairs)
        pairs, oa, ob = align_points(with_extra, hip, top)
        assert ob == [] and oa == [len(ora) - 1], (sign, oa, ob)
