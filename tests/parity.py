"""Column-level parity metrics between the HIP product and the CPU oracle (test infra).

Reassigned columns are variable-length, ascending-bin lists of (time_offset, freq_hz, power) with no
bin index, and a bin sitting on the 1e-14 analysis floor or on the 0 < f < fs/2 edge may appear in
one list only.  `align_points` walks both lists like a merge: points whose power agrees are paired;
anything else is an orphan and must be weak (checked by the caller).  All errors are normalised the
way SURVEY §7 defines "1e-5 relative": against the column's maximum, not per-bin.
"""
import numpy as np


def align_points(a, b, max_power):
    """Returns (pairs, orphans_a, orphans_b); pairs index into a / b."""
    i = j = 0
    pairs, oa, ob = [], [], []
    tol_abs = 1e-5 * max_power

    def shifted_pairing_is_clearly_better(i, j):
        """Weak points all 'agree' in power, so a floor-level orphan can slip the merge by one bin without being noticed
        (bins are 3 Hz apart at 16384 points).  If skipping one entry on either side brings the frequencies at least 4x
        closer, the current pair is a slip."""
        d0 = abs(a[i, 1] - b[j, 1])
        if d0 == 0.0 or (len(a) - i) == (len(b) - j):   # nothing left to re-synchronise: the rest pairs one to one
            return None
        if len(b) - j > len(a) - i and j + 1 < len(b) and abs(a[i, 1] - b[j + 1, 1]) < 0.25 * d0 and abs(a[i, 2] - b[j + 1, 2]) <= tol_abs + 2e-3 * min(a[i, 2], b[j + 1, 2]):
            return (0, 1)
        if len(a) - i > len(b) - j and i + 1 < len(a) and abs(a[i + 1, 1] - b[j, 1]) < 0.25 * d0 and abs(a[i + 1, 2] - b[j, 2]) <= tol_abs + 2e-3 * min(a[i + 1, 2], b[j, 2]):
            return (1, 0)
        return None

    while i < len(a) and j < len(b):
        pa, pb = a[i, 2], b[j, 2]
        if abs(pa - pb) <= tol_abs + 2e-3 * min(pa, pb) and abs(a[i, 1] - b[j, 1]) <= max(50.0, 0.02 * abs(b[j, 1])):
            slip = shifted_pairing_is_clearly_better(i, j) if max(pa, pb) < 1e-6 * max_power else None
            if slip is None:
                pairs.append((i, j))
                i += 1
                j += 1
            else:
                oa.extend(range(i, i + slip[0]))
                ob.extend(range(j, j + slip[1]))
                i += slip[0]
                j += slip[1]
            continue
        # look ahead a few entries for a re-synchronisation point
        found = None
        for da in range(0, 4):
            for db in range(0, 4):
                if da == db == 0 or i + da >= len(a) or j + db >= len(b):
                    continue
                qa, qb = a[i + da, 2], b[j + db, 2]
                if abs(qa - qb) <= tol_abs + 2e-3 * min(qa, qb) and abs(a[i + da, 1] - b[j + db, 1]) <= max(
                        50.0, 0.02 * abs(b[j + db, 1])):
                    found = (da, db)
                    break
            if found:
                break
        if not found:
            found = (1, 1)
        oa.extend(range(i, i + found[0]))
        ob.extend(range(j, j + found[1]))
        i += found[0]
        j += found[1]
    oa.extend(range(i, len(a)))
    ob.extend(range(j, len(b)))
    return pairs, oa, ob


def reassigned_column_metrics(hip, ora, sample_rate, hop):
    """Normalised error metrics for one column.
      power : max |dP| / max P
      freq  : max |df| * r / (fs/2),  r = sqrt(P / max P)   (amplitude-weighted: the reassignment ratio
              d*conj(b)/|b|^2 amplifies the f32 FFT noise floor by 1/r on weak bins)
      time  : max |dt| * r   (hops)
      orphan: max P(orphan) / max P
    """
    max_power = float(max(ora[:, 2].max() if len(ora) else 0.0, hip[:, 2].max() if len(hip) else 0.0))
    if max_power <= 0.0:
        return dict(power=0.0, freq=0.0, time=0.0, orphan=0.0, n=0, orphans=len(hip) + len(ora))
    pairs, oa, ob = align_points(hip, ora, max_power)
    pa = np.array([p[0] for p in pairs], int)
    pb = np.array([p[1] for p in pairs], int)
    m = dict(n=len(pairs), orphans=len(oa) + len(ob))
    if len(pairs):
        h, o = hip[pa].astype(np.float64), ora[pb].astype(np.float64)
        r = np.sqrt(o[:, 2] / max_power)
        m["power"] = float(np.abs(h[:, 2] - o[:, 2]).max() / max_power)
        m["freq"] = float((np.abs(h[:, 1] - o[:, 1]) * r).max() / (sample_rate * 0.5))
        m["time"] = float((np.abs(h[:, 0] - o[:, 0]) * r).max())
    else:
        m.update(power=0.0, freq=0.0, time=0.0)
    orphan_p = [hip[i, 2] for i in oa] + [ora[j, 2] for j in ob]
    m["orphan"] = float(max(orphan_p) / max_power) if orphan_p else 0.0
    return m


def classic_column_metrics(hip, ora):
    """u16 dB codes (code = (dB + 144) * 65535 / 156, one code = 0.0024 dB).
    max_code_diff / n_diff: over every bin.  An f32 FFT carries a noise floor of ~1e-7 of the column's largest amplitude,
    so bins far below the maximum legitimately differ between two correct FFTs (the generic kernel repeats the oracle's
    radix-2 order and is code-identical; the fused radix-16 kernels are not).  Hence also:
    power: max |dP| / max P in linear power;  loud_code_diff: max |d code| over the bins within 40 dB of the column maximum
    (one code at -40 dB is 5.5e-8 of the maximum: the weak-bin bound of check_classic, so the two criteria meet there)."""
    d = np.abs(hip.astype(np.int64) - ora.astype(np.int64))
    db_h, db_o = hip.astype(np.float64) * (156.0 / 65535.0) - 144.0, ora.astype(np.float64) * (156.0 / 65535.0) - 144.0
    p_h, p_o = 10.0 ** (db_h / 10.0), 10.0 ** (db_o / 10.0)
    loud = db_o >= db_o.max() - 40.0 if len(d) else np.zeros(0, bool)
    weak = ~loud
    return dict(max_code_diff=int(d.max()) if len(d) else 0, n_diff=int((d > 0).sum()), n=len(d),
                power=float(np.abs(p_h - p_o).max() / max(p_o.max(), 1e-300)) if len(d) else 0.0,
                weak_power=float(np.abs(p_h - p_o)[weak].max() / max(p_o.max(), 1e-300)) if weak.any() else 0.0,
                loud_code_diff=int(d[loud].max()) if loud.any() else 0)


def check_classic(got, want):
    """fused-kernel bar: codes within 1 for every bin within 40 dB of the column maximum; below that, linear power within
    6e-8 of the column maximum (= one code at -40 dB, where the two criteria meet) (an f32 FFT's own noise floor: the codes of bins 100 dB down are not reproducible between two
    correct transforms, and the two-columns-per-FFT packing lets each column see the other's rounding noise)"""
    assert len(got) == len(want)
    for h, o in zip(got, want):
        m = classic_column_metrics(h, o)
        assert m["loud_code_diff"] <= 1 and m["weak_power"] <= 6e-8, m   # 6e-8 = one code at -40 dB
