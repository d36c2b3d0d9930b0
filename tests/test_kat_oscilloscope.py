"""Known-answer tests ported from reference src/visuals/oscilloscope/processor.rs:957-1245."""
import numpy as np
import pytest

from openmeters_amd import capi
from openmeters_amd.capi import AudioBlock, OscilloscopeConfig, OscilloscopeProcessor
from oracle_kat import Kat
from signals import TAU, F32, fract, noise_samples, periodic_samples, sine_samples

RATE = 48000.0
BLOCK = 1024


def stable_config(**kw):
    base = dict(sample_rate=RATE, segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2)
    base.update(kw)
    return OscilloscopeConfig(**base)


def feed_blocks(p, signal, blocks, predicate=None):
    """first_block_where (:844-857): returns the block offset at which predicate first holds."""
    start = blocks.start
    for b in blocks:
        off = b * BLOCK
        p.process_block(AudioBlock(signal[off:off + BLOCK], 1, RATE))
        if predicate is not None and predicate(p):
            return b - start
    return None


def cycle_rate_switch_signal(frm, to, warmup, after):
    # :871-885
    switch = warmup * BLOCK
    n = np.arange(BLOCK * (warmup + after), dtype=np.float32)
    t = (n / F32(RATE)).astype(np.float32)
    t0 = F32(switch) / F32(RATE)
    phase0 = TAU * F32(frm) * t0
    a = np.sin((TAU * F32(frm) * t).astype(np.float32))
    b = np.sin((phase0 + TAU * F32(to) * (t - t0)).astype(np.float32))
    return np.where(np.arange(n.size) < switch, a, b).astype(np.float32)


def delayed_sine(freq, silence, signal_blocks):
    # :887-898
    onset = silence * BLOCK
    n = np.arange(BLOCK * (silence + signal_blocks))
    k = np.maximum(n - onset, 0).astype(np.float32)
    s = np.sin((TAU * F32(freq) * k / F32(RATE)).astype(np.float32))
    return np.where(n >= onset, s, 0.0).astype(np.float32)


@pytest.fixture(scope="module")
def kat(oracle):
    return Kat(oracle)


def test_period_estimation(kat):
    # :957-995
    long = int(RATE * 0.1)
    for freq, frames, max_error in [(41.0, long, 0.02), (110.0, long, 0.02), (440.0, long, 0.02), (1000.0, long, 0.02),
                                    (4000.0, long, 0.02), (8000.0, long, 0.02), (1000.0, 256, 0.03)]:
        est = kat.estimate_period(sine_samples(freq, RATE, frames), RATE)
        assert est is not None, "period"
        detected = RATE / est[0]
        assert abs(detected - freq) / freq < max_error, f"got {detected}Hz, expected {freq}Hz"
        assert est[1] > 0.9
    for freq, samples in [
        (110.0, periodic_samples(110.0, RATE, long, lambda c: F32(2.0) * fract(c) - F32(1.0))),
        (440.0, periodic_samples(440.0, RATE, long, lambda c: np.where(fract(c) < 0.5, 1.0, -1.0))),
        (440.0, periodic_samples(440.0, RATE, long, lambda c: np.sin(TAU * c) + F32(2.0) * np.sin(TAU * F32(2.0) * c))),
    ]:
        est = kat.estimate_period(samples, RATE)
        assert est is not None
        assert abs(RATE / est[0] - freq) / freq < 0.03
        assert est[1] >= 0.5
    assert kat.estimate_period(noise_samples(long), RATE) is None


def test_stable_trigger_limits_phase_jitter(kat):
    # :997-1019 via stable_phase_jitter (:933-955)
    frames = BLOCK * 60
    signals = {
        "sine": sine_samples(440.0, RATE, frames),
        "biased_am": periodic_samples(440.0, RATE, frames, lambda c: (F32(0.6) + F32(0.4) * np.sin(TAU * c / F32(37.0)))
                                      * np.sin(TAU * c) + F32(0.25)),
        "saw": periodic_samples(440.0, RATE, frames, lambda c: F32(2.0) * fract(c) - F32(1.0)),
        "square": periodic_samples(440.0, RATE, frames, lambda c: np.where(fract(c) < 0.5, 1.0, -1.0)),
    }
    period = RATE / 440.0
    for name, sig in signals.items():
        pos, locked = kat.stable_trigger_positions(sig, BLOCK, 60, RATE)
        first, jitter = None, 0.0
        for b in range(20, 60):
            if locked[b]:
                if first is None:
                    first = pos[b]
                d = (pos[b] - first + period * 0.5) % period - period * 0.5
                jitter = max(jitter, abs(d))
        assert first is not None
        assert jitter < 3.0, f"{name} jitter was {jitter:.3f} samples"


def test_stable_trigger_retunes_reference_around_center(kat):
    # :1021-1042
    ref = np.zeros(17, np.float32)
    ref[8], ref[10] = 0.25, 1.0
    out = kat.retune_reference(ref, 4.0, 8.0, 17)
    assert int(np.argmax(out)) == 12
    assert abs(out[8] - 0.25) < np.finfo(np.float32).eps


def test_stable_template_rebuild_discards_rejected_candidate(kat):
    # :1044-1060
    period = 8.0
    edge = kat.prepare_template(17, period)
    rejected = np.sin(np.arange(17, dtype=np.float32) * np.float32(0.7)).astype(np.float32)
    _, cand = kat.write_candidate(np.zeros(17, np.float32), rejected, period)
    assert not np.array_equal(cand, edge)
    assert np.array_equal(kat.prepare_template(17, period), edge)
    assert np.allclose(edge[:8], -edge[::-1][:8]) and edge[0] < 0 < edge[-1]


def test_stable_correlation_is_shape_based(kat):
    # :1062-1081
    for work in ([1.0, -1.0, 1.0, -1.0, 10.0, -10.0, 0.0, 0.0], [11.0, 9.0, 11.0, 9.0, 1.0, -1.0, 0.0, 0.0]):
        assert kat.find_best([1.0, -1.0, 1.0, -1.0], work, 4, 16.0)[0] == 0
    v, _ = kat.write_candidate([11.0, 9.0, 11.0, 9.0], [1.0, -1.0, 1.0, -1.0], 1000.0)
    assert v > 0.99


def test_zero_crossing_finds_edges_after_channel_projection(kat, oracle):
    # :1083-1110
    mono = sine_samples(440.0, RATE, 4800)
    for c in (kat.find_rising_zero_crossing(mono, 0, 3840, True), kat.find_rising_zero_crossing(mono, 0, 4799, False)):
        assert c is not None and mono[c] > 0.0 and mono[c - 1] <= 0.0
    same = np.stack([mono, mono], 1).reshape(-1)
    lr, _, _ = kat.stereo_frames(AudioBlock(same, 2, RATE))
    mid = ((lr[:, 0] + lr[:, 1]) * np.float32(0.5)).astype(np.float32)
    c = kat.find_rising_zero_crossing(mid, 0, 3840, True)
    assert mid[c] > 0.0 and mid[c - 1] <= 0.0
    inv = np.stack([mono, -mono], 1).reshape(-1)
    lr, _, _ = kat.stereo_frames(AudioBlock(inv, 2, RATE))
    mid = ((lr[:, 0] + lr[:, 1]) * np.float32(0.5)).astype(np.float32)
    assert kat.find_rising_zero_crossing(mid, 0, 4799, False) is None
    assert kat.find_rising_zero_crossing(lr[:, 0].copy(), 0, 4799, False) is not None


def test_zero_crossing_both_edges_near_zero(backend):
    # :1112-1138
    cfg = OscilloscopeConfig(segment_duration=0.01, trigger_mode=capi.TRIGGER_ZERO_CROSSING,
                             channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT)
    p = OscilloscopeProcessor(backend, cfg)
    mono = sine_samples(440.0, cfg.sample_rate, int(cfg.sample_rate * 0.1))
    snap = p.process_block(AudioBlock(np.stack([mono, mono], 1).reshape(-1), 2, cfg.sample_rate))
    assert snap is not None
    assert snap.channels == 2
    n = snap.samples_per_channel
    assert 0 < n <= 4096 and len(snap.samples) == n * 2
    assert 0.0 < snap.samples[0] < 0.15, "left edge"
    assert abs(snap.samples[n - 1]) < 0.15, "right edge"


def test_input_channel_count_change_resets_history_and_trigger_lock(backend):
    # :1140-1152
    p = OscilloscopeProcessor(backend, stable_config())
    signal = sine_samples(440.0, RATE, BLOCK * 20)
    feed_blocks(p, signal, range(0, 20))
    assert p.last_cycle_rate() is not None
    p.process_block(AudioBlock(np.zeros(BLOCK * 2, np.float32), 2, RATE))
    assert p.last_cycle_rate() is None


def test_stable_lock_has_bounded_aperiodic_holdover(backend):
    # :1154-1177
    warmup, noise = 20, 20
    signal = np.concatenate([sine_samples(440.0, RATE, BLOCK * warmup), noise_samples(BLOCK * noise)])
    p = OscilloscopeProcessor(backend, stable_config())
    feed_blocks(p, signal, range(0, warmup))
    assert p.last_cycle_rate() is not None
    ns = warmup * BLOCK
    p.process_block(AudioBlock(signal[ns:ns + BLOCK], 1, RATE))
    assert p.last_cycle_rate() is not None, "brief aperiodic input should hold lock"
    released = feed_blocks(p, signal, range(warmup + 1, warmup + noise), lambda q: q.last_cycle_rate() is None)
    assert released is not None, "sustained aperiodic input should release lock"
    assert released <= 8, f"release took {released} blocks"


def two_channel_correlation(snap):
    assert snap.channels == 2
    n = snap.samples_per_channel
    assert len(snap.samples) == n * 2
    a, b = snap.samples[:n], snap.samples[n:]
    return float(np.dot(a, b) / np.sqrt(np.dot(a, a) * np.dot(b, b)))


def test_fixed_trigger_source_preserves_visible_channel_phase(backend):
    # :1179-1193 via inverted_stereo_capture (:910-927)
    cfg = stable_config(trigger_source=capi.CH_LEFT, channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT)
    p = OscilloscopeProcessor(backend, cfg)
    mono = sine_samples(440.0, RATE, BLOCK * 20)
    stereo = np.stack([mono, -mono], 1).reshape(-1)
    snap = None
    for b in range(20):
        off = b * BLOCK * 2
        snap = p.process_block(AudioBlock(stereo[off:off + BLOCK * 2], 2, RATE))
    detected = p.last_cycle_rate()
    assert detected is not None, "trigger should lock"
    assert abs(detected - 440.0) < 20.0
    corr = two_channel_correlation(snap)
    assert corr < -0.9, f"linked trigger should preserve inverted stereo phase, got {corr}"
    # cfg4 shape: span = 2 periods -> samples_per_channel = round(2*48000/440)+1
    assert snap.samples_per_channel == 219


def test_lock_acquisition_and_cycle_rate_transitions(backend):
    # :1195-1245
    p = OscilloscopeProcessor(backend, stable_config())
    signal = sine_samples(440.0, RATE, BLOCK * 20)
    took = feed_blocks(p, signal, range(0, 20), lambda q: q.last_cycle_rate() is not None)
    assert took is not None and took <= 10, "lock on clean sine"

    p = OscilloscopeProcessor(backend, stable_config())
    warmup, after = 20, 20
    signal = cycle_rate_switch_signal(440.0, 880.0, warmup, after)
    feed_blocks(p, signal, range(0, warmup))
    pre = p.last_cycle_rate()
    assert pre is not None and abs(pre - 440.0) < 20.0
    took = feed_blocks(p, signal, range(warmup, warmup + after),
                       lambda q: q.last_cycle_rate() is not None and abs(q.last_cycle_rate() - 880.0) < 50.0)
    assert took is not None and took <= 10, "adapt to 880Hz"

    p = OscilloscopeProcessor(backend, stable_config())
    silence, blocks = 10, 20
    signal = delayed_sine(440.0, silence, blocks)
    feed_blocks(p, signal, range(0, silence))
    assert p.last_cycle_rate() is None, "should have no cycle rate during silence"
    took = feed_blocks(p, signal, range(silence, silence + blocks), lambda q: q.last_cycle_rate() is not None)
    assert took is not None and took <= 10, "lock after signal onset"


def test_update_config_bumps_epoch_and_rebuilds(backend):
    # :752-758, :593-600
    p = OscilloscopeProcessor(backend, stable_config())
    sig = sine_samples(440.0, RATE, BLOCK * 12)
    s0 = None
    for b in range(12):
        s0 = p.process_block(AudioBlock(sig[b * BLOCK:(b + 1) * BLOCK], 1, RATE)) or s0
    assert s0 is not None and s0.epoch == 0
    c = p.config()
    c.segment_duration = 0.03
    p.update_config(c)
    s1 = p.process_block(AudioBlock(sig[:BLOCK], 1, RATE))
    assert s1 is None or s1.epoch == 1
    p.reset_audio()
    for b in range(3):
        s1 = p.process_block(AudioBlock(sig[b * BLOCK:(b + 1) * BLOCK], 1, RATE)) or s1
    assert s1 is not None and s1.epoch == 2
