"""Parity of the HIP product against the CPU oracle on seeded inputs (the parity tests proper: `-m gpu`).

Tolerances ("1e-5 relative" is against the column maximum, the f32 FFT noise floor makes a
per-bin relative bound meaningless):
  reassigned power   |dP|            <= 1e-5 * max P         (measured 3e-7)
  reassigned freq    |df| * r        <= 1e-7 * fs/2          (measured 5e-10; r = sqrt(P/maxP))
  reassigned time    |dt| * r        <= 1e-4 hops            (measured 4e-6)
  membership         orphan points   <  1e-8 * max P         (bins on the 1e-14 floor / band edge)
  classic u16 codes  |d code|        <= 1  (0.0024 dB; device logf vs glibc logf)
  spectrum traces    |d 10^(dB/10)|  <= 1e-5 * max linear power of the trace (same definition as above);
                     plus |d dB| <= 0.05 dB for every bin more than 1 dB above the floor (weak-bin sanity)
  frame indexing     column counts / offsets bit-exact
"""
import numpy as np
import pytest

import openmeters_amd
from openmeters_amd import banks, capi
from openmeters_amd.capi import (AudioBlock, SpectrogramConfig, SpectrogramProcessor, SpectrumConfig,
                                 SpectrumProcessor)
from parity import bar, exemption, check_classic, check_reassigned_columns, classic_column_metrics, reassigned_column_metrics
from signals import exp_sweep, xorshift32_noise

pytestmark = pytest.mark.gpu


def stream_pcm(s, n, skip=20000):
    left = (exp_sweep(n + skip, phase0=2 * np.pi * s / 64) + xorshift32_noise(0x9E3779B9 ^ s, n + skip, 1e-3))[skip:]
    return np.stack([left, np.float32(0.8) * left], 1).astype(np.float32)


def check_reassigned(got, want, hop):
    check_reassigned_columns(got, want, 48000.0, hop)


@pytest.mark.parametrize("W,hop,zp", [(4096, 256, 1), (1024, 256, 1), (2048, 64, 1), (2048, 512, 4), (256, 32, 1), (8192, 512, 1),
                                      (16384, 2048, 1), (1024, 256, 2), (1024, 100, 4), (2048, 64, 2), (2048, 256, 8),
                                      (4096, 256, 2), (2048, 128, 4), (1024, 64, 8), (1024, 128, 16), (4096, 512, 4), (8192, 1024, 2)])
def test_reassigned_columns_match_oracle(omx, oracle, W, hop, zp):
    assert openmeters_amd.device_available()
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, zero_padding_factor=zp, use_reassignment=True, history_length=8192)
    ncols = 12
    pcm = stream_pcm(3, 2 * W + hop * (ncols - 1)).reshape(-1)
    got = SpectrogramProcessor(omx, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    assert got.fft_size == want.fft_size and got.reset == want.reset
    assert got.reassigned_power_scale == want.reassigned_power_scale
    assert len(got.new_columns) == ncols
    check_reassigned(got.new_columns, want.new_columns, hop)


@pytest.mark.parametrize("W,hop", [(1024, 256), (2048, 64), (4096, 256), (64, 16), (512, 128), (8192, 1024), (16384, 1024)])
def test_classic_columns_match_oracle(omx, oracle, W, hop):
    """W in {1024, 2048, 4096} runs the fused two-columns-per-FFT kernel (radix-16 passes: a different, equally valid f32
    rounding pattern -> level-aware bar, see parity.check_classic); other sizes run the generic kernel, which repeats the
    oracle's radix-2 order and must stay within one code everywhere."""
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, use_reassignment=False, history_length=8192)
    pcm = stream_pcm(5, W + hop * 15).reshape(-1)
    got = SpectrogramProcessor(omx, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    assert len(got.new_columns) == len(want.new_columns) == 16
    if W in (1024, 2048, 4096, 8192, 16384):
        check_classic(got.new_columns, want.new_columns)
    else:
        for h, o in zip(got.new_columns, want.new_columns):
            m = classic_column_metrics(h, o)
            assert m["max_code_diff"] <= 1 and m["n_diff"] <= max(4, m["n"] // 50), m


def test_classic_bank_fused_vs_generic_vs_oracle_and_odd_column_counts(omx, oracle):
    """bank of 5 streams, 7 columns (odd: the last complex FFT carries a single column), one silent stream"""
    S, ncols, W, hop = 5, 7, 1024, 256
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, use_reassignment=False, history_length=8192)
    pcm = np.stack([stream_pcm(s, W + hop * (ncols - 1)) for s in range(S)])
    pcm[3] = 0.0
    fast, gen = banks.SpectrogramBank(omx, cfg, S), banks.SpectrogramBank(omx, cfg, S)
    gen.set_option(capi.OPT_FORCE_GENERIC, 1)
    uf, ug = fast.process_host(pcm, 2, 48000.0), gen.process_host(pcm, 2, 48000.0)
    assert uf.n_columns == ug.n_columns == ncols
    floor_code = int(round((-140.0 + 144.0) * 65535.0 / 156.0))
    for s in range(S):
        want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm[s].reshape(-1), 2, 48000.0)).new_columns
        got_f = [fast.fetch_column(s, c, capi.COLUMN_CLASSIC, uf.column_stride)[:W // 2 + 1] for c in range(ncols)]
        got_g = [gen.fetch_column(s, c, capi.COLUMN_CLASSIC, ug.column_stride)[:W // 2 + 1] for c in range(ncols)]
        check_classic(got_f, want)
        for h, o in zip(got_g, want):
            assert classic_column_metrics(h, o)["max_code_diff"] <= 1
        if s == 3:
            assert all((c == floor_code).all() for c in got_f)


def test_block_partition_and_frame_indexing_are_bit_exact(omx, oracle):
    """256-frame blocks (the DspBatcher quantum, reference src/meter.rs:16): the column count per block,
    the `reset` flag and the per-column point counts must match the oracle block by block."""
    cfg = SpectrogramConfig(fft_size=4096, hop_size=256, use_reassignment=True, history_length=64)
    pcm = stream_pcm(9, 8192 + 256 * 20)
    a, b = SpectrogramProcessor(omx, cfg), SpectrogramProcessor(oracle, cfg)
    produced = 0
    for k in range(0, pcm.shape[0], 256):
        blk = pcm[k:k + 256].reshape(-1)
        g, w = a.process_block(AudioBlock(blk, 2, 48000.0)), b.process_block(AudioBlock(blk, 2, 48000.0))
        assert (g is None) == (w is None)
        if g is None:
            continue
        assert len(g.new_columns) == len(w.new_columns) == 1 and g.reset == w.reset
        check_reassigned(g.new_columns, w.new_columns, 256)
        produced += 1
    assert produced == 21


def test_fast_kernel_equals_generic_kernel_and_bank_equals_single(omx, oracle):
    """64-stream bank (fused 4096 kernel) vs the generic kernel vs single-stream handles."""
    S, ncols = 8, 6
    cfg = SpectrogramConfig(fft_size=4096, hop_size=256, use_reassignment=True, history_length=8192)
    pcm = np.stack([stream_pcm(s, 8192 + 256 * (ncols - 1)) for s in range(S)])
    fast = banks.SpectrogramBank(omx, cfg, S)
    gen = banks.SpectrogramBank(omx, cfg, S)
    gen.set_option(capi.OPT_FORCE_GENERIC, 1)
    uf, ug = fast.process_host(pcm, 2, 48000.0), gen.process_host(pcm, 2, 48000.0)
    assert uf.n_columns == ug.n_columns == ncols and uf.column_stride == 2049
    for s in (0, 3, 7):
        want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm[s].reshape(-1), 2, 48000.0)).new_columns
        got_f = [fast.fetch_column(s, c, capi.COLUMN_REASSIGNED, 2049) for c in range(ncols)]
        got_g = [gen.fetch_column(s, c, capi.COLUMN_REASSIGNED, 2049) for c in range(ncols)]
        check_reassigned(got_f, want, 256)
        check_reassigned(got_g, want, 256)


def test_silent_and_mixed_streams_in_one_bank(omx):
    """Silent fast path (:307-316) is per stream: a silent stream emits empty columns next to a live one."""
    S = 3
    cfg = SpectrogramConfig(fft_size=4096, hop_size=256, use_reassignment=True, history_length=8192)
    pcm = np.zeros((S, 8192 + 512, 2), np.float32)
    pcm[1] = stream_pcm(1, 8192 + 512)
    pcm[2, :100] = 0.25  # non-zero only before the second window's front... still inside window 0
    bank = banks.SpectrogramBank(omx, cfg, S)
    up = bank.process_host(pcm, 2, 48000.0)
    assert up.n_columns == 3
    assert all(len(bank.fetch_column(0, c, capi.COLUMN_REASSIGNED, 2049)) == 0 for c in range(3))
    assert all(len(bank.fetch_column(1, c, capi.COLUMN_REASSIGNED, 2049)) > 1000 for c in range(3))
    assert len(bank.fetch_column(2, 1, capi.COLUMN_REASSIGNED, 2049)) == 0  # front moved past the last non-zero


def check_trace(x, y, floor=-100.0, flush_ties=0):
    """dB traces.  The bound that always holds is the linear-power one (relative to the trace maximum, but not below -60 dB:
    a trace of near-silence is all f32 FFT noise).  dB differences are only meaningful away from (a) the f32 noise floor
    60 / 80 dB under the maximum and (b) the display / state floor, where `update_outputs` flushes a smoothed state to zero
    (spectrum/processor.rs:366-389) — a discontinuity of a few dB that a last-bit difference can trip on either side.
    `flush_ties` (averaging modes only): that flush acts on the STATE, so a bin whose power sat within rounding of the state floor one
    hop ago re-seeds on one side and keeps averaging on the other, and shows dBs of difference at whatever level it has now (soak
    seed 972183085: p = 7.46e-11 against a state floor of 7.46e-11, bin at -75 dB one hop later).  Up to `flush_ties` bins per trace
    may therefore leave the dB bars — the linear-power bar still binds them — and the ledger counts how many did."""
    x, y = x.astype(np.float64), y.astype(np.float64)
    px, py = 10.0 ** (x / 10.0), 10.0 ** (y / 10.0)
    ref = max(py.max(), 1e-6)
    bar("spectrum: |d 10^(dB/10)| / max", np.abs(px - py).max() / ref, 1e-5)
    clear = (y > floor + 12.0) & (x > floor + 12.0)   # a flushed state re-seeds (:366-369): its bin needs a few hops to re-converge
    d = np.abs(x - y)
    near = clear & (y > y.max() - 80.0)
    ties = 0
    if flush_ties and near.any():
        over = np.flatnonzero(near & (d > 0.01))
        if 0 < len(over) <= flush_ties:
            ties = len(over)
            clear = clear.copy()
            clear[over] = False
            near = clear & (y > y.max() - 80.0)
        bar("spectrum (averaging modes): bins beyond the dB bars through a state-flush tie, per trace", ties, flush_ties)
        exemption("spectrum: bins that leave the dB bars through a state-flush tie (averaging modes)", ties > 0, (ties, flush_ties))
    loud = clear & (y > y.max() - 60.0)
    if loud.any():
        bar("spectrum: |d dB| within 60 dB of max", d[loud].max(), 0.01)   # measured 2.1e-3 (profiles/parity_r*.txt)
    if near.any():
        assert d[near].max() <= 0.1


@pytest.mark.parametrize("mode,param", [(capi.AVG_NONE, 0.0), (capi.AVG_EXPONENTIAL, 0.5), (capi.AVG_PEAK_HOLD, 12.0)])
@pytest.mark.parametrize("N,hop", [(4096, 256), (1024, 512), (2048, 128), (512, 128), (8192, 512), (16384, 1024)])
def test_spectrum_matches_oracle(omx, oracle, mode, param, N, hop):
    cfg = SpectrumConfig(fft_size=N, hop_size=hop, averaging_mode=mode, averaging_param=param, source=capi.CH_MID,
                         secondary_source=capi.CH_SIDE, floor_db=-100.0)
    pcm = stream_pcm(11, N + hop * 9)
    pcm[:, 1] = pcm[::-1, 0] * np.float32(0.5)  # make Side differ from Mid
    a, b = SpectrumProcessor(omx, cfg), SpectrumProcessor(oracle, cfg)
    for k in range(0, pcm.shape[0], 1024):  # several calls so the averaging state carries over
        blk = pcm[k:k + 1024].reshape(-1)
        g, w = a.process_block(AudioBlock(blk, 2, 48000.0)), b.process_block(AudioBlock(blk, 2, 48000.0))
        assert (g is None) == (w is None)
        if g is None:
            continue
        assert np.array_equal(g.frequency_bins, w.frequency_bins)
        for t in range(2):
            for wt in range(2):
                check_trace(g.traces[t][wt], w.traces[t][wt])


def test_spectrum_bank_all_hops_equal_per_block_snapshots(omx, oracle):
    """emit_all_hops: hop h of a long bank call == the oracle's snapshot after block h (hop-sized blocks)."""
    cfg = SpectrumConfig(fft_size=4096, hop_size=256, floor_db=-100.0)
    S, hops = 4, 6
    pcm = np.stack([stream_pcm(20 + s, 4096 + 256 * (hops - 1)) for s in range(S)])
    bank = banks.SpectrumBank(omx, cfg, S, emit_all_hops=True)
    up = bank.process_host(pcm, 2, 48000.0)
    assert up.n_hops == up.n_hops_out == hops and up.bins == 2049
    for s in (0, 3):
        p = SpectrumProcessor(oracle, cfg)
        snaps = []
        p.process_block(AudioBlock(pcm[s, :4096 - 256].reshape(-1), 2, 48000.0))
        for h in range(hops):
            lo = 4096 - 256 + 256 * h
            snaps.append(p.process_block(AudioBlock(pcm[s, lo:lo + 256].reshape(-1), 2, 48000.0)))
        for h in range(hops):
            got = bank.fetch(s, h, 2049)
            for wt in range(2):
                check_trace(got[0, wt], snaps[h].traces[0][wt])


@pytest.mark.parametrize("form", [1, 2, 30, 31])
def test_equivalent_kernel_forms_compute_the_same_columns(omx, omx_tuning, oracle, form):
    """OMX_OPT_KERNEL_FORM: the size-templated kernel (30) and the three-kernel form (31) of the product, and the superseded round-1 (1)
    and round-2 pair (2) kernels — compiled into the tuning library only since round 5 (`make TUNING=1`; the product refuses them) —
    must produce the tuned kernel's columns: every form against the oracle at the usual bars, and against form 0 with the
    same point counts on strong columns; a silent stream and an odd window start (unaligned ring pairs) included.  Unknown
    forms are rejected.  (The A/B, phase-timing and knock-out builds live in the tuning library as well.)"""
    if form in (1, 2):
        probe = banks.SpectrogramBank(omx, SpectrogramConfig(fft_size=4096, hop_size=255, use_reassignment=True, history_length=8192), 1)
        with pytest.raises(capi.OmxError):
            probe.set_option(capi.OPT_KERNEL_FORM, form)   # the product holds no superseded form
        if omx_tuning is None:
            pytest.skip("libomx_hip_tuning.so is not built (make -C openmeters_amd/csrc TUNING=1)")
        omx = omx_tuning
    S, ncols = 5, 7
    cfg = SpectrogramConfig(fft_size=4096, hop_size=255, use_reassignment=True, history_length=8192)   # odd hop: odd starts
    pcm = np.stack([stream_pcm(s, 8192 + 255 * (ncols - 1)) for s in range(S)])
    pcm[2] = 0.0
    base, alt = banks.SpectrogramBank(omx, cfg, S), banks.SpectrogramBank(omx, cfg, S)
    alt.set_option(capi.OPT_KERNEL_FORM, form)
    with pytest.raises(capi.OmxError):
        alt.set_option(capi.OPT_KERNEL_FORM, 41)
    ub, ua = base.process_host(pcm, 2, 48000.0), alt.process_host(pcm, 2, 48000.0)
    assert ub.n_columns == ua.n_columns == ncols
    for s in range(S):
        want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm[s].reshape(-1), 2, 48000.0)).new_columns
        got_b = [base.fetch_column(s, c, capi.COLUMN_REASSIGNED, 2049) for c in range(ncols)]
        got_a = [alt.fetch_column(s, c, capi.COLUMN_REASSIGNED, 2049) for c in range(ncols)]
        check_reassigned(got_a, want, 255)
        check_reassigned(got_b, want, 255)
        check_reassigned(got_a, got_b, 255)
        if s == 2:
            assert all(len(c) == 0 for c in got_a)


def test_splat_accumulation_of_device_resident_columns_matches_oracle(omx, oracle):
    """the spectrogram bank's d_points / d_counts are splatted on the device (f32 atomics, order-free)
    and compared with the oracle's sequential accumulation of the same fetched columns: accumulated power within 1e-5 of the
    image maximum, resolved level within 0.05 dB on every pixel within 30 dB of the maximum, empty-pixel masks equal up to edge-sitting points."""
    import ctypes as C
    import torch
    S, ncols = 3, 40
    cfg = SpectrogramConfig(fft_size=4096, hop_size=256, use_reassignment=True, history_length=8192)
    pcm = np.stack([stream_pcm(s, 8192 + 256 * (ncols - 1)) for s in range(S)])
    bank = banks.SpectrogramBank(omx, cfg, S)
    up = bank.process_host(pcm, 2, 48000.0)
    assert up.n_columns == ncols
    f = omx.fn("spectrogram_splat", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p])
    for scale, sf, tilt in ((capi.FREQ_SCALE_LOGARITHMIC, 1.0, 0.0), (capi.FREQ_SCALE_ERB, 2.0, 3.0), (capi.FREQ_SCALE_LINEAR, 1.5, 0.0)):
        view = capi.splat_view(omx, 64.0 * sf, 300.0 * sf, scale_factor=sf, freq_scale=scale, tilt_db=tilt)
        acc = torch.empty((S, view.width, view.height), device="cuda:0", dtype=torch.float32)   # [width][height] per stream
        db = torch.empty_like(acc)
        omx.check(f(up.d_points, up.d_counts, 1, S, ncols, up.column_stride, up.reassigned_power_scale, C.byref(view), None,
                    acc.data_ptr(), db.data_ptr()))
        torch.cuda.synchronize()
        acc, db = acc.cpu().numpy().transpose(0, 2, 1), db.cpu().numpy().transpose(0, 2, 1)
        for s in range(S):
            cols = [bank.fetch_column(s, c, capi.COLUMN_REASSIGNED, 2049) for c in range(ncols)]
            want_acc, want_db = capi.spectrogram_splat(oracle, cols, capi.splat_view(oracle, 64.0 * sf, 300.0 * sf, scale_factor=sf,
                                                                                     freq_scale=scale, tilt_db=tilt),
                                                       up.reassigned_power_scale)
            assert want_acc.max() > 0 and np.abs(acc[s] - want_acc).max() <= 1e-5 * want_acc.max()
            # device asinhf / logf differ from glibc by an ulp: a point sitting on a pixel edge may land one pixel over
            flipped = np.isneginf(db[s]) != np.isneginf(want_db)
            assert flipped.mean() < 0.01, flipped.mean()
            assert (np.maximum(acc[s], want_acc)[flipped] <= 1e-5 * want_acc.max()).all()
            strong = want_acc > 1e-3 * want_acc.max()   # a stray floor-level point moves these by < 0.05 dB
            assert strong.sum() > 30, strong.sum()
            worst = np.abs(db[s][strong] - want_db[strong]).max()
            assert worst <= 0.05, worst


@pytest.mark.parametrize("form", ["1", "2"])
def test_splat_kernel_forms_stay_correct(form):
    """OMX_SPLAT_FORM pins the global-atomic (1) or the LDS-tiled (2) accumulation kernel: both must pass the behavioural
    cases and the device-parity case."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, OMX_SPLAT_FORM=form)
    env.pop("OMX_PARITY_REPORT", None)   # the child session must not overwrite the parent's ledger
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-k", "splat and not forms", os.path.join(here, "test_kat_splat.py"),
                        os.path.join(here, "test_gpu_parity.py")], env=env, capture_output=True, text=True, cwd=os.path.dirname(here))
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert r.returncode == 0 and m and int(m.group(1)) >= 13 and "failed" not in r.stdout, r.stdout[-2000:]


def test_history_ring_fed_from_device_resident_bank_updates_matches_oracle_ring(omx, oracle):
    """SURVEY §8f rank 2, the slot ring: bank updates stay on the device (omx_spectrogram_bank_history_apply scatters the new
    columns into their slots), across several calls, a history_length change (ring resize with remap) and a reset; every
    stream's ring — slot counts, newest slot content, and the image splatted from the ring — against the oracle's ring fed with
    the same columns fetched to the host."""
    S = 3
    cfg = SpectrogramConfig(fft_size=1024, hop_size=256, use_reassignment=True, history_length=12)
    bank = banks.SpectrogramBank(omx, cfg, S)
    hist = capi.SpectrogramHistory(omx, S)
    refs = [capi.SpectrogramHistory(oracle) for _ in range(S)]
    view = capi.splat_view(omx, 40.0, 96.0, scale_factor=2.0)
    at = 0
    total = 2048 + 256 * 60
    pcm = np.stack([stream_pcm(20 + s, total) for s in range(S)])
    plan = [(2048 + 256 * 4, None), (256 * 9, None), (256 * 3, 20), (256 * 15, None), (256 * 2, 5), (256 * 7, None), ("reset", None),
            (2048 + 256 * 3, None)]
    for n, new_len in plan:
        if n == "reset":
            bank.reset_audio()
            continue
        if new_len is not None:
            cfg = SpectrogramConfig(fft_size=1024, hop_size=256, use_reassignment=True, history_length=new_len)
            bank.update_config(cfg)
        up = bank.process_host(pcm[:, at:at + n], 2, 48000.0)
        at += n
        assert up is not None
        hist.apply_bank(up)
        for s in range(S):
            cols = [bank.fetch_column(s, c, capi.COLUMN_REASSIGNED, up.column_stride) for c in range(up.n_columns)]
            refs[s].apply(capi.SpectrogramUpdate(up.fft_size, up.hop_size, up.sample_rate, up.history_length, bool(up.reset),
                                                 up.reassigned_power_scale, up.kind, cols))
        gi = hist.info()
        acc, db = hist.splat(view, up.reassigned_power_scale)
        for s in range(S):
            ri = refs[s].info()
            assert (gi.ring_capacity, gi.write_slot, gi.col_count, gi.newest_slot, gi.visible_slots) == \
                   (ri.ring_capacity, ri.write_slot, ri.col_count, ri.newest_slot, ri.visible_slots)
            assert np.array_equal(hist.slot_counts(s), refs[s].slot_counts())
            assert np.array_equal(hist.fetch_slot(gi.newest_slot, s), refs[s].fetch_slot(ri.newest_slot))
            want_acc, want_db = refs[s].splat(capi.splat_view(oracle, 40.0, 96.0, scale_factor=2.0), up.reassigned_power_scale)
            bar("history ring splat: |d accumulated power| / max", np.abs(acc[s] - want_acc[0]).max() / max(want_acc.max(), 1e-30), 1e-5)
        assert gi.reassigned_points_per_slot == refs[0].info().reassigned_points_per_slot


@pytest.mark.parametrize("W,hop,zp,reassign", [(1000, 250, 1, True), (1536, 384, 1, True), (3000, 750, 1, True), (1000, 250, 3, True),
                                                (1024, 256, 3, True), (1000, 250, 1, False), (1536, 512, 1, False), (1024, 256, 3, False)])
def test_lengths_that_are_not_powers_of_two_match_oracle(omx, oracle, W, hop, zp, reassign):
    """The reference plans any transform length (rustfft; `spectrogram/processor.rs:71-82` only normalises the configuration).  The HIP
    path runs them as Bluestein chirp-z transforms on the generic kernel; the oracle evaluates the plain DFT sum."""
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, zero_padding_factor=zp, use_reassignment=reassign, history_length=16)
    need = (1 << int(np.ceil(np.log2(2 * W)))) if reassign else W * zp   # the Hilbert step works on next_pow2(2 W) samples (:225-227)
    pcm = stream_pcm(7, need + hop * 3)
    blk = AudioBlock(pcm.reshape(-1), 2, 48000.0)
    g, w = SpectrogramProcessor(omx, cfg).process_block(blk), SpectrogramProcessor(oracle, cfg).process_block(blk)
    assert g is not None and w is not None and len(g.new_columns) == len(w.new_columns) and len(w.new_columns) >= 3
    assert g.fft_size == w.fft_size == W * zp
    if reassign:
        assert g.reassigned_power_scale == w.reassigned_power_scale
        check_reassigned(g.new_columns, w.new_columns, hop)
    else:
        check_classic(g.new_columns, w.new_columns)


@pytest.mark.parametrize("N,hop", [(1000, 250), (3000, 1000), (1536, 384)])
def test_spectrum_lengths_that_are_not_powers_of_two_match_oracle(omx, oracle, N, hop):
    cfg = SpectrumConfig(fft_size=N, hop_size=hop)
    pcm = stream_pcm(8, N + hop * 5)
    blk = AudioBlock(pcm.reshape(-1), 2, 48000.0)
    g, w = SpectrumProcessor(omx, cfg).process_block(blk), SpectrumProcessor(oracle, cfg).process_block(blk)
    assert g is not None and w is not None and np.array_equal(g.frequency_bins, w.frequency_bins)
    for wt in range(2):
        check_trace(g.traces[0][wt], w.traces[0][wt])



def _quiet_then_loud(n_quiet, n_loud, level_db):
    """a tone at `level_db` dBFS for n_quiet frames, then the same at full scale: adjacent windows tens of dB apart"""
    t = np.arange(n_quiet + n_loud) / 48000.0
    x = 0.7 * np.sin(2 * np.pi * 1234.5 * t) + 0.2 * np.sin(2 * np.pi * 5432.1 * t + 0.4)
    x[:n_quiet] *= 10.0 ** (level_db / 20.0)
    return np.stack([x, 0.8 * x], 1).astype(np.float32)


@pytest.mark.parametrize("W,level_db", [(1024, -100.0), (2048, -60.0), (4096, -40.0)])
def test_classic_quiet_column_paired_with_a_loud_one(omx, oracle, W, level_db):
    """The mechanism behind soak seeds 21028002 / 21093005 / 21109003 (round 5), pinned: the fused classic kernels transform two
    consecutive columns as the real and imaginary part of ONE complex transform, and the split cancels the partner's spectrum only to
    ~4e-7 of the partner's largest bin — a column 97 / 64 / 42 dB under its partner came out 211 / 6 / 2 codes off.  Since round 5 every
    column is brought to a common level by an exact power of two before the transform; hop = W makes the windows disjoint, so column 1
    (quiet) rides with column 0 ... the pairs (0, 1), (2, 3): the boundary falls inside a pair."""
    cfg = SpectrogramConfig(fft_size=W, hop_size=W, use_reassignment=False, history_length=8192)
    pcm = _quiet_then_loud(3 * W, 3 * W, level_db).reshape(-1)   # columns 0 1 2 quiet, 3 4 5 loud: pair (2, 3) straddles the step
    got = SpectrogramProcessor(omx, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    assert len(got.new_columns) == len(want.new_columns) == 6
    for h, o in zip(got.new_columns, want.new_columns):
        m = classic_column_metrics(h, o)
        bar("classic (fused): |d code| within 40 dB of max, a quiet column paired with a loud one", m["loud_code_diff"], 1, m)


@pytest.mark.parametrize("N", [1024, 4096])
def test_spectrum_constant_and_silent_hops_paired_with_loud_ones(omx, oracle, N):
    """edge cases of the level equalisation of the paired hops (its scale comes from a hop's sample range, max - min): a hop of one
    constant (range 0: scale 1, the constant leaves with the mean, window.rs:80-84) and an all-zero hop, each beside a loud one.
    (Hops that are a LARGE constant plus a small signal: tests/test_gpu_dc_offset.py — the kernels take the mean in the reference's
    sequential order since round 6.)"""
    from openmeters_amd import banks
    cfg = SpectrumConfig(fft_size=N, hop_size=N, floor_db=-140.0)
    rng = np.random.default_rng(77)
    loud = rng.uniform(-1.0, 1.0, (N, 2)).astype(np.float32)
    flat, zero = np.full((N, 2), 0.25, np.float32), np.zeros((N, 2), np.float32)
    pcm = np.concatenate([flat, loud, loud, flat, zero, loud])   # pairs (flat, loud), (loud, flat), (zero, loud)
    bank = banks.SpectrumBank(omx, cfg, 1, emit_all_hops=True)
    up = bank.process_host(pcm[None], 2, 48000.0)
    assert up is not None and int(up.n_hops) == 6
    ref = SpectrumProcessor(oracle, cfg)
    for h in range(6):
        w = ref.process_block(AudioBlock(pcm[h * N:(h + 1) * N].reshape(-1), 2, 48000.0))
        g = bank.fetch(0, h, N // 2 + 1)
        check_trace(g[0][0], w.traces[0][0], floor=-140.0)
        check_trace(g[0][1], w.traces[0][1], floor=-140.0)


@pytest.mark.parametrize("N,level_db", [(1024, -90.0), (4096, -60.0)])
def test_spectrum_quiet_hop_paired_with_a_loud_one(omx, oracle, N, level_db):
    """the same mechanism in the spectrum bank (two consecutive hops share one complex transform): every hop's trace — emit_all_hops —
    against the oracle's per-hop snapshots, the quiet hops next to the loud ones included"""
    from openmeters_amd import banks
    cfg = SpectrumConfig(fft_size=N, hop_size=N, floor_db=-140.0)
    pcm = _quiet_then_loud(3 * N, 3 * N, level_db)
    bank = banks.SpectrumBank(omx, cfg, 1, emit_all_hops=True)
    up = bank.process_host(pcm[None], 2, 48000.0)
    assert up is not None and int(up.n_hops) == 6
    ref = SpectrumProcessor(oracle, cfg)
    for h in range(6):
        w = ref.process_block(AudioBlock(pcm[h * N:(h + 1) * N].reshape(-1), 2, 48000.0))
        g = bank.fetch(0, h, N // 2 + 1)
        check_trace(g[0][0], w.traces[0][0], floor=-140.0)
        check_trace(g[0][1], w.traces[0][1], floor=-140.0)
