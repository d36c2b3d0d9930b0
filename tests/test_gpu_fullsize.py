"""BASELINE.json's full single-GPU sizes (cfg2 64 streams, cfg3 1024 x 8 ch, cfg4 256 streams), checked through
size-independent properties — the CPU oracle cannot replay these volumes in seconds, so it only spot-checks.

  shift invariance   stream k carries stream 0's PCM advanced by k hops  =>  column c of stream k is BIT-identical to
                     column c + k of stream 0 (same samples, same arithmetic, different ring offsets / workgroups / XCDs)
  partition          one call == the same PCM fed in several calls (frame indexing, ring wrap, carried state), bit-exact
  replication        identical streams at different bank indices produce identical bits
  gain               PCM x 0.5 (exact in f32/f64) moves every LUFS / dB field by -6.0206 dB and leaves rho unchanged
"""
import numpy as np
import pytest

from openmeters_amd import banks, capi
from openmeters_amd.capi import (AudioBlock, LoudnessConfig, LoudnessProcessor, OscilloscopeConfig, OscilloscopeProcessor,
                                 SpectrogramConfig, SpectrogramProcessor, StereometerConfig, StereometerProcessor)
from golden_inputs import cfg2_pcm, cfg3_pcm, cfg4_pcm
from parity import bar, check_chunked_rho, check_reassigned_columns, reassigned_column_metrics, stereometer_band_rms

pytestmark = pytest.mark.gpu
FS = 48000.0


class View:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2,
                                         "strides": None}


def dview(torch, ptr, shape, typestr="<i4"):
    return torch.as_tensor(View(ptr, shape, typestr), device="cuda:0")


def spectrogram_outputs(torch, up):
    S, cols, stride = int(up.n_streams), int(up.n_columns), int(up.column_stride)
    counts = dview(torch, up.d_counts, (S, cols)).clone()
    points = dview(torch, up.d_points, (S, cols, stride, 3)).clone()  # f32 bits as i32: NaN-safe bit comparison
    valid = torch.arange(stride, device="cuda:0")[None, None, :] < counts[..., None]
    points[~valid] = 0
    return counts, points


def test_cfg2_full_size_shift_partition_and_oracle_spot_checks(omx, oracle):
    import torch
    S, cols, hop = 64, 1024, 256
    frames = 8192 + hop * (cols - 1)                       # 65 536 STFT frames per call: the bench launch
    base = cfg2_pcm(5, frames + hop * (S - 1))
    pcm = np.stack([base[k * hop:k * hop + frames] for k in range(S)])
    cfg = SpectrogramConfig(fft_size=4096, hop_size=hop, use_reassignment=True, history_length=8192)
    d_pcm = torch.from_numpy(pcm).to("cuda:0")
    pos = capi.positions_fallback(2)

    one = banks.SpectrogramBank(omx, cfg, S)
    up = one.process_device(d_pcm.data_ptr(), frames, 2, FS, pos)
    assert up.n_columns == cols and up.column_stride == 2049   # ready = (pending - 8192) / hop + 1
    counts, points = spectrogram_outputs(torch, up)
    assert int(counts.min()) > 1900                          # sweep + noise: nearly every bin is above the floor
    for k in (1, 2, 7, 8, 9, 31, 63):                        # across XCDs (k % 8) and far apart
        assert torch.equal(counts[k, :cols - k], counts[0, k:]), k
        assert torch.equal(points[k, :cols - k], points[0, k:]), k

    # partition: three uneven calls (the second one is shorter than a window) on a fresh bank
    parts = banks.SpectrogramBank(omx, cfg, S)
    got_c, got_p, at = [], [], 0
    for n in (8192 + hop * 300 + 77, 1000, frames - (8192 + hop * 300 + 77) - 1000):
        chunk = d_pcm[:, at:at + n].contiguous()
        u = parts.process_device(chunk.data_ptr(), n, 2, FS, pos)
        at += n
        if u is not None:
            c, p = spectrogram_outputs(torch, u)
            got_c.append(c)
            got_p.append(p)
    assert torch.equal(torch.cat(got_c, 1), counts)
    assert torch.equal(torch.cat(got_p, 1), points)

    # oracle spot checks: column c of stream k is the first column of a fresh processor fed from frame (k + c) * hop
    for k, c in ((0, 0), (17, 400), (63, 1023)):
        at = (k + c) * hop
        want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(base[at:at + 8192].reshape(-1), 2, FS)).new_columns[0]
        got = one.fetch_column(k, c, capi.COLUMN_REASSIGNED, 2049)
        check_reassigned_columns([got], [want], FS, hop)


def test_cfg3_full_size_replication_gain_and_oracle_spot_checks(omx, oracle):
    import torch
    S, C, blocks = 1024, 8, 80                               # 20 480 frames: the 400 ms momentary window is full
    frames = 256 * blocks
    distinct = np.stack([cfg3_pcm(s, frames, C) for s in range(16)])
    d_pcm = torch.from_numpy(distinct).to("cuda:0").repeat(S // 16, 1, 1).contiguous()   # stream s = distinct[s % 16]
    bank = banks.LoudnessBank(omx, LoudnessConfig(), S, C)
    ptr = bank.process_device(d_pcm.data_ptr(), 256, blocks, C, FS, capi.SURROUND)
    snaps = dview(torch, ptr, (S, blocks, 30)).clone()
    assert torch.equal(snaps[:16].repeat(S // 16, 1, 1), snaps)

    half = banks.LoudnessBank(omx, LoudnessConfig(), S, C)
    d_half = (d_pcm * 0.5).contiguous()
    hptr = half.process_device(d_half.data_ptr(), 256, blocks, C, FS, capi.SURROUND)
    a = dview(torch, ptr, (S, blocks, 30), "<f4")[:, -1]
    b = dview(torch, hptr, (S, blocks, 30), "<f4")[:, -1]
    step = 10.0 * np.log10(4.0)
    for lo, hi in ((0, 2), (2, 2 + C), (10, 10 + C), (18, 18 + C)):   # LUFS pair, rms fast, rms slow, true peak
        d = (a[:, lo:hi] - b[:, lo:hi]).cpu().numpy()
        bar("cfg3 full size: |d dB - 6.0206| under x0.5 gain", np.abs(d - step).max(), 2e-4, lo)

    for s in (0, 9, 1023):
        p = LoudnessProcessor(oracle, LoudnessConfig())
        for k in range(0, frames, 256):
            w = p.process_block(AudioBlock(distinct[s % 16, k:k + 256].reshape(-1), C, FS, capi.SURROUND))
        g = bank.fetch(s, blocks - 1)
        bar("loudness: |d momentary LUFS|", abs(g.momentary_loudness - w.momentary_loudness), 1e-4)
        bar("loudness: |d short-term LUFS|", abs(g.short_term_loudness - w.short_term_loudness), 1e-4)
        for f in ("rms_fast_db", "rms_slow_db", "true_peak_db"):
            bar(f"loudness: |d {f}|", np.abs(getattr(g, f) - getattr(w, f)).max(), 1e-4)


def test_cfg4_full_size_replication_and_oracle_spot_checks(omx, oracle):
    import torch
    S, blocks = 256, 60
    frames = 256 * blocks
    distinct = np.stack([cfg4_pcm(s, frames) for s in range(32)])
    d_pcm = torch.from_numpy(distinct).to("cuda:0").repeat(S // 32, 1, 1).contiguous()
    pos = capi.positions_fallback(2)
    scfg = StereometerConfig(analyze_bands=True, correlation_window=0.05, segment_duration=0.02, target_sample_count=2000)
    ocfg = OscilloscopeConfig(segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2, trigger_source=capi.CH_LEFT,
                              channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT)
    st, sc = banks.StereometerBank(omx, scfg, S), banks.OscilloscopeBank(omx, ocfg, S)
    us = st.process_device(d_pcm.data_ptr(), 256, blocks, 2, FS, pos)
    uo = sc.process_device(d_pcm.data_ptr(), 256, blocks, 2, FS, pos)
    corr = dview(torch, us.d_correlations, (S, blocks, 4)).clone()
    hdr = dview(torch, uo.d_headers, (S, blocks, 10))[:, :, :8].clone()   # capture_start / capture_frac follow
    smp = dview(torch, uo.d_samples, (S, 2, int(uo.sample_stride))).clone()
    assert torch.equal(corr[:32].repeat(S // 32, 1, 1), corr)
    assert torch.equal(hdr[:32].repeat(S // 32, 1, 1), hdr)
    n = int(hdr[0, -1, 4])                                   # samples_per_channel of the newest snapshot
    assert torch.equal(smp[:32, :, :n].repeat(S // 32, 1, 1), smp[:, :, :n])
    assert int(hdr[:, -1, 5].sum()) == S                     # every stream is locked after 320 ms of a periodic tone

    # gain leaves rho untouched (EMA of products scales numerator and denominator alike; x0.5 is exact)
    st2 = banks.StereometerBank(omx, scfg, S)
    d_half = (d_pcm * 0.5).contiguous()
    us2 = st2.process_device(d_half.data_ptr(), 256, blocks, 2, FS, pos)
    c1 = dview(torch, us.d_correlations, (S, blocks, 4), "<f4")
    c2 = dview(torch, us2.d_correlations, (S, blocks, 4), "<f4")
    bar("cfg4 full size: |d rho| under x0.5 gain", float((c1 - c2).abs().max()), 1e-6)

    for s in (0, 21, 255):
        sp, op = StereometerProcessor(oracle, scfg), OscilloscopeProcessor(oracle, ocfg)
        for k in range(0, frames, 256):
            blk = AudioBlock(distinct[s % 32, k:k + 256].reshape(-1), 2, FS)
            ws, wo = sp.process_block(blk), op.process_block(blk)
        got, produced = st.fetch(s, blocks - 1)
        assert produced      # 256 streams x 60 blocks: the bank takes the chunk-parallel form (stereometer_chunked.hip)
        check_chunked_rho(got, ws.correlations, stereometer_band_rms(distinct[s % 32]), s)
        h, samples = sc.fetch(s, blocks - 1, with_samples=True)
        assert bool(h.locked) == (op.last_cycle_rate() is not None) and h.samples_per_channel == wo.samples_per_channel
        bar("oscilloscope: rel |d cycle rate|", abs(h.period - FS / op.last_cycle_rate()) / h.period, 1e-4)
        flat = np.concatenate([samples[c, :h.samples_per_channel] for c in range(h.channels)])
        from test_gpu_parity_meters import check_stable_trace
        check_stable_trace("oscilloscope (Stable)", flat, wo.samples, (h.capture_start, h.capture_frac), op.last_capture(),
                           float(np.abs(np.diff(distinct[s % 32], axis=0)).max()), h.samples_per_channel,
                           abs(h.period - FS / op.last_cycle_rate()) / h.period, s)


@pytest.mark.parametrize("W,hop", [(2048, 64), (1024, 256), (8192, 512)])
def test_fused_small_windows_full_size_shift_partition_and_oracle(omx, oracle, W, hop):
    """The reference's DEFAULT spectrogram shape (2048 / hop 64, processor.rs:58-59) and 1024 / 256 go through the size-templated
    fused kernel (several columns per workgroup): shift invariance across frame slots / workgroups / XCDs, partition
    invariance, the generic kernel and the oracle as cross-checks."""
    import torch
    S, cols = 64, 1026                                        # not a multiple of the 2 / 4 columns a workgroup carries
    frames = 2 * W + hop * (cols - 1)
    base = cfg2_pcm(11, frames + hop * (S - 1))
    pcm = np.stack([base[k * hop:k * hop + frames] for k in range(S)])
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, use_reassignment=True, history_length=8192)
    d_pcm = torch.from_numpy(pcm).to("cuda:0")
    pos = capi.positions_fallback(2)
    one = banks.SpectrogramBank(omx, cfg, S)
    up = one.process_device(d_pcm.data_ptr(), frames, 2, FS, pos)
    assert up.n_columns == cols and up.column_stride == W // 2 + 1
    counts, points = spectrogram_outputs(torch, up)
    for k in (1, 2, 3, 5, 8, 63):
        assert torch.equal(counts[k, :cols - k], counts[0, k:]), k
        assert torch.equal(points[k, :cols - k], points[0, k:]), k
    parts = banks.SpectrogramBank(omx, cfg, S)
    got_c, got_p, at = [], [], 0
    for n in (2 * W + hop * 301 + 13, hop // 2, frames - (2 * W + hop * 301 + 13) - hop // 2):
        chunk = d_pcm[:, at:at + n].contiguous()
        u = parts.process_device(chunk.data_ptr(), n, 2, FS, pos)
        at += n
        if u is not None:
            c, p = spectrogram_outputs(torch, u)
            got_c.append(c)
            got_p.append(p)
    assert torch.equal(torch.cat(got_c, 1), counts) and torch.equal(torch.cat(got_p, 1), points)
    gen = banks.SpectrogramBank(omx, cfg, 2)
    gen.set_option(capi.OPT_FORCE_GENERIC, 1)
    gen.process_device(d_pcm[:2].contiguous().data_ptr(), frames, 2, FS, pos)
    for k, c in ((0, 0), (1, 517), (1, cols - 1)):
        at = (k + c) * hop
        want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(base[at:at + 2 * W].reshape(-1), 2, FS)).new_columns[0]
        for bank in (one, gen):
            got = bank.fetch_column(k, c, capi.COLUMN_REASSIGNED, W // 2 + 1)
            check_reassigned_columns([got], [want], FS, hop)


@pytest.mark.parametrize("history", [False, True])
def test_waveform_bank_1024_streams_partition_and_replication(omx, history):
    """The role-per-wavefront waveform kernel at bank size (1024 streams = 256 workgroups of 5 / 7 wavefronts): one call == the same
    PCM fed in several calls of uneven lengths (rounds, batches and PCM refills cut differently; carried filter / window / min-max
    state), bit-exact; identical streams at different bank indices (different workgroups, different lane groups) produce identical
    bits; and the one-wavefront kernel's partition agrees with it (OMX_WAVEFORM_SINGLE is per process: covered by
    test_gpu_waveform_forms.py)."""
    import torch
    S, frames = 1024, 6000
    cfg = capi.WaveformConfig(scroll_speed=350.0, max_columns=256, analyze_bands=True, track_history=history)
    base = np.stack([cfg4_pcm(s, frames) for s in range(8)])
    pcm = np.ascontiguousarray(base[np.arange(S) % 8])                 # stream s carries base[s % 8]
    d_pcm = torch.from_numpy(pcm).to("cuda:0")
    pos = capi.positions_fallback(2)

    def run(cuts, form=1):
        bank = banks.WaveformBank(omx, cfg, S)
        bank.set_option(capi.OPT_KERNEL_FORM, form)   # 1: the sequential kernels this test is about (by shape the single call would go chunk-parallel)
        cols, at = [], 0
        for n in cuts:
            part = d_pcm[:, at:at + n].contiguous()
            up = bank.process_device(part.data_ptr(), n, 2, FS, pos)
            torch.cuda.synchronize()
            if up is not None and int(up.n_columns):
                v = dview(torch, up.d_columns, (S, int(up.n_columns), 4, 11)).clone()
                cols.append(v)
            at += n
        return torch.cat(cols, dim=1)

    one = run([frames])
    many = run([1000, 17, 2048, 1, 935, 1999])
    assert one.shape == many.shape and one.shape[1] > 30
    assert torch.equal(one, many)
    for s in (8, 9, 511, 1016, 1023):
        assert torch.equal(one[s], one[s % 8]), s
    # the chunk-parallel form (waveform_chunked.hip) at the same size: min / max bit-identical to the sequential kernels', colour bands and
    # history under the three-way bars of tests/test_gpu_parity_meters.py — the sequential kernels are the reference's f32 evaluation bit for
    # bit, the chunk form follows the f64 recurrence on the 200 Hz sections, so their distance is the reference's own distance to exact —
    # identical streams identical bits, and its own partition (chunk-parallel calls handing over to each other)
    import test_gpu_parity_meters as meters
    chunk = run([frames], form=2)
    assert torch.equal(chunk[..., :2], one[..., :2])
    c32, o32 = chunk.view(torch.float32).cpu().numpy(), one.view(torch.float32).cpu().numpy()   # (run() returns the f32 bits as i32)
    for s in range(8):
        exact = meters.WaveExact(base[s], FS, 350.0)
        assert len(exact) == one.shape[1]
        meters.check_wave_three_way("waveform (chunk-parallel, 1024 streams)", c32[s], o32[s], exact, slice(0, one.shape[1]), history, s)
    for s in (8, 9, 511, 1016, 1023):
        assert torch.equal(chunk[s], chunk[s % 8]), s
    parts = run([2048, 1024, 2928], form=2)
    assert torch.equal(parts[..., :2], one[..., :2])
    p32 = parts.view(torch.float32).cpu().numpy()
    for s in range(8):
        meters.check_wave_three_way("waveform (chunk-parallel, 1024 streams, three calls)", p32[s], o32[s], meters.WaveExact(base[s], FS, 350.0),
                                    slice(0, one.shape[1]), history, s)


@pytest.mark.parametrize("S,frames", [(1024, 16384), (2048, 16384)])
def test_waveform_chunk_parallel_form_at_bench_size_against_the_sequential_kernels(omx, S, frames):
    """the chunk lengths a test-sized bank never reaches (the planner takes 64-frame chunks below 131072 (stream, chunk) items, 128
    at 1024 x 16384 — the bench call — and 256 from 2048 x 16384): two calls of the chunk-parallel form against two calls of the sequential
    kernels on the same device PCM; min / max bit-identical, colour bands under the three-way bars (the chunk form follows the f64
    recurrence on the 200 Hz sections: its distance to the sequential kernels is the reference's own distance to exact), identical
    streams identical bits"""
    import torch
    cfg = capi.WaveformConfig(scroll_speed=300.0, max_columns=256, analyze_bands=True, track_history=False)
    base = np.stack([cfg4_pcm(90 + s, 2 * frames) for s in range(8)])
    d_pcm = torch.from_numpy(np.ascontiguousarray(base[np.arange(S) % 8])).to("cuda:0")
    pos = capi.positions_fallback(2)
    out = {}
    for form in (1, 2):
        bank = banks.WaveformBank(omx, cfg, S)
        bank.set_option(capi.OPT_KERNEL_FORM, form)
        cols = []
        for k in range(2):
            part = d_pcm[:, k * frames:(k + 1) * frames].contiguous()
            up = bank.process_device(part.data_ptr(), frames, 2, FS, pos)
            torch.cuda.synchronize()
            assert bank.last_form() == form
            cols.append(dview(torch, up.d_columns, (S, int(up.n_columns), 4, 11), "<f4").clone())
        out[form] = torch.cat(cols, dim=1)
    seq, chunk = out[1].double(), out[2].double()
    assert seq.shape == chunk.shape and seq.shape[1] > 190
    assert torch.equal(out[1][..., :2], out[2][..., :2])
    import test_gpu_parity_meters as meters   # (the three-way bars: the sequential kernels = the reference's f32 evaluation, exact = the f64 recurrence)
    for s in range(8):
        exact = meters.WaveExact(base[s], FS, 300.0)
        kept = seq.shape[1]   # (max_columns = 256 per call: the newest columns of each call)
        per_call = kept // 2
        ends = np.searchsorted(exact.ends, [frames, 2 * frames])
        for k in range(2):
            cols = slice(int(ends[k]) - per_call, int(ends[k]))
            meters.check_wave_three_way("waveform (chunk-parallel, bench size)", out[2][s, k * per_call:(k + 1) * per_call].cpu().numpy(),
                                        out[1][s, k * per_call:(k + 1) * per_call].cpu().numpy(), exact, cols, False, (S, s, k))
    for s in (8, 9, S // 2 + 3, S - 1):
        assert torch.equal(out[2][s], out[2][s % 8]), s


def test_ragged_calls_with_equal_counts_are_bit_identical_to_lock_step_calls_at_bank_size(omx):
    """The ragged entry points run the same kernels as the lock-step ones with per-stream counters compiled in (RAGGED template
    parameters): given the same count for every stream they must reproduce the lock-step call BIT FOR BIT — loudness 1024 x 8 ch
    (cfg3), stereometer + oscilloscope 256 streams (cfg4), two calls each so that the second starts from carried per-stream state."""
    import torch
    blocks = 40
    frames = 256 * blocks
    # ---- loudness, cfg3 shape
    S, C = 1024, 8
    distinct = np.stack([cfg3_pcm(s, 2 * frames, C) for s in range(8)])
    d_all = torch.from_numpy(distinct).to("cuda:0").repeat(S // 8, 1, 1).contiguous()
    a, b = banks.LoudnessBank(omx, LoudnessConfig(), S, C), banks.LoudnessBank(omx, LoudnessConfig(), S, C)
    for call in range(2):
        part = d_all[:, call * frames:(call + 1) * frames].contiguous()
        pa = a.process_device(part.data_ptr(), 256, blocks, C, FS, capi.SURROUND)
        ub = b.process_ragged(part.data_ptr(), 256, blocks, [blocks] * S, C, FS, capi.SURROUND)
        assert a.last_form() == 2 and b.last_form() == 2
        assert torch.equal(dview(torch, pa, (S, blocks, 30)), dview(torch, ub.d_snapshots, (S, blocks, 30))), call
    del d_all, a, b
    # ---- stereometer + oscilloscope, cfg4 shape
    S = 256
    distinct = np.stack([cfg4_pcm(s, 2 * frames) for s in range(32)])
    d_all = torch.from_numpy(distinct).to("cuda:0").repeat(S // 32, 1, 1).contiguous()
    pos = capi.positions_fallback(2)
    scfg = StereometerConfig(analyze_bands=True, emit_band_points=True, correlation_window=0.05, segment_duration=0.02, target_sample_count=2000)
    ocfg = OscilloscopeConfig(segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2, trigger_source=capi.CH_LEFT,
                              channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT)
    st_a, st_b = banks.StereometerBank(omx, scfg, S), banks.StereometerBank(omx, scfg, S)
    sc_a, sc_b = banks.OscilloscopeBank(omx, ocfg, S), banks.OscilloscopeBank(omx, ocfg, S)
    for call in range(2):
        part = d_all[:, call * frames:(call + 1) * frames].contiguous()
        ua = st_a.process_device(part.data_ptr(), 256, blocks, 2, FS, pos)
        ub = st_b.process_ragged(part.data_ptr(), 256, blocks, [blocks] * S, 2, FS, pos)
        assert st_a.last_form() == 2 and st_b.last_form() == 2
        assert torch.equal(dview(torch, ua.d_correlations, (S, blocks, 4)), dview(torch, ub.d_correlations, (S, blocks, 4))), call
        assert torch.equal(dview(torch, ua.d_produced, (S, blocks)), dview(torch, ub.d_produced, (S, blocks))), call
        n = int(ua.target)
        assert n == int(ub.target)
        assert torch.equal(dview(torch, ua.d_points, (S, 4, n, 2)), dview(torch, ub.d_points, (S, 4, n, 2))), call
        oa = sc_a.process_device(part.data_ptr(), 256, blocks, 2, FS, pos)
        ob = sc_b.process_ragged(part.data_ptr(), 256, blocks, [blocks] * S, 2, FS, pos)
        assert torch.equal(dview(torch, oa.d_headers, (S, blocks, 10)), dview(torch, ob.d_headers, (S, blocks, 10))), call
        spc = dview(torch, oa.d_headers, (S, blocks, 10))[:, -1, 4]
        m = int(spc.min())
        assert m > 0
        stride = int(oa.sample_stride)
        assert torch.equal(dview(torch, oa.d_samples, (S, 2, stride))[:, :, :m], dview(torch, ob.d_samples, (S, 2, stride))[:, :, :m]), call
