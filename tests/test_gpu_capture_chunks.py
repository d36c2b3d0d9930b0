"""The capture group delivers the reference's blocks (missing #1).

`DspBatcher::push` hands `VisualManager::ingest_samples` chunks of 1 ... 4 quanta after a stall, EACH AS ONE CALL
(reference src/meter.rs:40-69), and `ingest_samples` builds ONE AudioBlock from the chunk and calls every visual's process_block once
(src/visuals/registry.rs:396-418): one oscilloscope trigger evaluation (oscilloscope/processor.rs:611-712), one true-peak take
(loudness/processor.rs:287-311), one stereo_channels scan (dsp.rs:190-213) over the whole chunk.

Here the product's own batcher (omx_batcher_push) is driven with stalls so that it emits 256 / 512 / 768 / 1024-frame chunks; every chunk is
forwarded whole to the capture group — omx_capture_group_ingest for captures in lock step, omx_capture_group_ingest_ragged when every
capture has its own batcher and its own packet sizes — and every capture is compared with single-stream ORACLE handles fed the same
chunk whole."""
import numpy as np
import pytest

from openmeters_amd import capi
from openmeters_amd.capi import (AudioBlock, LoudnessConfig, LoudnessProcessor, OscilloscopeConfig, OscilloscopeProcessor, SpectrogramConfig,
                                 SpectrogramProcessor, SpectrumConfig, SpectrumProcessor, StereometerConfig, StereometerProcessor)
from parity import bar
from test_kat_batcher import Batcher, fmt

pytestmark = pytest.mark.gpu
FS = 48000.0
QUANTUM = 256


def capture_pcm(s, frames, channels, rng):
    """capture s: a two-partial tone on the front pair (right = -0.7 left, so rho < 0), a click train (true peak between the samples), a
    little noise; with 8 channels the surround / centre channels carry signal only in every fourth quantum — bit-zero in the other
    three (stereo_channels trims them per BLOCK, dsp.rs:197-204: a 1024-frame chunk scanned whole keeps all eight)."""
    t = np.arange(frames) / FS
    f0 = 110.0 * 2.0 ** (s / 3.0)
    left = 0.55 * np.sin(2 * np.pi * f0 * t) + 0.2 * np.sin(2 * np.pi * 2 * f0 * t + 0.3 * s) + 0.003 * rng.standard_normal(frames)
    left[(np.arange(frames) % 1777) == 5 * s] += 0.35
    x = np.zeros((frames, channels), np.float64)
    x[:, 0] = left
    if channels > 1:
        x[:, 1] = -0.7 * left
    gate = ((np.arange(frames) // QUANTUM) % 4 == 3).astype(np.float64)
    for c in range(2, channels):
        x[:, c] = gate * 0.1 * np.sin(2 * np.pi * (300.0 + 70.0 * c + 11.0 * s) * t)
    return x.astype(np.float32)


def packet_schedule(rng, total_frames):
    """capture packets with stalls: mostly sub-quantum PipeWire packets, now and then a backlog of several quanta + a remainder"""
    sizes, left = [], total_frames
    while left > 0:
        r = rng.random()
        if r < 0.55:
            n = int(rng.choice([64, 128, 256, 480]))
        elif r < 0.8:
            n = int(rng.integers(1, 4)) * QUANTUM + int(rng.integers(0, 200))      # 1 ... 3 quanta in one chunk
        else:
            n = int(rng.integers(4, 8)) * QUANTUM + int(rng.integers(0, 200))      # a stall: 1024-frame chunks, then the rest
        n = min(n, left)
        sizes.append(n)
        left -= n
    return sizes


def configs():
    return dict(spectrogram=SpectrogramConfig(fft_size=2048, hop_size=256, history_length=8192),
                spectrum=SpectrumConfig(fft_size=2048, hop_size=512),
                loudness=LoudnessConfig(),
                stereometer=StereometerConfig(analyze_bands=True, correlation_window=0.05, segment_duration=0.02, target_sample_count=300),
                oscilloscope=OscilloscopeConfig(segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2, trigger_source=capi.CH_LEFT,
                                                channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT))


class OracleCapture:
    """one VisualManager's processors (the oracle's single-stream handles), fed whole chunks"""

    def __init__(self, oracle, cfgs, channels, positions):
        self.sg = SpectrogramProcessor(oracle, cfgs["spectrogram"])
        self.sp = SpectrumProcessor(oracle, cfgs["spectrum"])
        self.ld = LoudnessProcessor(oracle, cfgs["loudness"])
        self.st = StereometerProcessor(oracle, cfgs["stereometer"])
        self.sc = OscilloscopeProcessor(oracle, cfgs["oscilloscope"])
        self.channels, self.positions = channels, positions
        self.epoch_base = []

    def reset_audio(self):
        for p in (self.sg, self.sp, self.ld, self.st, self.sc):
            p.reset_audio()

    def ingest(self, chunk):
        blk = lambda: AudioBlock(np.ascontiguousarray(chunk).reshape(-1), self.channels, FS, self.positions)
        return dict(sg=self.sg.process_block(blk()), sp=self.sp.process_block(blk()), ld=self.ld.process_block(blk()),
                    st=self.st.process_block(blk()), sc=self.sc.process_block(blk()), capture=self.sc.last_capture(),
                    rate=self.sc.last_cycle_rate())


def check_capture(tag, want, got, max_step, stats):
    """`got`: the group's outputs for this capture and this chunk (numpy views), `want`: the oracle's snapshots for the same chunk"""
    from test_gpu_parity import check_trace
    from test_gpu_parity_meters import check_stable_trace
    # ---- loudness: one snapshot per chunk; true peak is the maximum over the WHOLE chunk (one take, :301)
    snap = got["loudness"]
    w = want["ld"]
    bar("group chunks: |d short-term LUFS|", abs(snap[0] - w.short_term_loudness), 1e-4, tag)
    bar("group chunks: |d momentary LUFS|", abs(snap[1] - w.momentary_loudness), 1e-4, tag)
    bar("group chunks: |d rms fast dB|", np.abs(snap[2:10] - w.rms_fast_db).max(), 1e-4, tag)
    bar("group chunks: |d rms slow dB|", np.abs(snap[10:18] - w.rms_slow_db).max(), 1e-4, tag)
    bar("group chunks: |d true peak dB| (one take per chunk)", np.abs(snap[18:26] - w.true_peak_db).max(), 1e-4, tag)
    # ---- oscilloscope: one trigger evaluation per chunk
    hdr, samples = got["scope_header"], got["scope_samples"]
    sc = want["sc"]
    assert bool(hdr["produced"]) == (sc is not None), tag
    assert bool(hdr["locked"]) == (want["rate"] is not None), tag
    if sc is not None:
        assert (hdr["channels"], hdr["samples_per_channel"]) == (sc.channels, sc.samples_per_channel), tag
        base = want["epoch_base"]                      # epochs count resets / rebuilds from different origins: compare the advance
        if not base:
            base.extend([int(sc.epoch), int(got["scope_epoch"])])
        assert int(got["scope_epoch"]) - base[1] == int(sc.epoch) - base[0], (tag, got["scope_epoch"], sc.epoch, base)
        if want["rate"] is not None and want["capture"] is not None:
            n = int(hdr["samples_per_channel"])
            mine = np.concatenate([samples[c, :n] for c in range(int(hdr["channels"]))])
            rel = abs(hdr["period"] - FS / want["rate"]) / hdr["period"]
            check_stable_trace("group chunks: oscilloscope", mine, sc.samples, (int(hdr["capture_start"]), float(hdr["capture_frac"])),
                               want["capture"], max_step, n, rel, tag, period=float(hdr["period"]), span=2.0 * float(hdr["period"]))
            stats["scope"] += 1
    # ---- spectrum: the snapshot of the chunk's last hop
    sp = want["sp"]
    assert (got["spectrum_hops"] > 0) == (sp is not None), tag
    if sp is not None:
        for t in range(2):
            for k in range(2):
                check_trace(got["spectrum"][t, k], sp.traces[t][k])
        stats["spectrum"] += 1
    # ---- stereometer
    st = want["st"]
    assert bool(got["stereo_produced"]) == (st is not None), tag
    if st is not None:
        bar("group chunks: |d rho|", np.abs(got["stereo_rho"] - st.correlations).max(), 1e-6, tag)
        full = np.asarray(st.points[0], np.float32).reshape(-1, 2)
        assert np.array_equal(got["stereo_points"][0][:full.shape[0]].view(np.uint32), full.view(np.uint32)), tag
        stats["stereo"] += 1
    # ---- spectrogram: as many columns as the oracle emitted for the chunk
    n_cols = len(want["sg"].new_columns) if want["sg"] is not None else 0
    assert got["spectrogram_columns"] == n_cols, (tag, got["spectrogram_columns"], n_cols)


def header_fields(raw):
    """omx_oscilloscope_block_header as 10 i32 words"""
    f = raw.view(np.float32)
    return dict(produced=int(raw[0]), channels=int(raw[1]), samples_per_channel=int(raw[4]), locked=int(raw[5]), period=float(f[6]),
                capture_start=int(raw[7]), capture_frac=float(f[8]))


@pytest.mark.parametrize("channels", [2, 8])
def test_lock_step_group_ingest_is_one_block_per_batcher_chunk(omx, oracle, channels):
    """S captures behind ONE packet schedule (lock step): the product's batcher cuts the timeline into chunks; each chunk goes to
    omx_capture_group_ingest as one call and must equal the oracle handles fed the chunk whole.  Control: the same 1024-frame chunk fed
    to an oracle as 4 x 256 gives a different oscilloscope state (several trigger updates), so the test can tell the two partitions apart."""
    import torch
    from openmeters_amd.pipeline import CaptureGroup
    from test_gpu_fullsize import dview
    rng = np.random.default_rng(77 + channels)
    S, total = 3, QUANTUM * 150
    positions = capi.SURROUND if channels == 8 else capi.positions_fallback(channels)
    pcm = np.stack([capture_pcm(s, total, channels, rng) for s in range(S)])
    cfgs = configs()
    group = CaptureGroup(omx, S, **cfgs)                      # block_frames = 0: the reference's partition
    refs = [OracleCapture(oracle, cfgs, channels, positions) for _ in range(S)]
    f = fmt(channels, FS, 1)
    f.positions = (capi.C.c_uint8 * 8)(*positions)
    batcher = Batcher(omx)
    at, lengths = 0, []
    for n in packet_schedule(rng, total):
        batcher.push(pcm[0, at:at + n].reshape(-1), f)
        at += n
    chunks = [len(b) // channels for b in batcher.blocks]
    assert set(chunks) >= {256, 512, 768, 1024}, sorted(set(chunks))
    stats = dict(scope=0, spectrum=0, stereo=0)
    max_step = [float(np.abs(np.diff(pcm[s, :, 0])).max()) for s in range(S)]
    pos = 0
    bins = cfgs["spectrum"].fft_size // 2 + 1
    for ci, n in enumerate(chunks):
        assert np.array_equal(batcher.blocks[ci], pcm[0, pos:pos + n].reshape(-1))          # the batcher's chunk IS the timeline slice
        chunk = torch.from_numpy(np.ascontiguousarray(pcm[:, pos:pos + n])).to("cuda:0")
        u = group.ingest(chunk.data_ptr(), n, channels, FS, positions)
        torch.cuda.synchronize()
        assert int(u.n_blocks) == 1 and int(u.block_frames) == n, (n, int(u.n_blocks), int(u.block_frames))
        loud = dview(torch, u.d_loudness, (S, 1, 30), "<f4").cpu().numpy()
        hdrs = dview(torch, u.oscilloscope.d_headers, (S, 1, 10)).cpu().numpy()
        smps = dview(torch, u.oscilloscope.d_samples, (S, 2, 4096), "<f4").cpu().numpy()
        rho = dview(torch, u.stereometer.d_correlations, (S, 1, 4), "<f4").cpu().numpy()
        prod = dview(torch, u.stereometer.d_produced, (S, 1)).cpu().numpy()
        target = int(u.stereometer.target)
        pts = dview(torch, u.stereometer.d_points, (S, 4, target, 2), "<f4").cpu().numpy() if u.stereometer.d_points and target else None
        spec = dview(torch, u.spectrum.d_traces, (S, 1, 2, 2, bins), "<f4").cpu().numpy() if (u.produced & capi.VISUAL_SPECTRUM) else None
        n_cols = int(u.spectrogram.n_columns) if (u.produced & capi.VISUAL_SPECTROGRAM) else 0
        for s in range(S):
            want = refs[s].ingest(pcm[s, pos:pos + n])
            want["epoch_base"] = refs[s].epoch_base
            got = dict(loudness=loud[s, 0], scope_header=header_fields(hdrs[s, 0]), scope_samples=smps[s], scope_epoch=int(u.oscilloscope.epoch),
                       spectrum=spec[s, 0] if spec is not None else None,
                       spectrum_hops=int(u.spectrum.n_hops) if spec is not None else 0, stereo_produced=int(prod[s, 0]), stereo_rho=rho[s, 0],
                       stereo_points=pts[s] if pts is not None else None, spectrogram_columns=n_cols)
            check_capture((channels, ci, n, s), want, got, max_step[s], stats)
        pos += n
    assert stats["scope"] > 40 and stats["spectrum"] > 40 and stats["stereo"] > 40, stats
    # control: ONE 1024-frame block is not four 256-frame blocks for the trigger (period smoothing / reference EMA run once per block)
    a, b = OscilloscopeProcessor(oracle, cfgs["oscilloscope"]), OscilloscopeProcessor(oracle, cfgs["oscilloscope"])
    differ = 0
    for k in range(0, QUANTUM * 120, 1024):
        blk = pcm[0, k:k + 1024]
        a.process_block(AudioBlock(blk.reshape(-1), channels, FS, positions))
        for q in range(4):
            b.process_block(AudioBlock(blk[q * 256:(q + 1) * 256].reshape(-1), channels, FS, positions))
        differ += int(a.last_capture() != b.last_capture())
    assert differ > 0


@pytest.mark.parametrize("channels", [2, 8])
def test_ragged_group_ingest_is_one_block_per_capture_chunk(omx, oracle, channels):
    """Every capture has its OWN batcher and packet schedule (one VisualManager per capture): in one omx_capture_group_ingest_ragged call
    capture 0 may deliver a 1024-frame catch-up chunk, capture 1 a regular 256-frame one, capture 2 nothing.  Each delivers ONE block:
    per capture every output must equal the oracle handles fed the same chunks whole; one capture is reset on its own mid-way."""
    import torch
    from openmeters_amd.pipeline import CaptureGroup
    from test_gpu_fullsize import dview
    rng = np.random.default_rng(177 + channels)
    S, total, cap = 4, QUANTUM * 120, 1024
    positions = capi.SURROUND if channels == 8 else capi.positions_fallback(channels)
    pcm = np.stack([capture_pcm(s, total, channels, rng) for s in range(S)])
    cfgs = configs()
    group = CaptureGroup(omx, S, stats=True, **cfgs)   # + the per-capture summary rows (OMX_OPT_GROUP_STATS)
    refs = [OracleCapture(oracle, cfgs, channels, positions) for _ in range(S)]
    holds = [capi.peak_holds_reset(oracle, 3, 0.0) for _ in range(S)]   # the oracle's PeakHolds, on every capture's own sample clock
    clocks = [0.0] * S
    rows_seen = 0
    f = fmt(channels, FS, 1)
    f.positions = (capi.C.c_uint8 * 8)(*positions)
    queues = []
    for s in range(S):
        b = Batcher(omx)
        at = 0
        for n in packet_schedule(rng, total):
            b.push(pcm[s, at:at + n].reshape(-1), f)
            at += n
        queues.append([len(x) // channels for x in b.blocks])
    assert {256, 512, 768, 1024} <= set(sum(queues, []))
    stats = dict(scope=0, spectrum=0, stereo=0)
    max_step = [float(np.abs(np.diff(pcm[s, :, 0])).max()) for s in range(S)]
    pos = [0] * S
    bins = cfgs["spectrum"].fft_size // 2 + 1
    call = mixed = 0
    while any(queues):
        frames = np.zeros(S, np.uint32)
        for s in range(S):
            if queues[s] and rng.random() < 0.8:       # a capture whose batcher had nothing ready sits the call out
                frames[s] = queues[s].pop(0)
        mask = np.zeros(S, np.uint8)
        if call == 25:
            mask[1] = 1                                # capture 1's VisualManager::reset_audio, alone
        if not frames.any() and not mask.any():
            continue
        mixed += int(len(set(int(x) for x in frames if x)) > 1)
        host = np.zeros((S, cap, channels), np.float32)
        for s in range(S):
            host[s, :frames[s]] = pcm[s, pos[s]:pos[s] + int(frames[s])]
        d = torch.from_numpy(host).to("cuda:0")
        u = group.ingest_ragged(d.data_ptr(), cap, frames, channels, FS, positions, reset_mask=mask)
        torch.cuda.synchronize()
        assert int(u.block_frames) == 0 and int(u.max_blocks) == 1
        assert u.ingest_launches == 1   # Spectrogram and Spectrum: one projection of the block for both, per-capture counts included
        nb = dview(torch, u.loudness.d_n_blocks, (S,)).cpu().numpy()
        assert np.array_equal(nb, (frames != 0).astype(nb.dtype))
        loud = dview(torch, u.loudness.d_snapshots, (S, 1, 30), "<f4").cpu().numpy()
        hdrs = dview(torch, u.oscilloscope.d_headers, (S, 1, 10)).cpu().numpy()
        smps = dview(torch, u.oscilloscope.d_samples, (S, 2, 4096), "<f4").cpu().numpy()
        epochs = dview(torch, u.oscilloscope.d_epochs, (S, 2)).cpu().numpy()[:, 0]
        rho = dview(torch, u.stereometer.d_correlations, (S, 1, 4), "<f4").cpu().numpy()
        prod = dview(torch, u.stereometer.d_produced, (S, 1)).cpu().numpy()
        target = int(u.stereometer.target)
        pts = dview(torch, u.stereometer.d_points, (S, 4, target, 2), "<f4").cpu().numpy()
        hops = dview(torch, u.spectrum.d_n_hops, (S,)).cpu().numpy()
        spec = dview(torch, u.spectrum.d_traces, (S, int(u.spectrum.n_hops_out), 2, 2, bins), "<f4").cpu().numpy() if u.spectrum.d_traces else None
        cols = dview(torch, u.spectrogram.d_n_columns, (S,)).cpu().numpy() if u.spectrogram.d_n_columns else np.zeros(S, np.int32)
        assert u.d_stats_rows
        table = dview(torch, u.d_stats_rows, (S, 12), "<f4").cpu().numpy().copy()
        for s in range(S):
            if mask[s]:
                refs[s].reset_audio()
                holds[s] = capi.peak_holds_reset(oracle, 3, 0.0)   # LoudnessState::reset_audio: fresh holds, fresh clock
                clocks[s] = 0.0
            if not frames[s]:
                assert table[s, 7] == 0 and (call == 0 or np.array_equal(table[s, :7], last_table[s, :7])), (call, s)   # the row stays
                continue
            n = int(frames[s])
            want = refs[s].ingest(pcm[s, pos[s]:pos[s] + n])
            # ---- the capture's summary row against its own oracle processors
            meter = capi.loudness_meters(oracle, [want["ld"]], 1, capi.METER_TRUE_PEAK, capi.METER_LUFS_SHORT_TERM, clocks[s], n / FS, holds[s])
            clocks[s] += n / FS
            bar("group chunks: stats row, |d momentary / short-term LUFS|", max(abs(table[s, 0] - want["ld"].momentary_loudness),
                                                                                abs(table[s, 1] - want["ld"].short_term_loudness)), 1e-4, (call, s))
            bar("group chunks: stats row, |d max true peak dB|", abs(table[s, 2] - want["ld"].true_peak_db[:channels].max()), 1e-4, (call, s))
            bar("group chunks: stats row, |d held true-peak bars|", np.abs(table[s, 10:12] - meter[0, -1]["peaks"][:2]).max(), 1e-4, (call, s))
            if want["st"] is not None:
                bar("group chunks: stats row, |d rho|", np.abs(table[s, 3:7] - want["st"].correlations).max(), 1e-6, (call, s))
            n_cols = len(want["sg"].new_columns) if want["sg"] is not None else 0
            assert table[s, 7] == n_cols, (call, s, table[s, 7], n_cols)
            if n_cols:
                cnt = [len(c) for c in want["sg"].new_columns]
                assert abs(table[s, 8] - np.mean(cnt)) < 4.5 and abs(table[s, 9] - cnt[-1]) <= 4, (call, s, table[s, 7:10], cnt)
            rows_seen += 1
            want["epoch_base"] = refs[s].epoch_base
            got = dict(loudness=loud[s, 0], scope_header=header_fields(hdrs[s, 0]), scope_samples=smps[s], scope_epoch=int(epochs[s]),
                       spectrum=spec[s, 0] if spec is not None else None, spectrum_hops=int(hops[s]),
                       stereo_produced=int(prod[s, 0]), stereo_rho=rho[s, 0], stereo_points=pts[s], spectrogram_columns=int(cols[s]))
            check_capture((channels, call, n, s), want, got, max_step[s], stats)
            pos[s] += n
        last_table = table
        call += 1
    assert mixed > 10 and rows_seen > 60, (mixed, rows_seen)   # calls in which captures delivered chunks of different lengths
    assert stats["scope"] > 60 and stats["spectrum"] > 60 and stats["stereo"] > 60, stats


def test_per_capture_reset_reaches_a_disabled_visual(omx):
    """VisualManager::reset_audio resets EVERY entry's module, enabled or not (registry.rs:360-365).  A capture that
    is reset while the stereometer / oscilloscope / waveform / loudness visuals are disabled must come back, once they are enabled again,
    from a cleared state: the group equals twin banks that got the reset with their next call."""
    import torch
    from openmeters_amd import banks
    from openmeters_amd.pipeline import CaptureGroup
    from test_gpu_fullsize import dview
    rng = np.random.default_rng(4242)
    S, cap, channels = 4, 1024, 2
    pos = capi.positions_fallback(2)
    cfgs = configs()
    cfgs["waveform"] = capi.WaveformConfig(analyze_bands=True)
    pcm = np.stack([capture_pcm(s, cap * 40, channels, rng) for s in range(S)])
    group = CaptureGroup(omx, S, **cfgs)
    ld, st = banks.LoudnessBank(omx, cfgs["loudness"], S, 2), banks.StereometerBank(omx, cfgs["stereometer"], S)
    sc, wf = banks.OscilloscopeBank(omx, cfgs["oscilloscope"], S), banks.WaveformBank(omx, cfgs["waveform"], S)
    off = (capi.VISUAL_LOUDNESS, capi.VISUAL_STEREOMETER, capi.VISUAL_OSCILLOSCOPE, capi.VISUAL_WAVEFORM)
    at = [0] * S
    missed = np.zeros(S, np.uint8)
    enabled = True
    for call in range(30):
        if call == 10:
            for v in off:
                group.set_enabled(v, False)
            enabled = False
        if call == 16:
            for v in off:
                group.set_enabled(v, True)
            enabled = True
        frames = (rng.integers(1, 5, S) * QUANTUM).astype(np.uint32)
        mask = np.zeros(S, np.uint8)
        if call in (12, 14):
            mask[2 if call == 12 else 0] = 1          # resets that arrive while the four visuals are disabled
        if call == 20:
            mask[3] = 1
        host = np.zeros((S, cap, channels), np.float32)
        for s in range(S):
            host[s, :frames[s]] = pcm[s, at[s]:at[s] + int(frames[s])]
            at[s] += int(frames[s])
        d = torch.from_numpy(host).to("cuda:0")
        u = group.ingest_ragged(d.data_ptr(), cap, frames, channels, FS, pos, reset_mask=mask)
        if not enabled:
            missed |= mask
            torch.cuda.synchronize()
            assert not (u.produced & (capi.VISUAL_LOUDNESS | capi.VISUAL_STEREOMETER | capi.VISUAL_OSCILLOSCOPE | capi.VISUAL_WAVEFORM))
            continue
        m = mask | missed
        missed[:] = 0
        r_ld = ld.process_chunks(d.data_ptr(), cap, frames, 2, FS, pos, m)
        r_st = st.process_chunks(d.data_ptr(), cap, frames, 2, FS, pos, m)
        r_sc = sc.process_chunks(d.data_ptr(), cap, frames, 2, FS, pos, m)
        r_wf = wf.process_ragged(d.data_ptr(), cap, frames, 2, FS, pos, m)
        torch.cuda.synchronize()
        assert torch.equal(dview(torch, u.loudness.d_snapshots, (S, 1, 30)), dview(torch, r_ld.d_snapshots, (S, 1, 30))), call
        assert torch.equal(dview(torch, u.stereometer.d_correlations, (S, 1, 4)), dview(torch, r_st.d_correlations, (S, 1, 4))), call
        assert torch.equal(dview(torch, u.stereometer.d_produced, (S, 1)), dview(torch, r_st.d_produced, (S, 1))), call
        assert torch.equal(dview(torch, u.oscilloscope.d_headers, (S, 1, 10)), dview(torch, r_sc.d_headers, (S, 1, 10))), call
        assert torch.equal(dview(torch, u.oscilloscope.d_epochs, (S, 2)), dview(torch, r_sc.d_epochs, (S, 2))), call
        assert torch.equal(dview(torch, u.waveform.d_n_columns, (S,)), dview(torch, r_wf.d_n_columns, (S,))), call
        if call == 16:
            # the stereometer of captures 0 and 2 restarts from an empty history (20 ms = 960 frames): a stale one would produce at once
            produced = dview(torch, u.stereometer.d_produced, (S, 1)).cpu().numpy()[:, 0]
            for s in (0, 2):
                assert produced[s] == (1 if frames[s] >= 960 else 0), (s, int(frames[s]), produced)
            assert produced[1] == 1 and produced[3] == 1
