"""omx_batcher_bank_*: one DspBatcher per capture (reference src/meter.rs:27-80) with the samples resident on the device — SURVEY §8f
rank 1's "device-side batching".  The checker is the product's HOST batcher (omx_batcher_push, batcher.cpp), which the KATs ported from
meter.rs:194-276 pin on both libraries (tests/test_kat_batcher.py): every capture of the bank must emit the chunks its own host batcher
emits for the same packets — the same lengths in the same order, the same samples bit for bit — and keep the same pending samples."""
import numpy as np
import pytest

from openmeters_amd import capi
from test_kat_batcher import Batcher, fmt

pytestmark = pytest.mark.gpu


def packets_for(rng, S, max_frames, quiet):
    """one push: a packet length per capture — PipeWire quanta, odd sizes, stalls that deliver several batches at once, nothing at all"""
    n = np.zeros(S, np.uint32)
    for s in range(S):
        u = rng.random()
        if u < quiet:
            n[s] = 0
        elif u < 0.5:
            n[s] = int(rng.choice([128, 256, 441, 480, 512, 1024]))
        elif u < 0.85:
            n[s] = int(rng.integers(1, 700))
        else:
            n[s] = int(rng.integers(700, max_frames + 1))
    return n


@pytest.mark.parametrize("channels,rate,S", [(2, 48000.0, 37), (1, 44100.0, 5), (6, 96000.0, 9), (8, 22050.0, 3)])
def test_every_capture_emits_the_chunks_of_its_own_host_batcher(omx, channels, rate, S):
    import torch
    from openmeters_amd.pipeline import BatcherBank
    rng = np.random.default_rng(4100 + channels)
    max_frames = 4096
    positions = capi.positions_fallback(channels)
    bank = BatcherBank(omx, S, max_frames)
    hosts = [Batcher(omx) for _ in range(S)]
    f = fmt(channels, rate, 1)
    seen, multi = set(), 0
    for push in range(40):
        n = packets_for(rng, S, max_frames, quiet=0.15)
        clear = np.zeros(S, np.uint8)
        if push in (11, 23):
            clear[rng.integers(S)] = 1          # one capture's DspBatcher::clear (its own reset) before its packet
        stride = int(max(n.max(), 1)) + int(rng.integers(0, 9))
        host = rng.uniform(-1.0, 1.0, (S, stride, channels)).astype(np.float32)
        d = torch.from_numpy(host).to("cuda:0")
        rounds = bank.push(d.data_ptr(), stride, n, channels, rate, positions, generation=1, clear_mask=clear)
        torch.cuda.synchronize()
        got = [[] for _ in range(S)]
        for ptr, cap, frames in rounds:
            from test_gpu_fullsize import dview
            buf = dview(torch, ptr, (S, cap, channels), "<f4").cpu().numpy()
            for s in range(S):
                if frames[s]:
                    got[s].append(buf[s, :frames[s]].reshape(-1).copy())
        for s in range(S):
            if clear[s]:
                omx.fn("batcher_clear", None, [capi.C.c_void_p])(hosts[s].h)
            hosts[s].blocks = []
            hosts[s].push(host[s, :n[s]].reshape(-1), f)
            want = hosts[s].blocks
            assert [len(x) for x in got[s]] == [len(x) for x in want], (push, s, n[s])
            for a, b in zip(got[s], want):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (push, s)
            seen |= {len(x) // channels for x in want}
            multi += len(want) > 1
            assert np.array_equal(bank.pending(s).view(np.uint32), hosts[s].pending().view(np.uint32)), (push, s)
    batch, chunk = max(int(round(256 * rate / 48000.0)), 1), max(int(round(1024 * rate / 48000.0)), 1)
    # regular blocks, catch-up chunks up to the cap (470 frames at 22.05 kHz: not a multiple of the 118-frame batch), several chunks from one packet
    assert {batch, 2 * batch, chunk} <= seen and max(seen) == chunk and multi > 10


def test_a_format_change_drops_every_pending_sample_and_rescales_the_batch(omx):
    """DspBatcher::push :46-48 (the pending samples of the old format are dropped) and :20-25 (batch and chunk scale with the rate)"""
    import torch
    from openmeters_amd.pipeline import BatcherBank
    from test_gpu_fullsize import dview
    S = 4
    bank = BatcherBank(omx, S, 2048)
    pos2, pos1 = capi.positions_fallback(2), capi.positions_fallback(1)
    x = torch.rand((S, 300, 2), device="cuda:0")
    assert bank.push(x.data_ptr(), 300, [100, 255, 256, 300], 2, 48000.0, pos2, generation=1)[0][2].tolist() == [0, 0, 256, 256]
    assert [len(bank.pending(s)) // 2 for s in range(S)] == [100, 255, 0, 44]
    y = torch.rand((S, 2048, 1), device="cuda:0")
    rounds = bank.push(y.data_ptr(), 2048, [511, 512, 2048, 0], 1, 96000.0, pos1, generation=2)   # batch 512, chunk 2048 at 96 kHz
    assert [r[2].tolist() for r in rounds] == [[0, 512, 2048, 0]] and rounds[0][1] == 2048
    assert [len(bank.pending(s)) for s in range(S)] == [511, 0, 0, 0]
    torch.cuda.synchronize()
    buf = dview(torch, rounds[0][0], (S, 2048, 1), "<f4").cpu().numpy()
    assert np.array_equal(buf[2, :, 0], y[2, :, 0].cpu().numpy()) and np.array_equal(buf[1, :512, 0], y[1, :512, 0].cpu().numpy())
    assert np.array_equal(bank.pending(0), y[0, :511, 0].cpu().numpy())


@pytest.mark.parametrize("channels,rate", [(2, 48000.0), (1, 44100.0), (8, 96000.0)])
def test_silence_goes_through_every_capture_as_ingest_silence_feeds_it(omx, channels, rate):
    """ingest_silence (meter.rs:145-166): silence is pushed through DspBatcher::push in pieces of the scratch, or — beyond two seconds of it —
    the capture is reset.  Packets and silence alternate; chunks, remainders and resets against every capture's host batcher
    (omx_batcher_push_silence)."""
    import torch
    from openmeters_amd.pipeline import BatcherBank
    from test_gpu_fullsize import dview
    rng = np.random.default_rng(99 + channels)
    S = 11
    positions = capi.positions_fallback(channels)
    bank = BatcherBank(omx, S, 1024)
    hosts = [Batcher(omx) for _ in range(S)]
    f = fmt(channels, rate, 1)
    resets = chunks_with_samples = long_silences = 0
    for step in range(24):
        if step % 2 == 0:   # a packet round: leaves partial batches behind
            n = packets_for(rng, S, 1024, quiet=0.2)
            host = rng.uniform(-1.0, 1.0, (S, 1024, channels)).astype(np.float32)
            d = torch.from_numpy(host).to("cuda:0")
            bank.push(d.data_ptr(), 1024, n, channels, rate, positions, generation=1)
            for s in range(S):
                hosts[s].blocks = []
                hosts[s].push(host[s, :n[s]].reshape(-1), f)
            continue
        sil = np.zeros(S, np.uint64)
        for s in range(S):
            u = rng.random()
            sil[s] = 0 if u < 0.2 else (int(rng.integers(1, 300)) if u < 0.5 else (int(rng.integers(300, 40000)) if u < 0.9 else int(2 * rate) + int(rng.integers(1, 50))))
        rounds, reset = bank.push_silence(sil, channels, rate, positions, generation=1)
        torch.cuda.synchronize()
        got = [[] for _ in range(S)]
        for ptr, cap, frames in rounds:
            buf = dview(torch, ptr, (S, cap, channels), "<f4").cpu().numpy()
            for s in range(S):
                if frames[s]:
                    got[s].append(buf[s, :frames[s]].reshape(-1).copy())
        for s in range(S):
            hosts[s].blocks, hosts[s].resets = [], 0
            hosts[s].push_silence(int(sil[s]), f)
            assert int(reset[s]) == hosts[s].resets, (step, s, sil[s])
            resets += hosts[s].resets
            long_silences += len(hosts[s].blocks) > 20
            assert [len(x) for x in got[s]] == [len(x) for x in hosts[s].blocks], (step, s, sil[s])
            for k, (a, b) in enumerate(zip(got[s], hosts[s].blocks)):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (step, s, k)
                chunks_with_samples += bool(np.any(b != 0))
            assert np.array_equal(bank.pending(s).view(np.uint32), hosts[s].pending().view(np.uint32)), (step, s)
    assert resets >= 3 and chunks_with_samples >= 5 and long_silences >= 5


def test_rounds_feed_the_capture_group_like_the_host_batchers_chunks(omx):
    """end to end: packets -> omx_batcher_bank_push -> omx_capture_group_ingest_ragged per round, against a second group fed the host
    batchers' chunks through a host-assembled buffer: the summary rows and every spectrogram column agree bit for bit"""
    import torch
    from openmeters_amd.pipeline import BatcherBank, CaptureGroup
    from test_gpu_fullsize import dview
    rng = np.random.default_rng(77)
    S, channels, cap = 6, 2, 1024
    positions = capi.positions_fallback(channels)
    cfgs = dict(spectrogram=capi.SpectrogramConfig(fft_size=1024, hop_size=256, use_reassignment=True, history_length=64),
                loudness=capi.LoudnessConfig(), stereometer=capi.StereometerConfig(analyze_bands=True))
    a, b = CaptureGroup(omx, S, stats=True, **cfgs), CaptureGroup(omx, S, stats=True, **cfgs)
    bank = BatcherBank(omx, S, 2048)
    hosts = [Batcher(omx) for _ in range(S)]
    f = fmt(channels, 48000.0, 1)
    t = np.arange(2048)[None, :, None]
    calls = 0
    for push in range(24):
        n = packets_for(rng, S, 2048, quiet=0.2)
        stride = 2048
        host = (0.4 * np.sin(2 * np.pi * (220.0 + 40.0 * np.arange(S))[:, None, None] * (t + push * 977) / 48000.0) * np.array([1.0, -0.6])[None, None, :] +
                0.01 * rng.standard_normal((S, stride, channels))).astype(np.float32)
        d = torch.from_numpy(host).to("cuda:0")
        rounds = bank.push(d.data_ptr(), stride, n, channels, 48000.0, positions, generation=1)
        queues = []
        for s in range(S):
            hosts[s].blocks = []
            hosts[s].push(host[s, :n[s]].reshape(-1), f)
            queues.append(list(hosts[s].blocks))
        assert len(rounds) == max(len(q) for q in queues)
        for r, (ptr, capacity, frames) in enumerate(rounds):
            assert capacity == cap
            ua = a.ingest_ragged(ptr, capacity, frames, channels, 48000.0, positions)
            staged = np.zeros((S, cap, channels), np.float32)
            for s in range(S):
                if r < len(queues[s]):
                    blk = queues[s][r].reshape(-1, channels)
                    assert len(blk) == frames[s]
                    staged[s, :len(blk)] = blk
            ds = torch.from_numpy(staged).to("cuda:0")
            ub = b.ingest_ragged(ds.data_ptr(), cap, frames, channels, 48000.0, positions)
            torch.cuda.synchronize()
            ra = dview(torch, ua.d_stats_rows, (S, 12), "<f4").cpu().numpy()
            rb = dview(torch, ub.d_stats_rows, (S, 12), "<f4").cpu().numpy()
            assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32)), (push, r)
            if ua.spectrogram.d_n_columns:
                ca = dview(torch, ua.spectrogram.d_n_columns, (S,)).cpu().numpy()
                cb = dview(torch, ub.spectrogram.d_n_columns, (S,)).cpu().numpy()
                assert np.array_equal(ca, cb)
                mc = int(ua.spectrogram.max_columns)
                if mc:
                    na = dview(torch, ua.spectrogram.d_counts, (S, mc)).cpu().numpy()
                    nb = dview(torch, ub.spectrogram.d_counts, (S, mc)).cpu().numpy()
                    assert np.array_equal(na, nb), (push, r)
            calls += 1
    assert calls > 24


def test_batcher_bank_rejects_what_it_cannot_hold(omx):
    from openmeters_amd.pipeline import BatcherBank
    bank = BatcherBank(omx, 2, 512)
    with pytest.raises(Exception):
        bank.push(0, 512, [513, 0], 2, 48000.0, capi.positions_fallback(2))   # longer than max_packet_frames
    with pytest.raises(Exception):
        bank.push(0, 100, [200, 0], 2, 48000.0, capi.positions_fallback(2))   # longer than the stride
    with pytest.raises(Exception):
        bank.push(0, 512, [10, 0], 2, 48000.0, capi.positions_fallback(2))    # a packet without a buffer
    assert bank.push(0, 512, [0, 0], 2, 48000.0, capi.positions_fallback(2)) == []   # nothing arrived anywhere: no round, no launch
