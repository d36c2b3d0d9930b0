"""SURVEY §8 row a6, `copy_dc_removed_windowed_from_deque` (window.rs:66-88): the window's mean is a SEQUENTIAL f32 fold divided by the
length.  Until round 6 the fused classic / spectrum kernels took that sum as a tree, and on a hop whose constant offset dwarfs its signal
(0.5 + 0.02 u) the residue of the mean times the window's DC gain put bins 0 ... 2 up to 8e-5 of the trace maximum away from the oracle
— outside north_star's 1e-5, and carved out of the suite.  Since round 6 `window_sums_seq_kernel` (window_sum_kernels.hip) takes every
hop's sum in the reference's order ahead of the transform kernels.  Here: (1) the pre-pass against numpy's sequential f32 fold, bit for
bit, over aligned and odd shapes, ring wrap, hop > window; (2) offset hops through the spectrum and classic paths against the oracle at
the PLAIN bars (no allowance on the window's own lines), alone and paired with loud and quiet neighbours."""
import ctypes as C

import numpy as np
import pytest

from openmeters_amd import banks, capi
from openmeters_amd.capi import AudioBlock, SpectrogramConfig, SpectrogramProcessor, SpectrumConfig, SpectrumProcessor
from parity import bar, check_classic
from test_gpu_parity import check_trace

pytestmark = pytest.mark.gpu


def _window_sums(api, ring, tail, hop, window, n_hops):
    ring = np.ascontiguousarray(ring, np.float32)
    S, cap = ring.shape
    out = np.zeros((S, n_hops), np.float32)
    f = api.fn("debug_window_sums", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p])
    api.check(f(ring.ctypes.data, S, cap, tail, hop, window, n_hops, out.ctypes.data))
    return out


def _sequential_sums(ring, tail, hop, window, n_hops):
    S, cap = ring.shape
    out = np.zeros((S, n_hops), np.float32)
    for s in range(S):
        for h in range(n_hops):
            idx = (tail + h * hop + np.arange(window)) % cap
            out[s, h] = np.cumsum(ring[s, idx], dtype=np.float32)[-1]   # cumsum is a plain left fold: the reference's order
    return out


@pytest.mark.parametrize("window,hop,n_hops,tail", [
    (4096, 256, 70, 0),            # the benchmark's shape; 70 hops = one full wavefront of quads and a partial one
    (1024, 256, 9, 12345),         # unaligned ring positions (16-byte loads at 4-byte alignment)
    (1024, 1024, 5, 3 * 8192 - 700),   # hop = window, the walk wraps around the ring's end
    (512, 1700, 6, 77),            # hop > window: the steps between two windows are skipped
    (1000, 37, 13, 5),             # events inside a 64-sample tile (partial tiles)
    (16384, 1024, 3, 1),           # fewer hops than a quad
    (64, 16, 1, 0), (100, 300, 2, 8190), (4096, 4, 8, 2), (7, 3, 21, 0),
])
def test_window_sums_are_the_reference_s_sequential_f32_fold(omx, window, hop, n_hops, tail):
    rng = np.random.default_rng(window * 31 + hop)
    need = window + hop * (n_hops - 1)
    cap = 1 << max(int(np.ceil(np.log2(need + 1))), 13)
    ring = (0.5 + 0.02 * rng.uniform(-1, 1, (3, cap))).astype(np.float32)
    ring[1] = rng.standard_normal(cap).astype(np.float32) * np.float32(1e-3)     # signed, cancelling
    ring[2, ::97] = np.float32(3e4)                                              # steps that change the running sum's exponent
    got, want = _window_sums(omx, ring, tail, hop, window, n_hops), _sequential_sums(ring, tail, hop, window, n_hops)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got - want)


def test_window_sums_keep_non_finite_windows_to_themselves(omx):
    """a lane adds samples before its window opens and after it closes (window_sum_kernels.hip): an Inf / NaN there must not leak"""
    rng = np.random.default_rng(5)
    ring = rng.uniform(-1, 1, (1, 8192)).astype(np.float32)
    ring[0, 300] = np.inf
    ring[0, 2000] = np.nan
    got, want = _window_sums(omx, ring, 0, 256, 1024, 20), _sequential_sums(ring, 0, 256, 1024, 20)
    assert np.array_equal(got.view(np.uint32) & 0x7FC00000, want.view(np.uint32) & 0x7FC00000)   # (NaN payloads aside)
    finite = np.isfinite(want)
    assert finite.sum() >= 8 and np.array_equal(got[finite].view(np.uint32), want[finite].view(np.uint32))


def _offset_hop(rng, n):
    return (0.5 + 0.02 * rng.uniform(-1, 1, (n, 1)) * np.array([[1.0, 0.7]])).astype(np.float32)


def _offset_sequence(N, seed):
    """hop-sized pieces: offset hops alone, beside a loud hop and beside a quiet one (the fused kernels transform two hops at a time)"""
    rng = np.random.default_rng(seed)
    loud = rng.uniform(-1.0, 1.0, (N, 2)).astype(np.float32)
    quiet = (1e-4 * rng.uniform(-1.0, 1.0, (N, 2))).astype(np.float32)
    return np.concatenate([_offset_hop(rng, N), loud, quiet, _offset_hop(rng, N), _offset_hop(rng, N), _offset_hop(rng, N), loud,
                           _offset_hop(rng, N)])


@pytest.mark.parametrize("N", [1024, 4096, 16384])
def test_spectrum_hops_with_a_large_constant_offset_hold_the_plain_bars(omx, oracle, N):
    cfg = SpectrumConfig(fft_size=N, hop_size=N, floor_db=-140.0)
    pcm = _offset_sequence(N, 600 + N)
    n_hops = pcm.shape[0] // N
    bank = banks.SpectrumBank(omx, cfg, 1, emit_all_hops=True)
    up = bank.process_host(pcm[None], 2, 48000.0)
    assert up is not None and int(up.n_hops) == n_hops
    ref = SpectrumProcessor(oracle, cfg)
    for h in range(n_hops):
        w = ref.process_block(AudioBlock(pcm[h * N:(h + 1) * N].reshape(-1), 2, 48000.0))
        g = bank.fetch(0, h, N // 2 + 1)
        for wt in range(2):
            check_trace(g[0][wt], w.traces[0][wt], floor=-140.0)
            # the window's own lines, where the mean's residue lands: held to the bar on their own (relative to the trace maximum)
            px, py = 10.0 ** (g[0][wt][:4].astype(np.float64) / 10.0), 10.0 ** (w.traces[0][wt][:4].astype(np.float64) / 10.0)
            bar("spectrum, offset hops: |d 10^(dB/10)| / max on bins 0 ... 3", np.abs(px - py).max() / max(10.0 ** (w.traces[0][wt].max() / 10.0), 1e-6), 1e-5)


@pytest.mark.parametrize("N,hop", [(1024, 256), (4096, 256), (16384, 1024)])
def test_spectrum_overlapping_hops_of_an_offset_signal_hold_the_plain_bars(omx, oracle, N, hop):
    """overlapping windows (the benchmark's 4096 / 256 among them): every hop of one long offset signal, hop-sized blocks into the oracle"""
    rng = np.random.default_rng(N + hop)
    n_hops = 11
    pcm = _offset_hop(rng, N + hop * (n_hops - 1))
    cfg = SpectrumConfig(fft_size=N, hop_size=hop, floor_db=-140.0)
    bank = banks.SpectrumBank(omx, cfg, 1, emit_all_hops=True)
    up = bank.process_host(pcm[None], 2, 48000.0)
    assert up is not None and int(up.n_hops) == n_hops
    ref = SpectrumProcessor(oracle, cfg)
    w = ref.process_block(AudioBlock(pcm[:N].reshape(-1), 2, 48000.0))
    for h in range(n_hops):
        if h:
            w = ref.process_block(AudioBlock(pcm[N + (h - 1) * hop:N + h * hop].reshape(-1), 2, 48000.0))
        g = bank.fetch(0, h, N // 2 + 1)
        for wt in range(2):
            check_trace(g[0][wt], w.traces[0][wt], floor=-140.0)


@pytest.mark.parametrize("W,zp", [(1024, 1), (4096, 1), (16384, 1), (1024, 4), (2048, 32)])
def test_classic_columns_with_a_large_constant_offset_hold_the_plain_bars(omx, oracle, W, zp):
    """classic columns (processor.rs:350-380) of the same sequence: fused kernels (W = F), zero-padded (window sum over W < F) and the
    residue form beyond 16384 points — |d code| <= 1 within 40 dB of the maximum and the transform-noise budget everywhere else, bins
    0 ... 3 included (plain=True: no allowance)"""
    cfg = SpectrogramConfig(fft_size=W, hop_size=W, zero_padding_factor=zp, use_reassignment=False, history_length=8192)
    pcm = _offset_sequence(W, 900 + W + zp).reshape(-1)
    got = SpectrogramProcessor(omx, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    assert len(got.new_columns) == len(want.new_columns) == 8
    check_classic(got.new_columns, want.new_columns)


@pytest.mark.parametrize("W,hop", [(1024, 256), (4096, 256)])
def test_classic_overlapping_columns_of_an_offset_signal_hold_the_plain_bars(omx, oracle, W, hop):
    rng = np.random.default_rng(W * 3 + hop)
    pcm = _offset_hop(rng, W + hop * 14).reshape(-1)
    cfg = SpectrogramConfig(fft_size=W, hop_size=hop, use_reassignment=False, history_length=8192)
    got = SpectrogramProcessor(omx, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    assert len(got.new_columns) == len(want.new_columns) == 15
    check_classic(got.new_columns, want.new_columns)


@pytest.mark.parametrize("N,hop,block", [(16384, 1024, 256), (4096, 256, 256), (1024, 512, 100), (2048, 300, 256), (1024, 1024, 256)])
def test_spectrum_fed_block_by_block_carries_the_window_folds_between_calls(omx, oracle, N, hop, block):
    """the reference's cadence (one batcher block per call, meter.rs:40-69): the bank keeps every started window's running fold between
    calls (window_sums_carry_kernel) instead of walking W samples per completed hop.  An offset signal block by block, with a long call
    in the middle (the walk takes over and drops the carried folds) and a reset_audio, every snapshot against the oracle at the plain bars."""
    rng = np.random.default_rng(N + hop + block)
    total = N + hop * 9
    pcm = _offset_hop(rng, total + 8 * N)
    cfg = SpectrumConfig(fft_size=N, hop_size=hop, floor_db=-140.0)
    a, b = SpectrumProcessor(omx, cfg), SpectrumProcessor(oracle, cfg)
    produced = 0

    def feed(lo, hi):
        nonlocal produced
        blk = pcm[lo:hi].reshape(-1)
        g, w = a.process_block(AudioBlock(blk, 2, 48000.0)), b.process_block(AudioBlock(blk, 2, 48000.0))
        assert (g is None) == (w is None)
        if g is not None:
            produced += 1
            for wt in range(2):
                check_trace(g.traces[0][wt], w.traces[0][wt], floor=-140.0)

    pos = 0
    while pos + block <= total // 2:                 # block by block: carried folds
        feed(pos, pos + block)
        pos += block
    feed(pos, pos + 2 * N + 7)                         # one long call: many hops complete, the walk computes them
    pos += 2 * N + 7
    while pos + block <= total + 2 * N:              # back to blocks: the open windows are folded from the ring again, then carried
        feed(pos, pos + block)
        pos += block
    a.reset_audio()
    b.reset_audio()
    start = pos
    while pos + block <= start + N + 3 * hop:        # after a reset: nothing carried, fresh windows
        feed(pos, pos + block)
        pos += block
    assert produced >= 8


@pytest.mark.parametrize("W", [1024, 4096])
def test_classic_rectangular_window_bin_zero_is_a_plain_fixed_bar_case(omx, oracle, W):
    """soak seed 12072005 (round 4): with the rectangular window bin 0 of a DC-removed column is sum (x - mean) — the rounding of the mean
    and nothing else; with the tree-summed mean the two sides sat ten or more dB apart near -135 dB there and the suite carried a 1e-9
    allowance on bins 0 ... 3.  With the reference's fold on both sides the allowance is gone (tests/parity.py::check_classic)."""
    from test_gpu_parity import stream_pcm
    cfg = SpectrogramConfig(fft_size=W, hop_size=W // 4, window=capi.WINDOW_RECTANGULAR, use_reassignment=False, history_length=8192)
    pcm = stream_pcm(7, W + (W // 4) * 15).reshape(-1)
    got = SpectrogramProcessor(omx, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    want = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(pcm, 2, 48000.0))
    assert len(got.new_columns) == len(want.new_columns) == 16
    check_classic(got.new_columns, want.new_columns)


@pytest.mark.parametrize("N,hop", [(1024, 256), (4096, 256), (16384, 1024)])
def test_ragged_spectrum_bank_carries_the_folds_per_stream(omx, oracle, N, hop):
    """per-capture frame counts and resets (omx_spectrum_bank_process_ragged): spectrum_plan_kernel keeps every stream's carried folds on
    the device — carried while the stream moves by small pushes, walked after a long one or a reset.  Offset signals, every completed
    snapshot of every stream against its own oracle handle at the plain bars."""
    import torch
    rng = np.random.default_rng(N * 7 + hop)
    S, cap = 5, 1024
    cfg = SpectrumConfig(fft_size=N, hop_size=hop, floor_db=-140.0)
    bank = banks.SpectrumBank(omx, cfg, S)
    refs = [SpectrumProcessor(oracle, cfg) for _ in range(S)]
    feeds = [_offset_hop(rng, 6 * N + 40 * cap) for _ in range(S)]
    at = [0] * S
    pos = capi.positions_fallback(2)
    bins = N // 2 + 1
    compared = 0
    for call in range(int(2.5 * N / 256) + 24):
        frames = rng.choice([0, 100, 256, 256, 256, 512, 1024], S)
        if call == 5:
            frames[:] = 1024                                  # (everyone long: several hops at the small sizes)
        mask = (rng.random(S) < 0.05).astype(np.uint8)
        pcm = np.zeros((S, cap, 2), np.float32)
        for s in range(S):
            pcm[s, :frames[s]] = feeds[s][at[s]:at[s] + frames[s]]
        d_pcm = torch.from_numpy(pcm).to("cuda:0")
        up = bank.process_ragged(d_pcm.data_ptr(), cap, frames, 2, 48000.0, pos, mask)
        torch.cuda.synchronize()
        n_hops = torch.as_tensor(_DeviceView(up.d_n_hops, (S,), "<u4"), device="cuda:0").cpu().numpy()
        traces = torch.as_tensor(_DeviceView(up.d_traces, (S, up.n_hops_out, 2, 2, bins), "<f4"), device="cuda:0").cpu().numpy() if up.d_traces else None
        for s in range(S):
            if mask[s]:
                refs[s].reset_audio()
            w = refs[s].process_block(AudioBlock(pcm[s, :frames[s]].reshape(-1), 2, 48000.0)) if frames[s] else None
            at[s] += int(frames[s])
            assert (int(n_hops[s]) > 0) == (w is not None), (call, s, int(n_hops[s]))
            if w is not None:
                for wt in range(2):
                    check_trace(traces[s, 0, 0, wt], np.asarray(w.traces[0][wt]), floor=-140.0)
                compared += 1
    assert compared >= 3 * S


class _DeviceView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}
