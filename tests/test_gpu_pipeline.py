"""cfg5 shape on one GPU: the full pipeline helper feeds three banks from the same device PCM and assembles the per-stream
summary rows (what K8 all-gathers); every row is checked against the CPU oracle run stream by stream."""
import numpy as np
import pytest

from openmeters_amd import capi
from openmeters_amd.capi import (AudioBlock, LoudnessConfig, LoudnessProcessor, SpectrogramConfig, SpectrogramProcessor,
                                 StereometerConfig, StereometerProcessor)
from golden_inputs import cfg2_pcm

pytestmark = pytest.mark.gpu

from parity import check_chunked_rho, stereometer_band_rms


def test_full_pipeline_summary_rows_match_oracle(omx, oracle):
    import torch
    from openmeters_amd.pipeline import FullPipeline, gather_stats, stats_rows_tensor
    dev = torch.device("cuda", 0)
    S, frames = 6, 256 * 48
    pcm = np.stack([cfg2_pcm(s, frames) for s in range(S)])
    pcm[:, :, 1] *= -1.0  # anti-phase right channel: rho < 0
    d_pcm = torch.from_numpy(pcm).to(dev).contiguous()
    pipe = FullPipeline(omx, S)
    up = pipe.step(d_pcm.data_ptr(), frames, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert up.produced == capi.VISUAL_SPECTROGRAM | capi.VISUAL_LOUDNESS | capi.VISUAL_STEREOMETER and up.ingest_launches == 1
    table = gather_stats(stats_rows_tensor(torch, dev, up, S), S).cpu().numpy()
    assert table.shape == (S, 12)
    for s in range(S):
        snaps = []
        lp = LoudnessProcessor(oracle, LoudnessConfig())
        sp = StereometerProcessor(oracle, StereometerConfig(analyze_bands=True, correlation_window=0.05, segment_duration=0.02,
                                                            target_sample_count=2000))
        for k in range(0, frames, 256):
            blk = AudioBlock(pcm[s, k:k + 256].reshape(-1), 2, 48000.0)
            ls, ss = lp.process_block(blk), sp.process_block(blk)
            snaps.append(ls)
        holds = capi.peak_holds_reset(oracle, 3, 0.0)
        rows = capi.loudness_meters(oracle, snaps, 1, capi.METER_TRUE_PEAK, capi.METER_LUFS_SHORT_TERM, 0.0, 256.0 / 48000.0, holds)
        assert np.abs(table[s, 10:12] - rows[0, -1]["peaks"][:2]).max() < 1e-4 and table[s, 10] > -20.0
        sg = SpectrogramProcessor(oracle, SpectrogramConfig(fft_size=4096, hop_size=256, history_length=8192)).process_block(
            AudioBlock(pcm[s].reshape(-1), 2, 48000.0))
        counts = [len(c) for c in sg.new_columns]
        assert abs(table[s, 0] - ls.momentary_loudness) < 1e-4 and abs(table[s, 1] - ls.short_term_loudness) < 1e-4
        assert abs(table[s, 2] - ls.true_peak_db[:2].max()) < 1e-4
        # (48 blocks per call: the stereometer bank runs its chunk-parallel form by shape — the level-aware bars of that form, parity.py)
        check_chunked_rho(table[s, 3:7], ss.correlations, stereometer_band_rms(pcm[s]), s)
        assert table[s, 3] < -0.99
        assert table[s, 7] == len(counts) and abs(table[s, 8] - np.mean(counts)) < 0.5 and abs(table[s, 9] - counts[-1]) <= 4


def test_cfg5_full_shard_1024_streams_replication_shift_partition_and_oracle_rows(omx, oracle):
    """BASELINE configs[4], one GPU's shard at full size: 1024 x 2-ch streams through the capture group (the bench step:
    reassigned STFT on the main HIP stream, loudness + stereometer banks on the group's side streams), 2 x 16 384 frames.
      replication  stream s = distinct[s % 32] => rows / columns of every replica are BIT-identical (different workgroups, XCDs,
                   bank slots, side-stream interleavings)
      shift        distinct stream 1 carries distinct stream 0 advanced by one hop => column c equals column c + 1
      partition    the two concurrent steps == one serial 32 768-frame step on a fresh pipeline (spectrogram columns and
                   stereometer correlations bit-exact, loudness rows bit-exact)
      oracle       summary rows of 3 streams vs the CPU oracle run block by block (the 6-stream test's bars)"""
    import torch
    from openmeters_amd.pipeline import FullPipeline, stats_rows_tensor
    from test_gpu_fullsize import dview, spectrogram_outputs
    dev = torch.device("cuda", 0)
    S, D, frames, hop = 1024, 32, 16384, 256
    base0 = cfg2_pcm(3, 2 * frames + hop)
    distinct = [base0[:2 * frames], base0[hop:hop + 2 * frames]] + [cfg2_pcm(40 + d, 2 * frames) for d in range(2, D)]
    distinct = np.stack(distinct)
    distinct[:, :, 1] *= np.float32(-1.0)      # anti-phase right channel: rho < 0
    d_all = torch.from_numpy(distinct).to(dev).repeat(S // D, 1, 1).contiguous()      # [S][2 frames][2]
    pipe = FullPipeline(omx, S)
    outs = []
    for k in range(2):
        chunk = d_all[:, k * frames:(k + 1) * frames].contiguous()
        up = pipe.step(chunk.data_ptr(), frames, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        n_blocks = int(up.n_blocks)
        table = stats_rows_tensor(torch, dev, up, S)
        counts, points = spectrogram_outputs(torch, up.spectrogram)
        outs.append((counts, points, dview(torch, up.d_loudness, (S, n_blocks, 30)).clone(),
                     dview(torch, up.stereometer.d_correlations, (S, n_blocks, 4)).clone(), table.clone()))
    assert outs[0][0].shape == (S, (frames - 8192) // hop + 1) and outs[1][0].shape == (S, frames // hop)
    for counts, points, snaps, corr, table in outs:
        assert torch.equal(counts[:D].repeat(S // D, 1), counts)
        assert torch.equal(points[:D].repeat(S // D, 1, 1, 1), points)
        assert torch.equal(snaps[:D].repeat(S // D, 1, 1), snaps)
        assert torch.equal(corr[:D].repeat(S // D, 1, 1), corr)
        assert torch.equal(table[:D].repeat(S // D, 1), table)
        n = counts.shape[1]
        assert torch.equal(counts[1, :n - 1], counts[0, 1:]) and torch.equal(points[1, :n - 1], points[0, 1:])
    assert torch.equal(outs[0][0][1, -1], outs[1][0][0, 0]) and torch.equal(outs[0][1][1, -1], outs[1][1][0, 0])   # across the step boundary

    serial = FullPipeline(omx, S)
    up = serial.step(d_all.data_ptr(), 2 * frames, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    n_blocks = int(up.n_blocks)
    counts, points = spectrogram_outputs(torch, up.spectrogram)
    assert torch.equal(counts, torch.cat([outs[0][0], outs[1][0]], 1)) and torch.equal(points, torch.cat([outs[0][1], outs[1][1]], 1))
    assert torch.equal(dview(torch, up.d_loudness, (S, n_blocks, 30)), torch.cat([outs[0][2], outs[1][2]], 1))
    assert torch.equal(dview(torch, up.stereometer.d_correlations, (S, n_blocks, 4)), torch.cat([outs[0][3], outs[1][3]], 1))

    table = outs[1][4].cpu().numpy()
    for s in (0, 17, 1023):
        pcm = distinct[s % D]
        lp = LoudnessProcessor(oracle, LoudnessConfig())
        sp = StereometerProcessor(oracle, StereometerConfig(analyze_bands=True, correlation_window=0.05, segment_duration=0.02,
                                                            target_sample_count=2000))
        for k in range(0, 2 * frames, 256):
            blk = AudioBlock(pcm[k:k + 256].reshape(-1), 2, 48000.0)
            ls, ss = lp.process_block(blk), sp.process_block(blk)
        sg = SpectrogramProcessor(oracle, SpectrogramConfig(fft_size=4096, hop_size=256, history_length=8192)).process_block(
            AudioBlock(pcm[frames - 8192 + hop:].reshape(-1), 2, 48000.0))      # the second step's 64 columns
        cnt = [len(c) for c in sg.new_columns]
        assert len(cnt) == 64 and table[s, 7] == 64
        assert abs(table[s, 0] - ls.momentary_loudness) < 1e-4 and abs(table[s, 1] - ls.short_term_loudness) < 1e-4
        assert abs(table[s, 2] - ls.true_peak_db[:2].max()) < 1e-4
        from parity import check_chunked_rho, stereometer_band_rms
        check_chunked_rho(table[s, 3:7], ss.correlations, stereometer_band_rms(pcm), s)   # 1024 x 64 blocks: chunk-parallel form
        assert table[s, 3] < -0.99
        assert abs(table[s, 8] - np.mean(cnt)) < 0.5 and abs(table[s, 9] - cnt[-1]) <= 4
        n_last = int(outs[1][0][s, 63])
        got = outs[1][1][s, 63, :n_last].contiguous().view(torch.float32).cpu().numpy()   # the stream's newest column: (f, t, power) rows
        from parity import check_reassigned_columns, reassigned_column_metrics
        check_reassigned_columns([got], [sg.new_columns[-1]], 48000.0, hop)


def test_bench_two_ranks_on_one_gpu_over_gloo_runs_the_cfg5_step():
    """`python bench.py --gpus 2` (no RANK in the environment) on a 1-GPU box: the launcher starts two ranks, both run the
    cfg5 step (full pipeline, small shard) on device 0, the summary rows travel over gloo; one JSON line, n_gpus = 2, and
    `python bench.py --config cfg5` at N = 1 runs the same step."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(OMX_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    common = ["--steps", "3", "--warmup", "2", "--streams", "16", "--no-secondary", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + common, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["config"]["name"] == "cfg5" and two["value"] > 0 and two["roofline"]["kernel_ms"] > 0
    assert two["n1_same_workload"]["measured"]["value"] > 0 and two["weak_scaling_vs_measured_n1"] > 0   # the run's own N = 1 leg
    assert two["config"]["columns_per_step_per_gpu"] == 16 * 64
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "cfg5"] + common, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    one = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert one["n_gpus"] == 1 and one["config"]["name"] == "cfg5" and one["roofline"]["mean_points_per_frame"] > 1900


def test_capture_group_equals_separate_banks_plus_torch_rows(omx):
    """The capture group (one ingest call: fan-out, side streams and summary rows inside libomx_hip.so) against the same three banks
    run one after the other with the summary rows assembled by torch glue (tests/pipeline_reference.py): rows, spectrogram columns,
    loudness snapshots and correlations bit for bit, step after step (peak holds and clocks carried)."""
    import torch
    from openmeters_amd.pipeline import FullPipeline
    from pipeline_reference import SeparateBanks
    from test_gpu_fullsize import dview, spectrogram_outputs
    S, frames = 48, 4096
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    pcm = ((torch.rand((S, frames * 5, 2), device=dev, generator=g) - 0.5) * 0.8).contiguous()
    a, b = FullPipeline(omx, S), SeparateBanks(omx, S)
    for k in range(5):
        chunk = pcm[:, k * frames:(k + 1) * frames].contiguous()
        up_a, rows_a = a.step_with_stats(torch, dev, chunk.data_ptr(), frames)
        up_b, snaps, st, n_blocks = b.step(chunk.data_ptr(), frames, torch.cuda.current_stream().cuda_stream)
        rows_b = b.stats(torch, dev, up_b, snaps, st, n_blocks)
        torch.cuda.synchronize()
        assert bool(up_a.produced & capi.VISUAL_SPECTROGRAM) == (up_b is not None)
        assert torch.equal(rows_a.view(torch.int32), rows_b.view(torch.int32)), k
        assert torch.equal(dview(torch, up_a.d_loudness, (S, n_blocks, 30)), dview(torch, snaps, (S, n_blocks, 30)))
        assert torch.equal(dview(torch, up_a.stereometer.d_correlations, (S, n_blocks, 4)), dview(torch, st.d_correlations, (S, n_blocks, 4)))
        if up_b is not None:
            ca, pa = spectrogram_outputs(torch, up_a.spectrogram)
            cb, pb = spectrogram_outputs(torch, up_b)
            assert torch.equal(ca, cb) and torch.equal(pa, pb)
    assert float(rows_a[:, 7].max()) > 0


def test_capture_group_cfg2_one_ingest_launch_feeds_spectrogram_and_spectrum(omx):
    """BASELINE configs[1] through the capture group: Spectrogram{4096, 256, reassigned} + Spectrum{4096, 256, Mid} read the same
    block, so ONE projection launch fills both banks' rings (update.ingest_launches == 1); with the shared launch switched off
    (OMX_OPT_GROUP_SHARED_INGEST = 0: two launches) every output is bit-identical, and both equal the two banks run separately."""
    import torch
    from openmeters_amd import banks
    from openmeters_amd.pipeline import CaptureGroup
    from test_gpu_fullsize import dview, spectrogram_outputs
    dev = torch.device("cuda", 0)
    S, frames = 8, 8192 + 256 * 15
    pcm = torch.from_numpy(np.stack([cfg2_pcm(s, 3 * frames) for s in range(S)])).to(dev)
    pos = capi.positions_fallback(2)
    sg_cfg = SpectrogramConfig(fft_size=4096, hop_size=256, history_length=8192, use_reassignment=True)
    sp_cfg = capi.SpectrumConfig(fft_size=4096, hop_size=256, source=capi.CH_MID, secondary_source=capi.CH_NONE, floor_db=-100.0)
    shared = CaptureGroup(omx, S, spectrogram=sg_cfg, spectrum=sp_cfg)
    split = CaptureGroup(omx, S, spectrogram=sg_cfg, spectrum=sp_cfg)
    split.set_option(capi.OPT_GROUP_SHARED_INGEST, 0)
    sg, sp = banks.SpectrogramBank(omx, sg_cfg, S), banks.SpectrumBank(omx, sp_cfg, S)
    for k in range(3):
        chunk = pcm[:, k * frames:(k + 1) * frames].contiguous()
        ua = shared.ingest(chunk.data_ptr(), frames, 2, 48000.0, pos)
        ub = split.ingest(chunk.data_ptr(), frames, 2, 48000.0, pos)
        u_sg = sg.process_device(chunk.data_ptr(), frames, 2, 48000.0, pos)
        u_sp = sp.process_device(chunk.data_ptr(), frames, 2, 48000.0, pos)
        torch.cuda.synchronize()
        assert ua.ingest_launches == 1 and ub.ingest_launches == 2
        assert ua.produced == ub.produced == capi.VISUAL_SPECTROGRAM | capi.VISUAL_SPECTRUM
        ca, pa = spectrogram_outputs(torch, ua.spectrogram)
        for other in (ub.spectrogram, u_sg):
            cb, pb = spectrogram_outputs(torch, other)
            assert torch.equal(ca, cb) and torch.equal(pa, pb)
        bins = int(ua.spectrum.bins)
        ta = dview(torch, ua.spectrum.d_traces, (S, 1, 2, 2, bins))
        assert int(ua.spectrum.n_hops) == int(ub.spectrum.n_hops) == int(u_sp.n_hops) > 0
        assert torch.equal(ta, dview(torch, ub.spectrum.d_traces, (S, 1, 2, 2, bins))) and torch.equal(ta, dview(torch, u_sp.d_traces, (S, 1, 2, 2, bins)))


def test_capture_group_all_six_visuals_equal_their_banks(omx):
    """One block to EVERY visual (registry.rs:396-418): a group with all six enabled against six banks fed the same blocks —
    every output bit for bit, reset_audio included (registry.rs:360-365)."""
    import torch
    from openmeters_amd import banks
    from openmeters_amd.pipeline import CaptureGroup
    from test_gpu_fullsize import dview, spectrogram_outputs
    dev = torch.device("cuda", 0)
    S, frames = 5, 2048
    g = torch.Generator(device=dev).manual_seed(11)
    n = torch.arange(frames * 8, device=dev, dtype=torch.float64)
    pcm = torch.empty((S, frames * 8, 2), device=dev, dtype=torch.float32)
    for s in range(S):
        tone = (0.6 * torch.sin(2 * np.pi * (220.0 + 55.0 * s) * n / 48000.0)).to(torch.float32)
        pcm[s, :, 0] = tone + 0.01 * (torch.rand(frames * 8, device=dev, generator=g) - 0.5)
        pcm[s, :, 1] = -0.7 * tone
    pos = capi.positions_fallback(2)
    cfgs = dict(spectrogram=SpectrogramConfig(fft_size=2048, hop_size=64, history_length=8192), spectrum=capi.SpectrumConfig(fft_size=4096, hop_size=1024),
                loudness=LoudnessConfig(), stereometer=StereometerConfig(analyze_bands=True),
                oscilloscope=capi.OscilloscopeConfig(trigger_source=capi.CH_LEFT, channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT),
                waveform=capi.WaveformConfig(analyze_bands=True))
    group = CaptureGroup(omx, S, block_frames=256, **cfgs)
    sg, sp = banks.SpectrogramBank(omx, cfgs["spectrogram"], S), banks.SpectrumBank(omx, cfgs["spectrum"], S)
    ld, st = banks.LoudnessBank(omx, cfgs["loudness"], S, 2), banks.StereometerBank(omx, cfgs["stereometer"], S)
    sc, wf = banks.OscilloscopeBank(omx, cfgs["oscilloscope"], S), banks.WaveformBank(omx, cfgs["waveform"], S)
    for k in range(8):
        if k == 5:
            group.reset_audio()
            for b in (sg, sp, ld, st, sc, wf):
                b.reset_audio()
        chunk = pcm[:, k * frames:(k + 1) * frames].contiguous()
        u = group.ingest(chunk.data_ptr(), frames, 2, 48000.0, pos)
        r_sg = sg.process_device(chunk.data_ptr(), frames, 2, 48000.0, pos)
        r_sp = sp.process_device(chunk.data_ptr(), frames, 2, 48000.0, pos)
        r_ld = ld.process_device(chunk.data_ptr(), 256, frames // 256, 2, 48000.0, pos)
        r_st = st.process_device(chunk.data_ptr(), 256, frames // 256, 2, 48000.0, pos)
        sc.process_device(chunk.data_ptr(), 256, frames // 256, 2, 48000.0, pos)
        r_wf = wf.process_device(chunk.data_ptr(), frames, 2, 48000.0, pos)
        torch.cuda.synchronize()
        nb = frames // 256
        assert int(u.n_blocks) == nb and u.ingest_launches == 1
        assert bool(u.produced & capi.VISUAL_SPECTROGRAM) == (r_sg is not None)
        if r_sg is not None:
            ca, pa = spectrogram_outputs(torch, u.spectrogram)
            cb, pb = spectrogram_outputs(torch, r_sg)
            assert torch.equal(ca, cb) and torch.equal(pa, pb)
        assert bool(u.produced & capi.VISUAL_SPECTRUM) == (r_sp is not None)
        if r_sp is not None:
            bins = int(r_sp.bins)
            assert torch.equal(dview(torch, u.spectrum.d_traces, (S, 1, 2, 2, bins)), dview(torch, r_sp.d_traces, (S, 1, 2, 2, bins)))
        assert torch.equal(dview(torch, u.d_loudness, (S, nb, 30)), dview(torch, r_ld, (S, nb, 30)))
        assert torch.equal(dview(torch, u.stereometer.d_correlations, (S, nb, 4)), dview(torch, r_st.d_correlations, (S, nb, 4)))
        assert torch.equal(dview(torch, u.stereometer.d_produced, (S, nb)), dview(torch, r_st.d_produced, (S, nb)))
        for s in range(S):
            hdr, smp = sc.fetch(s, nb - 1, with_samples=True)
            got = dview(torch, u.oscilloscope.d_headers, (S, nb, 10))[s, nb - 1].cpu().numpy()
            assert got[0] == hdr.produced and got[4] == hdr.samples_per_channel and got[7] == hdr.capture_start, (k, s)
            if hdr.produced:
                mine = dview(torch, u.oscilloscope.d_samples, (S, 2, 4096), "<f4")[s, :hdr.channels, :hdr.samples_per_channel].cpu().numpy()
                assert np.array_equal(mine, smp[:hdr.channels, :hdr.samples_per_channel]), (k, s)
        assert int(u.waveform.n_columns) == int(r_wf.n_columns)
        if int(r_wf.n_columns):
            nc = int(r_wf.n_columns)
            assert torch.equal(dview(torch, u.waveform.d_columns, (S, nc, 4, 11)), dview(torch, r_wf.d_columns, (S, nc, 4, 11)))


def test_capture_group_ragged_ingest_with_toggles_and_config_changes_equals_its_banks(omx):
    """The group as VisualManager (registry.rs:266-277, :343-365, :396-418) with per-capture frame counts: all six visuals against six
    banks driven through their own ragged entry points with the same per-capture counts and reset flags — the device outputs of
    every call bit for bit — while visuals are switched off and on (a disabled one sits the call out and keeps its state), the
    spectrogram's hop and the stereometer's correlation window change mid-stream, one capture is reset alone, and a format-generation
    change resets everything (after which lock-step ingest works again until the next ragged call)."""
    import torch
    from openmeters_amd import banks
    from openmeters_amd.pipeline import CaptureGroup
    from test_gpu_fullsize import dview
    dev = torch.device("cuda", 0)
    S, block, maxb = 5, 256, 4
    cap = block * maxb
    rng = np.random.default_rng(2024)
    total = cap * 24
    n = torch.arange(total, device=dev, dtype=torch.float64)
    feed = torch.empty((S, total, 2), device=dev, dtype=torch.float32)
    for s in range(S):
        tone = (0.6 * torch.sin(2 * np.pi * (220.0 + 55.0 * s) * n / 48000.0) + 0.05 * torch.sin(2 * np.pi * 2500.0 * n / 48000.0)).to(torch.float32)
        feed[s, :, 0] = tone
        feed[s, :, 1] = -0.7 * tone
    pos = capi.positions_fallback(2)
    cfgs = dict(spectrogram=SpectrogramConfig(fft_size=2048, hop_size=256, history_length=8192), spectrum=capi.SpectrumConfig(fft_size=4096, hop_size=1024),
                loudness=LoudnessConfig(), stereometer=StereometerConfig(analyze_bands=True),
                oscilloscope=capi.OscilloscopeConfig(trigger_source=capi.CH_LEFT, channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT),
                waveform=capi.WaveformConfig(analyze_bands=True))
    group = CaptureGroup(omx, S, block_frames=block, **{k: v for k, v in cfgs.items() if k != "oscilloscope"})
    assert group.enabled() == capi.VISUAL_SPECTROGRAM | capi.VISUAL_SPECTRUM | capi.VISUAL_LOUDNESS | capi.VISUAL_STEREOMETER | capi.VISUAL_WAVEFORM
    group.update_config(capi.VISUAL_OSCILLOSCOPE, cfgs["oscilloscope"])   # stored: the bank does not exist yet
    sg, sp = banks.SpectrogramBank(omx, cfgs["spectrogram"], S), banks.SpectrumBank(omx, cfgs["spectrum"], S)
    ld, st = banks.LoudnessBank(omx, cfgs["loudness"], S, 2), banks.StereometerBank(omx, cfgs["stereometer"], S)
    sc, wf = banks.OscilloscopeBank(omx, cfgs["oscilloscope"], S), banks.WaveformBank(omx, cfgs["waveform"], S)
    on = dict(stereometer=True, oscilloscope=False, waveform=True)
    at = [0] * S
    generation = 3
    for call in range(16):
        if call == 2:
            group.set_enabled(capi.VISUAL_OSCILLOSCOPE, True)     # created and prepared here
            on["oscilloscope"] = True
        if call == 4:
            group.set_enabled(capi.VISUAL_STEREOMETER, False)
            group.set_enabled(capi.VISUAL_WAVEFORM, False)
            on["stereometer"] = on["waveform"] = False
        if call == 7:
            group.set_enabled(capi.VISUAL_STEREOMETER, True)
            group.set_enabled(capi.VISUAL_WAVEFORM, True)
            on["stereometer"] = on["waveform"] = True
        if call == 6:
            cfgs["spectrogram"] = SpectrogramConfig(fft_size=2048, hop_size=128, history_length=8192)
            group.update_config(capi.VISUAL_SPECTROGRAM, cfgs["spectrogram"])
            sg.update_config(cfgs["spectrogram"])
        if call == 9:
            cfgs["stereometer"] = StereometerConfig(analyze_bands=True, correlation_window=0.1)
            group.update_config(capi.VISUAL_STEREOMETER, cfgs["stereometer"])
            st.update_config(cfgs["stereometer"])
        mask = np.zeros(S, np.uint8)
        if call == 10:
            mask[3] = 1
        if call == 12:
            generation += 1
        was_reset = group.note_format(generation)
        assert was_reset == (call == 12)
        if was_reset:
            for b in (sg, sp, ld, st, sc, wf):
                b.reset_audio()
            # lock-step positions again: one lock-step call goes through (and equals the banks'), the next ragged call switches back
            chunk = torch.stack([feed[s, at[s]:at[s] + cap] for s in range(S)]).contiguous()
            u = group.ingest(chunk.data_ptr(), cap, 2, 48000.0, pos)
            r_ld = ld.process_device(chunk.data_ptr(), block, maxb, 2, 48000.0, pos)
            sg.process_device(chunk.data_ptr(), cap, 2, 48000.0, pos)
            sp.process_device(chunk.data_ptr(), cap, 2, 48000.0, pos)
            st.process_device(chunk.data_ptr(), block, maxb, 2, 48000.0, pos)
            sc.process_device(chunk.data_ptr(), block, maxb, 2, 48000.0, pos)
            wf.process_device(chunk.data_ptr(), cap, 2, 48000.0, pos)
            torch.cuda.synchronize()
            assert torch.equal(dview(torch, u.d_loudness, (S, maxb, 30)), dview(torch, r_ld, (S, maxb, 30)))
            at = [a + cap for a in at]
        frames = (rng.integers(0, maxb + 1, S) * block).astype(np.uint32)
        frames[rng.integers(0, S)] = 0
        if call == 0:
            frames[:] = cap
        chunk = torch.zeros((S, cap, 2), device=dev, dtype=torch.float32)
        for s in range(S):
            chunk[s, :frames[s]] = feed[s, at[s]:at[s] + int(frames[s])]
            at[s] += int(frames[s])
        nb = (frames // block).astype(np.uint32)
        u = group.ingest_ragged(chunk.data_ptr(), cap, frames, 2, 48000.0, pos, reset_mask=mask)
        assert u.ingest_launches == 1   # one shared projection for the Spectrogram and Spectrum banks (the twin banks below launch one each)
        with pytest.raises(capi.OmxError):
            group.ingest(chunk.data_ptr(), cap, 2, 48000.0, pos)
        r_sg = sg.process_ragged(chunk.data_ptr(), cap, frames, 2, 48000.0, pos, mask)
        r_sp = sp.process_ragged(chunk.data_ptr(), cap, frames, 2, 48000.0, pos, mask)
        r_ld = ld.process_ragged(chunk.data_ptr(), block, maxb, nb, 2, 48000.0, pos, mask)
        # a disabled visual is not fed; a per-capture reset still reaches it with its next call, so the twin bank sits the call out
        # too and gets the mask it missed OR-ed into its next one
        r_st = st.process_ragged(chunk.data_ptr(), block, maxb, nb, 2, 48000.0, pos, mask) if on["stereometer"] else None
        r_sc = sc.process_ragged(chunk.data_ptr(), block, maxb, nb, 2, 48000.0, pos, mask) if on["oscilloscope"] else None
        r_wf = wf.process_ragged(chunk.data_ptr(), cap, frames, 2, 48000.0, pos, mask) if on["waveform"] else None
        torch.cuda.synchronize()
        assert int(u.block_frames) == block and int(u.max_blocks) == maxb
        mc = int(r_sg.max_columns)
        assert int(u.spectrogram.max_columns) == mc
        assert torch.equal(dview(torch, u.spectrogram.d_n_columns, (S,)), dview(torch, r_sg.d_n_columns, (S,)))
        assert torch.equal(dview(torch, u.spectrogram.d_reset, (S,)), dview(torch, r_sg.d_reset, (S,)))
        if mc:
            stride = int(r_sg.column_stride)
            assert torch.equal(dview(torch, u.spectrogram.d_counts, (S, mc)), dview(torch, r_sg.d_counts, (S, mc)))
            ca = dview(torch, r_sg.d_counts, (S, mc)).cpu().numpy()
            pa, pb = dview(torch, u.spectrogram.d_points, (S, mc, stride, 3)), dview(torch, r_sg.d_points, (S, mc, stride, 3))
            for s in range(S):
                for c in range(mc):
                    assert torch.equal(pa[s, c, :ca[s, c]], pb[s, c, :ca[s, c]]), (call, s, c)
        assert torch.equal(dview(torch, u.loudness.d_n_blocks, (S,)), dview(torch, r_ld.d_n_blocks, (S,)))
        la, lb = dview(torch, u.loudness.d_snapshots, (S, maxb, 30)), dview(torch, r_ld.d_snapshots, (S, maxb, 30))
        for s in range(S):
            assert torch.equal(la[s, :nb[s]], lb[s, :nb[s]]), (call, s)
        assert bool(u.produced & capi.VISUAL_STEREOMETER) <= on["stereometer"]
        if on["stereometer"]:
            ra, rb = dview(torch, u.stereometer.d_correlations, (S, maxb, 4)), dview(torch, r_st.d_correlations, (S, maxb, 4))
            for s in range(S):
                assert torch.equal(ra[s, :nb[s]], rb[s, :nb[s]]), (call, s)
        if on["oscilloscope"]:
            ha, hb = dview(torch, u.oscilloscope.d_headers, (S, maxb, 10)), dview(torch, r_sc.d_headers, (S, maxb, 10))
            for s in range(S):
                assert torch.equal(ha[s, :nb[s]], hb[s, :nb[s]]), (call, s)
        if on["waveform"]:
            assert torch.equal(dview(torch, u.waveform.d_n_columns, (S,)), dview(torch, r_wf.d_n_columns, (S,)))


def test_rccl_all_gather_of_the_summary_rows_runs_at_world_size_one():
    """K8 on the hardware this box has: `nccl` (= RCCL) initialised at world size 1, the capture group's summary rows pushed through
    all_gather_into_tensor on a side stream (sharding.gather_stats(always_collective=True)) while the next step's kernels are enqueued
    on the main one — the call pattern of bench.py at N > 1.  In a child process: the process group must not leak into pytest's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
import openmeters_amd
from openmeters_amd.pipeline import FullPipeline
from openmeters_amd.sharding import gather_stats
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
S, frames = 32, 4096
pipe = FullPipeline(openmeters_amd.api(), S)
g = torch.Generator(device=dev).manual_seed(3)
pcm = ((torch.rand((S, frames * 4, 2), device=dev, generator=g) - 0.5) * 0.8).contiguous()
side = torch.cuda.Stream(device=dev)
gathered = []
for k in range(4):
    chunk = pcm[:, k * frames:(k + 1) * frames].contiguous()
    up, rows = pipe.step_with_stats(torch, dev, chunk.data_ptr(), frames)
    snapshot = rows.clone()
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        side.wait_event(ready)
        out = gather_stats(snapshot, S, always_collective=True)
        snapshot.record_stream(side)
    gathered.append((snapshot, out))
torch.cuda.synchronize()
for snapshot, out in gathered:
    assert out.shape == (S, 12) and torch.equal(out.view(torch.int32), snapshot.view(torch.int32))
assert float(gathered[-1][1][:, 7].max()) == 16.0
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
""" % root
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
