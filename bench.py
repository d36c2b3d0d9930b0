#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native OpenMeters DSP hot path.

Metric (BASELINE.json): STFT frames/s (4096-pt Hann, hop 256, time-frequency reassignment on) +
achieved HBM GB/s vs the 8 TB/s roofline, on config[1]:
    64 streams x 2 ch, 48 kHz f32, Spectrogram{4096, hop 256, Hann, reassigned} (+ A-weighted Spectrum
    {4096, hop 256, Hann} once its bank is built — reported in `config.spectrum`).

A "step" is one pass of the hot path over one batch of synthetic PCM per stream (`--frames-per-step`
new samples per stream, already resident in HBM): K0 ingest (stereo fold + Mid projection into the
per-stream rings) + K2 fused reassigned STFT for every ready (stream, hop).

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL); streams are
independent, so ranks shard them with no data-path collective (weak scaling: 64 streams per GPU) and
only all-gather a small per-stream summary once per step.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (spec)
FP32_VECTOR_PEAK_TFLOPS = 157.3  # same guide: peak FP32 vector
BYTES_PER_FRAME_DENSE = 256 * 2 * 4 + 4 + 2049 * 12  # 26,640 B/frame (dense column)
FLOPS_PER_FRAME = 1.80e6        # 2*5*8192*13 + 3*5*4096*12


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=64, help="streams per GPU")
    ap.add_argument("--frames-per-step", type=int, default=256 * 1024, help="new PCM frames per stream per step")
    ap.add_argument("--cpu-columns", type=int, default=8192, help="columns per stream in the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spectrum", action="store_true", help="leave the A-weighted spectrum bank out of the step")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg3 / cfg4 / cfg5 side measurements (N = 1 only)")
    return ap.parse_args()


def synth_pcm(torch, device, n_streams, frames, stream0):
    """cfg2 generator: per-stream exponential sweep 20 Hz -> 20 kHz over 10 s, start phase
    2*pi*s/64, amplitude 0.5, R = 0.8 L, plus -60 dBFS white noise; generated on the GPU."""
    t = torch.arange(frames, device=device, dtype=torch.float64) / 48000.0
    k = float(np.log(1000.0))
    seconds = 10.0
    base = 2.0 * np.pi * 20.0 * seconds / k * (torch.exp((t % seconds) / seconds * k) - 1.0)
    gen = torch.Generator(device=device)
    gen.manual_seed(0x9E3779B9 + stream0)
    pcm = torch.empty((n_streams, frames, 2), device=device, dtype=torch.float32)
    for s in range(n_streams):
        phase0 = 2.0 * np.pi * ((stream0 + s) % 64) / 64.0
        left = (0.5 * torch.sin(base + phase0)).to(torch.float32)
        left += (torch.rand(frames, device=device, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * 1e-3
        pcm[s, :, 0] = left
        pcm[s, :, 1] = 0.8 * left
    return pcm.contiguous()


class DeviceView:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def cpu_baseline(columns, log):
    """Times the CPU oracle (kind "port": our C++ restatement, the reference's Rust cannot be built
    here) on a bounded sample of the same workload, on this node's host cores."""
    from openmeters_amd import capi
    lib_path = os.path.join(ROOT, "oracle", "libomx_oracle.so")
    if not os.path.exists(lib_path):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    oracle = capi.Api(lib_path, "omxo_")
    f = oracle.lib.omxo_bench_spectrogram
    f.restype = C.c_double
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    frames = 8192 + 256 * (columns - 1)
    cfg = capi.SpectrogramConfig(fft_size=4096, hop_size=256, history_length=8192, use_reassignment=True).to_c()

    def exp_sweep(n, phase0):  # SURVEY §8(d) cfg2 generator (same formula as tests/signals.py)
        t = np.arange(n, dtype=np.float64) / 48000.0
        k = np.log(1000.0)
        return (0.5 * np.sin(2.0 * np.pi * 20.0 * 10.0 / k * (np.exp((t % 10.0) / 10.0 * k) - 1.0) + phase0)).astype(np.float32)

    rng = np.random.default_rng(1234)
    out = {}
    for label, T, S in (("single_thread", 1, 1), ("all_threads", threads, threads)):
        pcm = np.empty((S, frames, 2), np.float32)
        for s in range(S):
            left = exp_sweep(frames, phase0=2 * np.pi * s / 64) + (rng.random(frames, dtype=np.float32) * 2 - 1) * 1e-3
            pcm[s, :, 0] = left
            pcm[s, :, 1] = 0.8 * left
        cols = C.c_uint64()
        secs = f(C.byref(cfg), pcm.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(S), C.c_uint64(frames), C.c_uint32(2),
                 C.c_uint64(256), C.c_uint32(T), C.byref(cols))
        out[label] = cols.value / secs
        log(f"cpu_baseline {label}: {cols.value} frames in {secs:.2f} s on {T} thread(s)")
    return {"value": out["all_threads"], "unit": "frames/s", "cores": threads, "kind": "port",
            "single_thread": out["single_thread"],
            "sample": f"{threads} streams x {columns} reassigned 4096/256 columns each (blocks of 256 frames), "
                      f"C++ oracle -O2, {threads} threads; {cores} host cores visible"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    log = (lambda m: print(m, file=sys.stderr, flush=True)) if rank == 0 else (lambda m: None)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)  # == local_rank on a real multi-GPU node
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = os.environ.get("OMX_BENCH_BACKEND", "nccl")  # "gloo": test hook to exercise N > 1 on a 1-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import openmeters_amd
    from openmeters_amd import capi
    from openmeters_amd.banks import SpectrogramBank, SpectrumBank
    from openmeters_amd.sharding import STATS_COLUMNS, gather_stats

    api = openmeters_amd.api()
    assert openmeters_amd.device_available()

    S, F = args.streams, args.frames_per_step
    hop, W = 256, 4096
    cfg = capi.SpectrogramConfig(sample_rate=48000.0, fft_size=W, hop_size=hop, window=capi.WINDOW_HANN,
                                 history_length=8192, use_reassignment=True, zero_padding_factor=1)
    cols_per_step = F // hop
    if cols_per_step > 8192:
        raise SystemExit("frames-per-step / hop must stay <= 8192 (history retention clamp, SURVEY §7)")
    pcm = synth_pcm(torch, device, S, F, stream0=rank * S)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    bank = SpectrogramBank(api, cfg, S)
    spectrum = None
    if not args.no_spectrum:
        # "+ A-weighted spectrum" of configs[1]: Spectrum{4096, hop 256, Hann, averaging None, source Mid}; every hop
        # is materialised (what the reference computes when fed one hop per block)
        spectrum = SpectrumBank(api, capi.SpectrumConfig(sample_rate=48000.0, fft_size=W, hop_size=hop, window=capi.WINDOW_HANN,
                                                         averaging_mode=capi.AVG_NONE, source=capi.CH_MID,
                                                         secondary_source=capi.CH_NONE, floor_db=-100.0), S, emit_all_hops=True)
    positions = capi.positions_fallback(2)

    # K8 runs beside the data path: the per-stream counts are snapshotted on the compute stream (the bank reuses its output
    # buffers every call), then the summary rows are assembled and all-gathered over RCCL/xGMI on a second HIP stream while
    # the next step's kernels run (48 B/stream: latency-bound, must not sit between two K2 launches)
    side = torch.cuda.Stream(device=device) if world > 1 else None

    def step():
        up = bank.process_device(pcm.data_ptr(), F, 2, 48000.0, positions, stream)
        if spectrum is not None:
            spectrum.process_device(pcm.data_ptr(), F, 2, 48000.0, positions, stream)
        if world > 1 and up is not None:
            counts = torch.as_tensor(DeviceView(up.d_counts, (S, up.n_columns), "<i4"), device=device).clone()
            n_columns = float(up.n_columns)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                side.wait_event(ready)
                cf = counts.to(torch.float32)
                stats = torch.zeros((S, len(STATS_COLUMNS)), device=device, dtype=torch.float32)
                stats[:, 7] = n_columns
                stats[:, 8] = cf.mean(dim=1)
                stats[:, 9] = cf[:, -1]
                gather_stats(stats if backend == "nccl" else stats.cpu(), world * S)
                counts.record_stream(side)
        return up

    for _ in range(args.warmup):
        step()
    bank.set_option(capi.OPT_KERNEL_TIMING, 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    kernel_ms, launches = bank.kernel_time()
    counts = torch.as_tensor(DeviceView(last.d_counts, (S, last.n_columns), "<i4"), device=device)
    mean_points = float(counts.to(torch.float64).mean().item())
    assert last.n_columns == cols_per_step, (last.n_columns, cols_per_step)

    frames_total = world * S * cols_per_step * args.steps
    value = frames_total / elapsed
    frames_per_launch = S * cols_per_step
    bytes_per_frame = hop * 2 * 4 + 4 + mean_points * 12.0  # PCM once + count + points actually written
    achieved_gbs = frames_per_launch * bytes_per_frame / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE cannot be sampled from inside this process): the newest
    # profiles/*_traffic.json written by tools/profile_bench.sh + tools/summarize_pmc.py for this same workload, else null.
    traffic, traffic_source = None, None
    try:
        import glob
        here = os.path.dirname(os.path.abspath(__file__))
        for path in sorted(glob.glob(os.path.join(here, "profiles", "*_traffic.json")), reverse=True):
            with open(path) as fh:
                rec = json.load(fh)
            w = rec.get("workload", {})
            if w.get("streams_per_gpu") == S and w.get("columns_per_step_per_gpu") == frames_per_launch:
                traffic, traffic_source = rec["hbm_bytes_per_launch"], "profiles/" + os.path.basename(path)
                break
    except Exception:
        pass
    result = {
        "metric": "STFT frames/s (4096-pt Hann, hop 256, reassignment on)",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: 64-stream x 2-ch 48 kHz, 4096-pt Hann STFT hop 256, "
                               "time-frequency reassignment",
                   "streams_per_gpu": S, "frames_per_stream_per_step": F, "columns_per_step_per_gpu": frames_per_launch,
                   "spectrum": ("A-weighted Spectrum{4096, hop 256, Hann, avg None, Mid}, every hop materialised, in the timed step"
                                if spectrum is not None else "excluded (--no-spectrum)"),
                   "parallelism": f"streams sharded x{world}"},
        "roofline": {
            "bound": "hbm",
            "achieved": achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": "stft_reassigned_4096_kernel",
            "kernel_ms": kernel_ms,
            "launches_timed": launches,
            "bytes_per_frame": bytes_per_frame,
            "bytes_per_frame_dense": BYTES_PER_FRAME_DENSE,
            "mean_points_per_frame": mean_points,
            "note": "this kernel is VALU/LDS-bound by construction (68 flop/B, SURVEY §8d): see fp32 fraction",
            "fp32_vector_tflops": frames_per_launch * FLOPS_PER_FRAME / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0,
            "fp32_vector_frac": (frames_per_launch * FLOPS_PER_FRAME / (kernel_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS)
            if kernel_ms > 0 else 0.0,
        },
    }
    if rank == 0:
        if world == 1 and not args.no_secondary:
            # BASELINE.json's other single-GPU configurations, measured after the timed region (a few seconds; never part of
            # `value`): cfg3 loudness, cfg4 oscilloscope + stereometer, cfg5's per-GPU shard of the full pipeline
            try:
                del pcm, bank, spectrum
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_meters
                import bench_pipeline
                side = {"cfg3_loudness": bench_meters.loudness(out=sys.stderr)}
                side.update({"cfg4_" + k: v for k, v in bench_meters.scope_stereo(out=sys.stderr).items()})
                side["cfg5_shard"] = bench_pipeline.shard_pipeline(out=sys.stderr)
                result["secondary"] = side
            except Exception as e:  # the headline line must survive a failure here
                result["secondary"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.cpu_columns, log)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
