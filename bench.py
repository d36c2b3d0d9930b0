#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native OpenMeters DSP hot path.

Metric (BASELINE.json): STFT frames/s (4096-pt Hann, hop 256, time-frequency reassignment on) + achieved HBM GB/s vs the
8 TB/s roofline.

Workloads (`--config`):
  cfg2  BASELINE.json configs[1] — the N = 1 default: 64 streams x 2 ch, 48 kHz f32, Spectrogram{4096, hop 256, Hann,
        reassigned} + A-weighted Spectrum{4096, hop 256, Hann}; one step = 262 144 new frames per stream.
  cfg5  BASELINE.json configs[4] — the N > 1 default (and `--config cfg5` at N = 1): every GPU owns 1024 contiguous 2-ch
        streams of the 8192-stream set and runs the FULL pipeline on them (reassigned STFT 4096/256 + BS.1770 loudness +
        band-split phase correlation, the two recurrence banks on side HIP streams beside the FFT kernel); one step = 16 384
        new frames per stream (64 STFT columns, 64 blocks of 256); the per-stream summary rows are assembled and all-gathered
        over RCCL once per step on a side stream.  No data-path collective: streams are independent.

A "step" is one pass of the hot path over one batch of synthetic PCM per stream, already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg5]

N > 1: the driver launches `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in
the environment).  Started WITHOUT those (`python bench.py --gpus 8`), this process spawns exactly that launcher as a child —
before torch is imported or the GPU is touched, never by re-exec — and relays rank 0's JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (spec)
FP32_VECTOR_PEAK_TFLOPS = 157.3  # same guide: peak FP32 vector
BYTES_PER_FRAME_DENSE = 256 * 2 * 4 + 4 + 2049 * 12  # 26,640 B/frame (dense column)
FLOPS_PER_FRAME = 1.80e6         # SURVEY §8(d), the REFERENCE algorithm: 2*5*8192*13 + 3*5*4096*12
# what the kernel executes (5 N log2 N per complex transform): one packed-real forward + one inverse for the Hilbert pair
# (single-IFFT shortcut) + the windowed 4096-point transforms of the kernel form in use
FLOPS_PER_FFT4096 = 5.0 * 4096 * 12
TOTAL_STREAMS_CFG5 = 8192


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=("auto", "cfg2", "cfg5"), default="auto",
                    help="auto = cfg2 (BASELINE configs[1]) at N = 1, cfg5 (configs[4], 1024 streams per GPU, full pipeline) at N > 1")
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default 64 for cfg2, 1024 for cfg5)")
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="new PCM frames per stream per step (default 262144 for cfg2, 16384 for cfg5)")
    ap.add_argument("--cpu-columns", type=int, default=8192, help="columns per stream in the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spectrum", action="store_true", help="cfg2: leave the A-weighted spectrum bank out of the step")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg3 / cfg4 / cfg5 side measurements (N = 1 only)")
    ap.add_argument("--no-n1", action="store_true",
                    help="N > 1: do not measure the same per-GPU workload at N = 1 first (rank 0 runs it as a child process before the "
                         "ranks meet; the line then carries n1_same_workload.measured)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / gather check without device work (CPU tests: OMX_BENCH_BACKEND=gloo)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with no RANK in the environment: start N rank processes through torch.distributed.run as a
    CHILD of this (GPU-untouched, torch-unimported) process and pass rank 0's JSON line through."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank launch failed (exit code {proc.returncode}, result line {'present' if line else 'missing'})",
              file=sys.stderr)
        return proc.returncode or 1
    if json.loads(line).get("n_gpus") != args.gpus:
        print(f"bench.py: the ranks report n_gpus != {args.gpus}", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


class DeviceView:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


# ------------------------------------------------------------------------------------------------------------ CPU leg
def cpu_baseline(columns, log):
    """The CPU oracle (kind "port": our C++ restatement — the reference's Rust cannot be built here) timed on this node's host
    cores on a bounded sample of the same workload: SURVEY §8(d) — `-O3 -march=native`, 1 thread and
    T = hardware_concurrency threads (streams statically partitioned), cfg2 shape and cfg1 (1024-pt classic) shape."""
    import tempfile
    from openmeters_amd import capi
    import workloads
    # -march=native objects do not travel between hosts: build in a temp dir of THIS host, every time
    build_dir = tempfile.mkdtemp(prefix="omx_oracle_native_")
    lib_path = os.path.join(build_dir, "libomx_oracle_native.so")
    flags = "-O3 -march=native"
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "native", f"NATIVE_OUT={lib_path}"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(lib_path):
        log("cpu_baseline: native build failed, falling back to the -O2 oracle:\n" + r.stderr[-500:])
        lib_path, flags = os.path.join(ROOT, "oracle", "libomx_oracle.so"), "-O2"
        if not os.path.exists(lib_path):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    oracle = capi.Api(lib_path, "omxo_")
    f = oracle.lib.omxo_bench_spectrogram
    f.restype = C.c_double
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0)) or cores
    except (AttributeError, OSError):
        pass
    # a container's CPU quota (cgroup v2 cpu.max / v1 cfs_quota): threads beyond it only time-slice
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
            if q != "max":
                quota = max(1, int(float(q) / float(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = max(1, q // period)
        except (OSError, ValueError):
            pass
    visible = cores
    if quota:
        cores = min(cores, quota)
    threads = cores
    log(f"cpu_baseline: {visible} host cores visible, cgroup quota {quota}, {threads} threads used")

    def run(cfg, pcm, T):
        S, frames = pcm.shape[0], pcm.shape[1]
        cols = C.c_uint64()
        secs = f(C.byref(cfg), pcm.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint64(S), C.c_uint64(frames), C.c_uint32(2),
                 C.c_uint64(256), C.c_uint32(T), C.byref(cols))
        return cols.value, secs

    out = {}
    # cfg2 shape: reassigned 4096 / hop 256, blocks of 256 frames (the DspBatcher quantum)
    cfg = capi.SpectrogramConfig(fft_size=4096, hop_size=256, history_length=8192, use_reassignment=True).to_c()
    frames = 8192 + 256 * (columns - 1)
    one = workloads.cfg2_bank(0, 1, frames)
    n, secs = run(cfg, one, 1)
    out["single_thread"] = n / secs
    log(f"cpu_baseline cfg2 shape, 1 thread: {n} frames in {secs:.2f} s")
    reps = -(-threads // 8)
    all_cols = max(256, columns // 2) if threads > 16 else columns    # ~0.5 s of work per thread at 8 k frames/s (the timed region starts once every thread is warm)
    frames = 8192 + 256 * (all_cols - 1)
    many = np.tile(workloads.cfg2_bank(0, min(threads, 8), frames), (reps, 1, 1))[:threads]  # T streams (8 distinct ones, tiled)
    n, secs = run(cfg, np.ascontiguousarray(many), threads)
    out["all_threads"] = n / secs
    log(f"cpu_baseline cfg2 shape, {threads} threads: {n} frames in {secs:.2f} s")
    # cfg1 (BASELINE configs[0], the reference's own CPU case): 1024-pt Hann classic, hop 256
    cfg1 = capi.SpectrogramConfig(fft_size=1024, hop_size=256, history_length=8192, use_reassignment=False).to_c()
    frames1 = 1024 + 256 * (8 * columns - 1)
    p1 = workloads.cfg1_pcm(frames1)[None]
    n1, s1 = run(cfg1, np.ascontiguousarray(p1), 1)
    nT, sT = run(cfg1, np.ascontiguousarray(np.tile(p1, (threads, 1, 1))), threads)
    log(f"cpu_baseline cfg1 (classic 1024/256): 1 thread {n1 / s1:.0f} frames/s, {threads} threads {nT / sT:.0f} frames/s")
    return {"value": out["all_threads"], "unit": "frames/s", "cores": threads, "kind": "port",
            "single_thread": out["single_thread"],
            "cfg1_classic_1024": {"single_thread": n1 / s1, "all_threads": nT / sT, "unit": "frames/s",
                                  "sample": f"{8 * columns} columns per stream, 1 and {threads} streams"},
            "host_cores_visible": visible, "cgroup_cpu_quota": quota,
            "sample": f"{threads} streams x {all_cols} reassigned 4096/256 columns each (blocks of 256 frames), C++ oracle "
                      f"{flags} -ffp-contract=off, {threads} threads = min(host cores visible to the process, cgroup CPU quota), timed from the "
                      f"moment every thread has built its tables and touched its PCM; "
                      f"single_thread = 1 stream x {columns} columns"}


# ------------------------------------------------------------------------------------------------------------ rank body
def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}) or drop RANK/WORLD_SIZE")
    log = (lambda m: print(m, file=sys.stderr, flush=True)) if rank == 0 else (lambda m: None)
    config = args.config if args.config != "auto" else ("cfg2" if world == 1 else "cfg5")
    S = args.streams or (64 if config == "cfg2" else 1024)
    F = args.frames_per_step or (256 * 1024 if config == "cfg2" else 16384)
    hop, W = 256, 4096
    cols_per_step = F // hop
    if cols_per_step > 8192 or F % 256:
        raise SystemExit("frames-per-step must be a multiple of 256 and frames-per-step / hop <= 8192 (history retention clamp)")

    import torch
    import torch.distributed as dist
    from openmeters_amd.sharding import STATS_COLUMNS, gather_stats, shard_streams

    backend = os.environ.get("OMX_BENCH_BACKEND", "nccl")  # "gloo": test hook to exercise N > 1 on a 1-GPU (or GPU-less) box
    if args.dry_run:
        # launcher / rendezvous / shard map / gather, no device work
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        total = S * world
        first, count = shard_streams(total, rank, world)
        local = torch.zeros((count, len(STATS_COLUMNS)), dtype=torch.float32)
        local[:, 7] = torch.arange(first, first + count, dtype=torch.float32)
        table = gather_stats(local, total)
        assert table.shape == (total, len(STATS_COLUMNS)) and torch.equal(table[:, 7], torch.arange(total, dtype=torch.float32))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "STFT frames/s (4096-pt Hann, hop 256, reassignment on)", "value": None, "unit": "frames/s",
                              "n_gpus": world, "steps": 0, "warmup": 0, "dry_run": True,
                              "config": {"workload": config, "streams_per_gpu": S, "gathered_rows": int(table.shape[0])}}), flush=True)
        return

    # N > 1: the same per-GPU workload at N = 1, measured by THIS run (rank 0, as a child process that is gone before this process
    # touches the GPU; the other ranks wait in the rendezvous meanwhile) so that weak-scaling efficiency is read against a number of
    # the same box, clocks and build
    n1_measured = None
    if world > 1 and rank == 0 and not args.no_n1:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                                                                "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config", config, "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--streams", str(S), "--frames-per-step", str(F), "--no-secondary", "--no-cpu-baseline"]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
            if r.returncode == 0 and lines:
                rec = json.loads(lines[-1])
                n1_measured = {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "kernel_ms": rec["roofline"]["kernel_ms"],
                               "steps": rec["steps"], "command": " ".join(cmd[1:])}
            else:
                log(f"n1 leg failed (exit {r.returncode}): {r.stderr[-400:]}")
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
            log(f"n1 leg failed: {e}")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)  # == local_rank on a real multi-GPU node
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (rank 0 joins after its N = 1 leg: the others wait for its store meanwhile — a generous rendezvous timeout)
        from datetime import timedelta
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=timedelta(minutes=30))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timedelta(minutes=30))

    import openmeters_amd
    from openmeters_amd import capi
    from openmeters_amd.pipeline import CaptureGroup, FullPipeline
    import workloads

    api = openmeters_amd.api()
    assert openmeters_amd.device_available()
    stream0 = rank * S
    # cfg2 / cfg5 share one generator: sweep with start phase 2 pi s / 64 + xorshift32 noise at -60 dBFS
    t_gen = time.perf_counter()
    pcm = torch.from_numpy(workloads.cfg2_bank(stream0, S, F)).to(device).contiguous()
    torch.cuda.synchronize()
    log(f"synthetic PCM: streams {stream0}..{stream0 + S - 1}, {F} frames each, generated in {time.perf_counter() - t_gen:.1f} s")
    stream = torch.cuda.current_stream().cuda_stream
    positions = capi.positions_fallback(2)
    side = torch.cuda.Stream(device=device)

    def gather_on_side(make_rows, keep_alive):
        """K8: summary rows assembled and all-gathered on the side stream while the next step's kernels run"""
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            side.wait_event(ready)
            rows = make_rows()
            gather_stats(rows if backend == "nccl" else rows.cpu(), world * S)
            for t in keep_alive:
                t.record_stream(side)

    # Both workloads run through the capture group of the C-ABI (omx_capture_group_*: VisualManager::ingest_samples,
    # registry.rs:396-418): one ingest call per step feeds every enabled visual, with one projection launch for the banks that keep
    # pending audio, the meter banks on the library's side streams and the summary rows assembled by its own kernels.
    if config == "cfg2":
        cfg = capi.SpectrogramConfig(sample_rate=48000.0, fft_size=W, hop_size=hop, window=capi.WINDOW_HANN,
                                     history_length=8192, use_reassignment=True, zero_padding_factor=1)
        # "+ A-weighted spectrum" of configs[1]: Spectrum{4096, hop 256, Hann, averaging None, source Mid}; every hop is
        # materialised (what the reference computes when fed one hop per block)
        spectrum_cfg = None if args.no_spectrum else capi.SpectrumConfig(
            sample_rate=48000.0, fft_size=W, hop_size=hop, window=capi.WINDOW_HANN, averaging_mode=capi.AVG_NONE, source=capi.CH_MID,
            secondary_source=capi.CH_NONE, floor_db=-100.0)
        group = CaptureGroup(api, S, spectrogram=cfg, spectrum=spectrum_cfg, spectrum_emit_all_hops=True)

        def step():
            g = group.ingest(pcm.data_ptr(), F, 2, 48000.0, positions, stream)
            up = g.spectrogram if g.produced & capi.VISUAL_SPECTROGRAM else None
            if world > 1 and up is not None:
                counts = torch.as_tensor(DeviceView(up.d_counts, (S, up.n_columns), "<i4"), device=device).clone()
                n_columns = float(up.n_columns)

                def rows():
                    cf = counts.to(torch.float32)
                    stats = torch.zeros((S, len(STATS_COLUMNS)), device=device, dtype=torch.float32)
                    stats[:, 7] = n_columns
                    stats[:, 8] = cf.mean(dim=1)
                    stats[:, 9] = cf[:, -1]
                    return stats
                gather_on_side(rows, [counts])
            return up
    else:
        pipe = FullPipeline(api, S)
        group = pipe.group

        def step():
            # the three banks side by side inside the library; the summary rows come back as a view of its device buffer and are
            # all-gathered on a further side stream
            g, rows = pipe.step_with_stats(torch, device, pcm.data_ptr(), F)
            up = g.spectrogram if g.produced & capi.VISUAL_SPECTROGRAM else None
            if up is not None and world > 1:
                snapshot = rows.clone()   # (the library reuses its row buffer in the next ingest)
                gather_on_side(lambda: snapshot, [snapshot])
            return up

    # BASELINE.json's other single-GPU configurations (a few seconds; never part of `value`): cfg3 loudness, cfg4 oscilloscope +
    # stereometer, the other one of cfg2 / cfg5, the waveform bank, the streaming cadence.  Measured AHEAD of the timed region since round 6:
    # behind it, a `--steps 20` run (36 ms) was timed on a chip still ramping its clocks — 36.4 M frames/s where the same box gives 37.8 M
    # over 200 steps ("make 20 steps robust"); `step_ms_spread` still shows every step of the timed region.
    secondary_result = None
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            import bench_meters
            import bench_pipeline
            sec = {"cfg3_loudness": bench_meters.loudness(out=sys.stderr)}
            sec.update({"cfg4_" + k: v for k, v in bench_meters.scope_stereo(out=sys.stderr).items()})
            sec.update(bench_meters.reference_defaults(out=sys.stderr))   # the reference's default shapes (2048 / 64, 16384 / 1024)
            if config == "cfg2":
                sec["cfg5_shard"] = bench_pipeline.shard_pipeline(out=sys.stderr)
            sec["waveform_1024"] = bench_meters.waveform(sizes=(1024,), out=sys.stderr)   # §8f rank 3, with its roofline objects
            import bench_stream
            sec["streaming_256"] = bench_stream.streaming(out=sys.stderr)   # the reference's own cadence: one batcher block per call
            import bench_scope_rates
            sec["oscilloscope_rates"] = bench_scope_rates.rates(which=(96000.0, 192000.0), out=sys.stderr)   # cfg4's bank at the high rates
            secondary_result = sec
        except Exception as e:  # the headline line must survive a failure here
            secondary_result = {"error": repr(e)}
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    group.set_option(capi.OPT_KERNEL_TIMING, 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    # one event per step boundary on the step's stream (no synchronisation: the spread of the steps inside the timed region, read back
    # after it — a 20-step run is a 40 ms sample of a clock that moves +-15 % box to box)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    marks[0].record()
    for i in range(args.steps):
        last = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    step_spread = {"min": step_ms[0], "median": step_ms[len(step_ms) // 2], "p95": step_ms[min(len(step_ms) - 1, int(0.95 * len(step_ms)))],
                   "max": step_ms[-1], "unit": "ms", "note": "device time between consecutive step boundaries on the step's stream (HIP events)"}
    kernel_ms, launches = group.kernel_time()
    assert last is not None and last.n_columns == cols_per_step, (last.n_columns if last else None, cols_per_step)
    counts = torch.as_tensor(DeviceView(last.d_counts, (S, last.n_columns), "<i4"), device=device)
    mean_points = float(counts.to(torch.float64).mean().item())

    frames_total = world * S * cols_per_step * args.steps
    value = frames_total / elapsed
    frames_per_launch = S * cols_per_step
    bytes_per_frame = hop * 2 * 4 + 4 + mean_points * 12.0  # PCM once + count + points actually written
    per_s = frames_per_launch / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
    achieved_gbs = per_s * bytes_per_frame / 1e9
    transforms = api.fn("debug_transforms_per_frame", C.c_int, [])()
    executed_flops = transforms * FLOPS_PER_FFT4096
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot be sampled from inside
    # this process): tools/profile_bench.sh writes profiles/<tag>_traffic.json with the git commit and kernel_ms of THAT run.
    # It is echoed here only as a tagged record of that profile, never as a measurement of this run.
    traffic, traffic_source = None, None
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
            with open(path) as fh:
                rec = json.load(fh)
            w = rec.get("workload", {})
            if w.get("config", "cfg2") == config and w.get("streams_per_gpu") == S and w.get("columns_per_step_per_gpu") == frames_per_launch:
                traffic = rec["hbm_bytes_per_launch"]
                traffic_source = {"file": "profiles/" + os.path.basename(path), "commit": rec.get("commit"),
                                  "kernel_ms_of_that_run": rec.get("kernel_ms"), "transforms_per_frame": rec.get("transforms_per_frame")}
                break
    except Exception:
        pass
    workload = ("BASELINE.json configs[1]: 64-stream x 2-ch 48 kHz, 4096-pt Hann STFT hop 256, time-frequency reassignment + A-weighted spectrum"
                if config == "cfg2" else
                "BASELINE.json configs[4]: 8192 independent 2-ch streams sharded 1024/GPU, full pipeline (reassigned STFT 4096/256 + "
                "BS.1770 LUFS + band-split correlation), RCCL gather of per-stream stats once per step")
    result = {
        "metric": "STFT frames/s (4096-pt Hann, hop 256, reassignment on)",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        # ranks the collective backend itself reports after its init (N > 1: torch.distributed over RCCL; N = 1: no communicator is made)
        "rccl_ranks_seen": (dist.get_world_size() if world > 1 else 1),
        "collective_backend": (backend if world > 1 else None),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "step_ms_spread": step_spread,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload, "name": config,
                   "streams_per_gpu": S, "frames_per_stream_per_step": F, "columns_per_step_per_gpu": frames_per_launch,
                   "pipeline": ("Spectrogram{4096, 256, Hann, reassigned}" +
                                ("" if args.no_spectrum else " + A-weighted Spectrum{4096, 256, Hann, avg None, Mid}, every hop materialised") +
                                " through omx_capture_group_ingest (one projection launch for both banks)")
                   if config == "cfg2" else
                   "omx_capture_group_ingest: Spectrogram{4096, 256, Hann, reassigned} || Loudness{BS.1770 M/S LUFS, 4x true peak} || Stereometer{bands, 50 ms} + stats rows + gather",
                   "noise": "xorshift32(0x9E3779B9 ^ s) at -60 dBFS (SURVEY §8d)",
                   "parallelism": f"streams sharded x{world}, no data-path collective; all_gather of {len(STATS_COLUMNS)} f32 per stream per step"},
        "roofline": {
            "bound": "hbm",
            "achieved": achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": "stft_reassigned_4096_tri_kernel",
            "kernel_ms": kernel_ms,
            "launches_timed": launches,
            "bytes_per_frame": bytes_per_frame,
            "bytes_per_frame_dense": BYTES_PER_FRAME_DENSE,
            "mean_points_per_frame": mean_points,
            "note": "this kernel is VALU/LDS-bound by construction (68 flop/B, SURVEY §8d): see the fp32 fractions",
            "transforms_per_frame": transforms,
            "fp32_vector_frac_executed": per_s * executed_flops / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
            "fp32_vector_frac_reference_algorithm": per_s * FLOPS_PER_FRAME / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
        },
    }
    if world > 1:
        # Weak scaling of the N > 1 workload is measured against the SAME workload at N = 1 (`python bench.py --config cfg5`), not against
        # the default N = 1 line (cfg2, the configuration the metric is quoted on): the newest recorded N = 1 line of it is echoed, tagged
        result["per_gpu_value"] = value / world
        ref = None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_bench_line_{config}.json")), reverse=True):
            try:
                with open(path) as fh:
                    rec = json.load(fh)
                if rec.get("n_gpus") == 1 and rec.get("config", {}).get("name") == config:
                    ref = {"file": "profiles/" + os.path.basename(path), "value": rec["value"], "ms_per_step": rec["ms_per_step"]}
                    break
            except (OSError, ValueError, KeyError):
                continue
        # The comparable N = 1 line of THIS line: the same per-GPU workload (`--gpus 1 --config <this config>`), measured by this very
        # run.  `bench.py --gpus 1` with no --config is BASELINE configs[1] (cfg2, 64 streams per GPU); `--gpus N` with no --config is
        # configs[4] (cfg5, 1024 streams per GPU, full pipeline): their `value`s are different workloads and must not be divided.
        n1 = {"command": f"python bench.py --gpus 1 --config {config}", "measured": n1_measured, "recorded": ref,
              "note": "weak-scaling reference of this line: same config, same per-GPU streams and frames; the default N = 1 line (cfg2) is a different workload"}
        if n1_measured:
            n1["weak_scaling_vs_measured_n1"] = (value / world) / n1_measured["value"]
        # first in the line, right behind n_gpus
        ordered = {}
        for k, v in result.items():
            ordered[k] = v
            if k == "n_gpus":
                ordered["n1_same_workload"] = n1
        result = ordered
        if n1_measured:
            result["weak_scaling_vs_measured_n1"] = n1["weak_scaling_vs_measured_n1"]
    if rank == 0:
        if secondary_result is not None:
            result["secondary"] = secondary_result
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.cpu_columns, log)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
