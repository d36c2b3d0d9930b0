"""cfg5 of BASELINE.json: the full per-stream pipeline (reassigned STFT + BS.1770 loudness + phase correlation) on one
GPU's shard of streams, plus the per-stream summary row that is all-gathered over RCCL once per epoch (K8).

Only glue lives here: the three banks do the work (HIP kernels behind the C-ABI); the summary table is assembled from
their device-resident outputs with a handful of torch ops (it is 40 bytes per stream)."""
from __future__ import annotations

import numpy as np

from . import banks, capi
from .sharding import STATS_COLUMNS, gather_stats, shard_streams


class _DeviceView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


LOUDNESS_SNAPSHOT_FLOATS = 30  # sizeof(omx_loudness_snapshot) / 4


class FullPipeline:
    """One GPU's shard: `n_streams` 2-channel streams, blocks of 256 frames."""

    def __init__(self, api: capi.Api, n_streams: int, channels: int = 2, sample_rate: float = 48000.0):
        self.api, self.n_streams, self.channels, self.sample_rate = api, n_streams, channels, sample_rate
        self.positions = capi.positions_fallback(channels)
        self.spectrogram = banks.SpectrogramBank(api, capi.SpectrogramConfig(sample_rate=sample_rate, fft_size=4096, hop_size=256,
                                                                             history_length=8192, use_reassignment=True), n_streams)
        self.loudness = banks.LoudnessBank(api, capi.LoudnessConfig(sample_rate=sample_rate), n_streams, channels)
        self._holds = None      # device bytes: omx_peak_hold [n_streams][3], carried across steps (K9)
        self._clock = 0.0       # sample clock of the next applied snapshot, seconds
        self.stereometer = banks.StereometerBank(api, capi.StereometerConfig(sample_rate=sample_rate, analyze_bands=True,
                                                                             correlation_window=0.05, segment_duration=0.02,
                                                                             target_sample_count=2000), n_streams)

    def step(self, device_ptr: int, frames: int, stream: int = 0):
        """Feeds `frames` (a multiple of 256) new frames per stream to the three banks.  Returns the raw bank updates."""
        assert frames % 256 == 0
        up = self.spectrogram.process_device(device_ptr, frames, self.channels, self.sample_rate, self.positions, stream)
        snaps = self.loudness.process_device(device_ptr, 256, frames // 256, self.channels, self.sample_rate, self.positions, stream)
        st = self.stereometer.process_device(device_ptr, 256, frames // 256, self.channels, self.sample_rate, self.positions, stream)
        return up, snaps, st, frames // 256

    def step_concurrent(self, torch, device_ptr: int, frames: int):
        """Same as `step` on torch's current stream, but the loudness and stereometer banks (register-pipeline kernels: a few
        waves per CU, latency-bound) run on two side streams beside the FFT-bound spectrogram kernel; joined before returning
        to the caller's stream order."""
        assert frames % 256 == 0
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = [torch.cuda.Stream(), torch.cuda.Stream()]
        fork = torch.cuda.Event()
        fork.record(main)
        up = self.spectrogram.process_device(device_ptr, frames, self.channels, self.sample_rate, self.positions, main.cuda_stream)
        results = []
        for side, bank in zip(self._side, (self.loudness, self.stereometer)):
            side.wait_event(fork)
            results.append(bank.process_device(device_ptr, 256, frames // 256, self.channels, self.sample_rate, self.positions,
                                               side.cuda_stream))
            done = torch.cuda.Event()
            done.record(side)
            main.wait_event(done)
        return up, results[0], results[1], frames // 256

    def step_with_stats(self, torch, device, device_ptr: int, frames: int):
        """`step_concurrent` + `stats` with the summary columns of the loudness and stereometer banks assembled on THEIR side
        streams, beside the spectrogram kernel, instead of after the join: the K9 peak-hold kernel (one lane per stream walking the
        call's blocks on the sample clock: ~140 us of pure latency) and a dozen small torch ops leave the step's critical path; only
        the three columns taken from the spectrogram's point counts are written after it.  Same rows, bit for bit
        (tests/test_gpu_pipeline.py).  Returns (spectrogram update or None, rows [n_streams, len(STATS_COLUMNS)])."""
        assert frames % 256 == 0
        n_blocks = frames // 256
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = [torch.cuda.Stream(), torch.cuda.Stream()]
        out = torch.zeros((self.n_streams, len(STATS_COLUMNS)), device=device, dtype=torch.float32)   # on main, ahead of the fork
        fork = torch.cuda.Event()
        fork.record(main)
        up = self.spectrogram.process_device(device_ptr, frames, self.channels, self.sample_rate, self.positions, main.cuda_stream)
        joins = []
        for side, which in zip(self._side, ("loudness", "stereometer")):
            side.wait_event(fork)
            with torch.cuda.stream(side):
                out.record_stream(side)
                if which == "loudness":
                    snaps = self.loudness.process_device(device_ptr, 256, n_blocks, self.channels, self.sample_rate, self.positions,
                                                         side.cuda_stream)
                    self._loudness_columns(torch, device, out, snaps, n_blocks)
                else:
                    st = self.stereometer.process_device(device_ptr, 256, n_blocks, self.channels, self.sample_rate, self.positions,
                                                         side.cuda_stream)
                    self._stereometer_columns(torch, device, out, st, n_blocks)
                done = torch.cuda.Event()
                done.record(side)
                joins.append(done)
        for done in joins:
            main.wait_event(done)
        self._spectrogram_columns(torch, device, out, up)
        return up, out

    def _loudness_columns(self, torch, device, out, snaps_ptr, n_blocks):
        """columns 0-2 and 10-11 of the summary rows, on torch's current stream"""
        import ctypes as C
        S = self.n_streams
        if snaps_ptr:
            snap = torch.as_tensor(_DeviceView(snaps_ptr, (S, n_blocks, LOUDNESS_SNAPSHOT_FLOATS), "<f4"), device=device)[:, -1]
            out[:, 0] = snap[:, 1]                     # momentary LUFS
            out[:, 1] = snap[:, 0]                     # short-term LUFS
            out[:, 2] = snap[:, 18:18 + self.channels].max(dim=1).values  # max true peak dBTP
            # K9: true-peak bars + their 2 s / 60 dB/s peak holds on the sample clock (loudness/state.rs:36-60, 178-217)
            api, hs = self.api, torch.cuda.current_stream().cuda_stream
            if self._holds is None:
                self._holds = torch.empty(S * 3 * 16, device=device, dtype=torch.uint8)
                api.check(api.fn("peak_holds_reset", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_double, C.c_void_p])(
                    self._holds.data_ptr(), 1, S * 3, self._clock, hs))
            rows = torch.empty((S, n_blocks, 6), device=device, dtype=torch.float32)
            dt = 256.0 / self.sample_rate
            api.check(api.fn("loudness_meters", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double,
                                                          C.c_double, C.c_void_p, C.c_void_p, C.c_void_p])(
                snaps_ptr, 1, S, n_blocks, capi.METER_TRUE_PEAK, capi.METER_LUFS_SHORT_TERM, self._clock, dt, self._holds.data_ptr(), hs,
                rows.data_ptr()))
            self._clock += n_blocks * dt
            out[:, 10:12] = rows[:, -1, 3:5]

    def _stereometer_columns(self, torch, device, out, st, n_blocks):
        """columns 3-6 (rho full / low / mid / high of the last block), on torch's current stream"""
        corr = torch.as_tensor(_DeviceView(st.d_correlations, (self.n_streams, n_blocks, 4), "<f4"), device=device)[:, -1]
        out[:, 3:7] = corr

    def _spectrogram_columns(self, torch, device, out, up):
        """columns 7-9 (columns per step, mean points per column, points of the newest column), on torch's current stream"""
        if up is not None:
            cols = int(up.n_columns)
            counts = torch.as_tensor(_DeviceView(up.d_counts, (self.n_streams, cols), "<i4"), device=device).to(torch.float32)
            out[:, 7] = float(cols)
            out[:, 8] = counts.mean(dim=1)
            out[:, 9] = counts[:, -1]

    def stats(self, torch, device, up, snaps_ptr, st, n_blocks):
        """[n_streams, len(STATS_COLUMNS)] float32 summary rows in sharding.STATS_COLUMNS order (everything on the current stream)."""
        out = torch.zeros((self.n_streams, len(STATS_COLUMNS)), device=device, dtype=torch.float32)
        self._loudness_columns(torch, device, out, snaps_ptr, n_blocks)
        self._stereometer_columns(torch, device, out, st, n_blocks)
        self._spectrogram_columns(torch, device, out, up)
        return out


__all__ = ["FullPipeline", "gather_stats", "shard_streams", "STATS_COLUMNS"]
