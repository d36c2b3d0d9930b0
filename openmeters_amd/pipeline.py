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

    def stats(self, torch, device, up, snaps_ptr, st, n_blocks):
        """[n_streams, len(STATS_COLUMNS)] float32 summary rows in sharding.STATS_COLUMNS order."""
        import ctypes as C
        S = self.n_streams
        out = torch.zeros((S, len(STATS_COLUMNS)), device=device, dtype=torch.float32)
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
        corr = torch.as_tensor(_DeviceView(st.d_correlations, (S, n_blocks, 4), "<f4"), device=device)[:, -1]
        out[:, 3:7] = corr                             # rho full / low / mid / high
        if up is not None:
            cols = int(up.n_columns)
            counts = torch.as_tensor(_DeviceView(up.d_counts, (S, cols), "<i4"), device=device).to(torch.float32)
            out[:, 7] = float(cols)
            out[:, 8] = counts.mean(dim=1)
            out[:, 9] = counts[:, -1]
        return out


__all__ = ["FullPipeline", "gather_stats", "shard_streams", "STATS_COLUMNS"]
