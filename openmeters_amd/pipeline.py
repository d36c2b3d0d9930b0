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

    def stats(self, torch, device, up, snaps_ptr, st, n_blocks):
        """[n_streams, 10] float32 summary rows in sharding.STATS_COLUMNS order."""
        S = self.n_streams
        out = torch.zeros((S, len(STATS_COLUMNS)), device=device, dtype=torch.float32)
        if snaps_ptr:
            snap = torch.as_tensor(_DeviceView(snaps_ptr, (S, n_blocks, LOUDNESS_SNAPSHOT_FLOATS), "<f4"), device=device)[:, -1]
            out[:, 0] = snap[:, 1]                     # momentary LUFS
            out[:, 1] = snap[:, 0]                     # short-term LUFS
            out[:, 2] = snap[:, 18:18 + self.channels].max(dim=1).values  # max true peak dBTP
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
