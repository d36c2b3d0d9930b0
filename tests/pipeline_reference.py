"""Test infrastructure: the cfg5 step as THREE SEPARATE BANKS plus torch glue for the summary rows (what openmeters_amd.pipeline did
before the capture group moved the fan-out and the rows into the library).  The capture group must reproduce these rows and bank
outputs bit for bit; nothing here is used by the product."""
import ctypes as C

from openmeters_amd import banks, capi
from openmeters_amd.sharding import STATS_COLUMNS


class _DeviceView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}


LOUDNESS_SNAPSHOT_FLOATS = 30  # sizeof(omx_loudness_snapshot) / 4


class SeparateBanks:
    def __init__(self, api, n_streams, channels=2, sample_rate=48000.0):
        self.api, self.n_streams, self.channels, self.sample_rate = api, n_streams, channels, sample_rate
        self.positions = capi.positions_fallback(channels)
        self.spectrogram = banks.SpectrogramBank(api, capi.SpectrogramConfig(sample_rate=sample_rate, fft_size=4096, hop_size=256,
                                                                             history_length=8192, use_reassignment=True), n_streams)
        self.loudness = banks.LoudnessBank(api, capi.LoudnessConfig(sample_rate=sample_rate), n_streams, channels)
        self.stereometer = banks.StereometerBank(api, capi.StereometerConfig(sample_rate=sample_rate, analyze_bands=True, correlation_window=0.05,
                                                                             segment_duration=0.02, target_sample_count=2000), n_streams)
        self._holds, self._clock = None, 0.0

    def step(self, device_ptr, frames, stream=0):
        up = self.spectrogram.process_device(device_ptr, frames, self.channels, self.sample_rate, self.positions, stream)
        snaps = self.loudness.process_device(device_ptr, 256, frames // 256, self.channels, self.sample_rate, self.positions, stream)
        st = self.stereometer.process_device(device_ptr, 256, frames // 256, self.channels, self.sample_rate, self.positions, stream)
        return up, snaps, st, frames // 256

    def stats(self, torch, device, up, snaps_ptr, st, n_blocks):
        S = self.n_streams
        out = torch.zeros((S, len(STATS_COLUMNS)), device=device, dtype=torch.float32)
        if snaps_ptr:
            snap = torch.as_tensor(_DeviceView(snaps_ptr, (S, n_blocks, LOUDNESS_SNAPSHOT_FLOATS), "<f4"), device=device)[:, -1]
            out[:, 0] = snap[:, 1]
            out[:, 1] = snap[:, 0]
            out[:, 2] = snap[:, 18:18 + self.channels].max(dim=1).values
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
        out[:, 3:7] = torch.as_tensor(_DeviceView(st.d_correlations, (S, n_blocks, 4), "<f4"), device=device)[:, -1]
        if up is not None:
            cols = int(up.n_columns)
            counts = torch.as_tensor(_DeviceView(up.d_counts, (S, cols), "<i4"), device=device).to(torch.float32)
            out[:, 7] = float(cols)
            out[:, 8] = counts.mean(dim=1)
            out[:, 9] = counts[:, -1]
        return out
