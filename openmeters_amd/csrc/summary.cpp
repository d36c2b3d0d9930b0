// C entry points of the state-side summary reductions (include/omx.h).
// on_device = 0 stages the host arrays through temporary device buffers: the arithmetic always runs in
// summary_kernels.hip, there is no host implementation.
#include "summary.hpp"

using namespace omx;

#define REQUIRE_DEVICE()                       \
    do {                                       \
        const int _rc = ::omx::device_ready(); \
        if (_rc < 0) return _rc;               \
    } while (0)

extern "C" {

int omx_spectrum_peaks(const float* bins, const float* db, int on_device, uint64_t n_bins, uint64_t n_rows, uint64_t row_stride,
                       float min_f, float max_f, void* stream, omx_spectrum_peak* out) {
    if (!bins || !db || !out || row_stride < n_bins) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (on_device) {
            launch_spectrum_peaks(bins, db, n_bins, n_rows, row_stride, min_f, max_f, out, st);
            OMX_HIP(hipGetLastError());
            return (int)OMX_PRODUCED;
        }
        if (n_rows == 0) return (int)OMX_PRODUCED;
        DeviceBuffer<float> d_bins, d_db;
        DeviceBuffer<omx_spectrum_peak> d_out;
        const size_t span = (size_t)((n_rows - 1) * row_stride + n_bins);
        d_bins.reserve(n_bins);
        d_db.reserve(span);
        d_out.reserve(n_rows);
        OMX_HIP(hipMemcpyAsync(d_bins.ptr, bins, n_bins * sizeof(float), hipMemcpyHostToDevice, st));
        OMX_HIP(hipMemcpyAsync(d_db.ptr, db, span * sizeof(float), hipMemcpyHostToDevice, st));
        launch_spectrum_peaks(d_bins.ptr, d_db.ptr, n_bins, n_rows, row_stride, min_f, max_f, d_out.ptr, st);
        OMX_HIP(hipGetLastError());
        OMX_HIP(hipMemcpyAsync(out, d_out.ptr, n_rows * sizeof(omx_spectrum_peak), hipMemcpyDeviceToHost, st));
        OMX_HIP(hipStreamSynchronize(st));
        return (int)OMX_PRODUCED;
    });
}

int omx_peak_holds_reset(omx_peak_hold* holds, int on_device, uint64_t n, double now, void* stream) {
    if (!holds) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (on_device) {
            launch_peak_holds_reset(holds, n, now, st);
            OMX_HIP(hipGetLastError());
            return (int)OMX_NONE;
        }
        DeviceBuffer<omx_peak_hold> d;
        d.reserve(n);
        launch_peak_holds_reset(d.ptr, n, now, st);
        OMX_HIP(hipGetLastError());
        OMX_HIP(hipMemcpyAsync(holds, d.ptr, n * sizeof(omx_peak_hold), hipMemcpyDeviceToHost, st));
        OMX_HIP(hipStreamSynchronize(st));
        return (int)OMX_NONE;
    });
}

int omx_loudness_meters(const omx_loudness_snapshot* snapshots, int on_device, uint64_t n_streams, uint64_t n_blocks,
                        uint32_t left_mode, uint32_t right_mode, double t0, double dt, omx_peak_hold* holds, void* stream,
                        omx_meter_row* rows) {
    if (!snapshots || !holds || !rows || left_mode > OMX_METER_TRUE_PEAK || right_mode > OMX_METER_TRUE_PEAK) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (on_device) {
            launch_loudness_meters(snapshots, n_streams, n_blocks, left_mode, right_mode, t0, dt, holds, rows, st);
            OMX_HIP(hipGetLastError());
            return (int)OMX_PRODUCED;
        }
        const size_t n = (size_t)(n_streams * n_blocks);
        if (n == 0) return (int)OMX_PRODUCED;
        DeviceBuffer<omx_loudness_snapshot> d_snap;
        DeviceBuffer<omx_peak_hold> d_holds;
        DeviceBuffer<omx_meter_row> d_rows;
        d_snap.reserve(n);
        d_holds.reserve(3 * n_streams);
        d_rows.reserve(n);
        OMX_HIP(hipMemcpyAsync(d_snap.ptr, snapshots, n * sizeof(omx_loudness_snapshot), hipMemcpyHostToDevice, st));
        OMX_HIP(hipMemcpyAsync(d_holds.ptr, holds, 3 * n_streams * sizeof(omx_peak_hold), hipMemcpyHostToDevice, st));
        launch_loudness_meters(d_snap.ptr, n_streams, n_blocks, left_mode, right_mode, t0, dt, d_holds.ptr, d_rows.ptr, st);
        OMX_HIP(hipGetLastError());
        OMX_HIP(hipMemcpyAsync(holds, d_holds.ptr, 3 * n_streams * sizeof(omx_peak_hold), hipMemcpyDeviceToHost, st));
        OMX_HIP(hipMemcpyAsync(rows, d_rows.ptr, n * sizeof(omx_meter_row), hipMemcpyDeviceToHost, st));
        OMX_HIP(hipStreamSynchronize(st));
        return (int)OMX_PRODUCED;
    });
}

}  // extern "C"
