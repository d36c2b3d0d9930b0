// C entry points of the splat accumulation (include/omx.h).  on_device = 0 stages host arrays through
// temporary device buffers: the arithmetic always runs in splat_kernels.hip.
#include "splat.hpp"

using namespace omx;

namespace {
float freq_scale_host(uint32_t scale, float hz) {  // util/audio/frequency.rs:25-31
    switch (scale) {
        case OMX_FREQ_SCALE_LOGARITHMIC: return std::asinh(hz / 20.0f);
        case OMX_FREQ_SCALE_ERB: return 21.4f * std::log10(1.0f + hz / 228.8f);
        default: return hz;
    }
}
int splat_form() {  // OMX_SPLAT_FORM=1|2 pins the kernel form (A/B and tests); default: chosen per image size
    static const int form = [] {
        const char* e = getenv("OMX_SPLAT_FORM");
        return e ? atoi(e) : 0;
    }();
    return form;
}
}  // namespace

extern "C" {

void omx_splat_view_size(omx_splat_view* view) {
    if (!view) return;
    view->width = (uint32_t)std::ceil(std::fmax(view->extent_x, 1.0f));
    view->height = (uint32_t)std::ceil(std::fmax(view->extent_y, 1.0f));
}

int omx_spectrogram_splat(const omx_spectrogram_point* points, const uint32_t* counts, int on_device, uint64_t n_streams,
                          uint64_t n_columns, uint64_t column_stride, float reassigned_power_scale, const omx_splat_view* view,
                          void* stream, float* accum, float* db) {
    if (!points || !counts || !view || !accum || view->width == 0 || view->height == 0 || !(view->scale_factor >= 1.0f) ||
        n_columns > 65535 || n_streams > 65535 || view->freq_scale > OMX_FREQ_SCALE_ERB)
        return OMX_ERR_INVALID;
    const int rc = device_ready();
    if (rc < 0) return rc;
    return guarded([&] {
        hipStream_t st = static_cast<hipStream_t>(stream);
        SplatArgs a{};
        a.n_streams = (uint32_t)n_streams;
        a.n_columns = (uint32_t)n_columns;
        a.column_stride = (uint32_t)column_stride;
        a.width = view->width;
        a.height = view->height;
        a.freq_scale = view->freq_scale;
        a.extent_x = view->extent_x;
        a.extent_y = view->extent_y;
        a.scale_factor = view->scale_factor;
        const float lo = freq_scale_host(view->freq_scale, view->freq_min), hi = freq_scale_host(view->freq_scale, view->freq_max);
        a.axis_lo = lo;
        a.axis_inv = 1.0f / std::fmax(hi - lo, 1e-12f);
        a.uv_lo = view->uv_lo;
        a.inv_uv = 1.0f / std::fmax(view->uv_hi - view->uv_lo, 1e-12f);
        a.tilt_db = view->tilt_db;
        const size_t px = (size_t)n_streams * view->width * view->height;
        if (on_device) {
            a.points = points;
            a.counts = counts;
            a.accum = accum;
            launch_splat(a, db, reassigned_power_scale, st, splat_form());
            OMX_HIP(hipGetLastError());
            return (int)OMX_PRODUCED;
        }
        DeviceBuffer<omx_spectrogram_point> d_points;
        DeviceBuffer<uint32_t> d_counts;
        DeviceBuffer<float> d_accum, d_db;
        const size_t n_points = (size_t)(n_streams * n_columns * column_stride), n_counts = (size_t)(n_streams * n_columns);
        d_points.reserve(n_points);
        d_counts.reserve(n_counts);
        d_accum.reserve(px);
        if (db) d_db.reserve(px);
        if (n_points) OMX_HIP(hipMemcpyAsync(d_points.ptr, points, n_points * sizeof(omx_spectrogram_point), hipMemcpyHostToDevice, st));
        if (n_counts) OMX_HIP(hipMemcpyAsync(d_counts.ptr, counts, n_counts * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        a.points = d_points.ptr;
        a.counts = d_counts.ptr;
        a.accum = d_accum.ptr;
        launch_splat(a, db ? d_db.ptr : nullptr, reassigned_power_scale, st, splat_form());
        OMX_HIP(hipGetLastError());
        OMX_HIP(hipMemcpyAsync(accum, d_accum.ptr, px * sizeof(float), hipMemcpyDeviceToHost, st));
        if (db) OMX_HIP(hipMemcpyAsync(db, d_db.ptr, px * sizeof(float), hipMemcpyDeviceToHost, st));
        OMX_HIP(hipStreamSynchronize(st));
        return (int)OMX_PRODUCED;
    });
}

}  // extern "C"
