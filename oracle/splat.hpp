// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Reassigned-splat accumulation + dB resolve, restated from
//   src/visuals/render/shaders/spectrogram.wgsl:66-76    freq_to_norm
//   src/visuals/render/shaders/spectrogram.wgsl:126-147  vs_accum_splat (position, cull)
//   src/visuals/render/shaders/spectrogram.wgsl:215-237  fs_accum (tilt), fs_resolve (scale, dB)
//   src/util/audio/frequency.rs:25-31                    FrequencyScale::scale
//   src/visuals/spectrogram/render.rs:214-224            freq_axis = (scale(min), 1 / (scale(max) - scale(min)))
// PARITY UNPINNED: the reference has no test for these shaders and its numbers come out of a GPU raster pipeline with an
// Rg16Float blend target; this restatement keeps the per-point arithmetic and the pixel-centre coverage rule, sums in f32.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>

#include "../include/omx.h"

namespace omxo {

inline float freq_scale_value(uint32_t scale, float hz) {
    switch (scale) {
        case OMX_FREQ_SCALE_LOGARITHMIC: return std::asinh(hz / 20.0f);
        case OMX_FREQ_SCALE_ERB: return 21.4f * std::log(1.0f + hz / 228.8f) * 0.4342944819f;  // wgsl: log * LOG10_E
        default: return hz;
    }
}

// FrequencyScale::scale (frequency.rs:25-31) — the host side of the uniforms uses log10, the shader ln * LOG10_E
inline float freq_scale_host(uint32_t scale, float hz) {
    switch (scale) {
        case OMX_FREQ_SCALE_LOGARITHMIC: return std::asinh(hz / 20.0f);
        case OMX_FREQ_SCALE_ERB: return 21.4f * std::log10(1.0f + hz / 228.8f);
        default: return hz;
    }
}

inline void splat_view_size(omx_splat_view* v) {
    v->width = (uint32_t)std::ceil(std::fmax(v->extent_x, 1.0f));
    v->height = (uint32_t)std::ceil(std::fmax(v->extent_y, 1.0f));
}

struct SplatConsts {
    float axis_lo, axis_inv, inv_uv;
};
inline SplatConsts splat_consts(const omx_splat_view& v) {
    const float lo = freq_scale_host(v.freq_scale, v.freq_min), hi = freq_scale_host(v.freq_scale, v.freq_max);
    return SplatConsts{lo, 1.0f / std::fmax(hi - lo, 1e-12f), 1.0f / std::fmax(v.uv_hi - v.uv_lo, 1e-12f)};
}

// one point of the column with age `age`; adds its (tilted) power to every covered pixel of `accum` [width][height]
inline void splat_point(const omx_spectrogram_point& p, uint32_t age, const omx_splat_view& v, const SplatConsts& c, float* accum) {
    const float zoomed = ((freq_scale_value(v.freq_scale, p.freq_hz) - c.axis_lo) * c.axis_inv - v.uv_lo) * c.inv_uv;
    if (!(p.power > 0.0f) || zoomed < -0.01f || zoomed > 1.01f) return;
    float power = p.power;
    if (v.tilt_db != 0.0f && !(power > 1.0023052e-14f)) return;  // fs_accum: floor bins are not lifted
    if (v.tilt_db != 0.0f && p.freq_hz > 0.0f) power *= std::exp2(v.tilt_db * std::log2(p.freq_hz / 1000.0f) * 0.3321928095f);
    const float sf = v.scale_factor;
    const float x = v.extent_x - ((float)age - p.time_offset) * sf, y = (1.0f - zoomed) * v.extent_y;
    const float x0 = x - 0.5f * sf, x1 = x + 0.5f * sf, y0 = y - 0.5f * sf, y1 = y + 0.5f * sf;
    // pixel (i, j) is covered when its centre (i + 0.5, j + 0.5) lies in [x0, x1) x [y0, y1)
    const float fi0 = std::ceil(x0 - 0.5f), fi1 = std::ceil(x1 - 0.5f), fj0 = std::ceil(y0 - 0.5f), fj1 = std::ceil(y1 - 0.5f);
    if (!(fi1 > 0.0f && fj1 > 0.0f && fi0 < (float)v.width && fj0 < (float)v.height)) return;
    const int64_t i0 = (int64_t)std::fmax(fi0, 0.0f), i1 = (int64_t)std::fmin(fi1, (float)v.width);
    const int64_t j0 = (int64_t)std::fmax(fj0, 0.0f), j1 = (int64_t)std::fmin(fj1, (float)v.height);
    for (int64_t j = j0; j < j1; ++j)
        for (int64_t i = i0; i < i1; ++i) accum[(size_t)i * v.height + (size_t)j] += power;  // [width][height]: one time column is contiguous
}

inline float splat_resolve(float accumulated, float reassigned_power_scale) {
    const float power = accumulated * reassigned_power_scale;
    if (power <= 0.0f || power != power) return -std::numeric_limits<float>::infinity();
    return std::fmax(std::log(std::fmax(power, 1e-20f)) * 4.342944819f, -140.0f);
}

inline void spectrogram_splat(const omx_spectrogram_point* points, const uint32_t* counts, uint64_t n_streams, uint64_t n_columns,
                              uint64_t column_stride, float power_scale, const omx_splat_view& view, float* accum, float* db) {
    const SplatConsts c = splat_consts(view);
    const size_t px = (size_t)view.width * view.height;
    for (uint64_t s = 0; s < n_streams; ++s) {
        float* acc = accum + s * px;
        for (size_t i = 0; i < px; ++i) acc[i] = 0.0f;
        for (uint64_t col = 0; col < n_columns; ++col) {
            const omx_spectrogram_point* col_points = points + (s * n_columns + col) * column_stride;
            const uint32_t n = counts[s * n_columns + col];
            for (uint32_t i = 0; i < n && i < column_stride; ++i)
                splat_point(col_points[i], (uint32_t)(n_columns - 1 - col), view, c, acc);
        }
        if (db)
            for (size_t i = 0; i < px; ++i) db[s * px + i] = splat_resolve(acc[i], power_scale);
    }
}

}  // namespace omxo
