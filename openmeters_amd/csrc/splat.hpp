// Reassigned-splat accumulation + dB resolve: launcher of splat_kernels.hip.
#pragma once
#include "common.hpp"

namespace omx {
struct SplatArgs {
    const omx_spectrogram_point* points;  // [n_streams][n_columns][column_stride]
    const uint32_t* counts;               // [n_streams][n_columns]
    uint32_t n_streams, n_columns, column_stride;
    uint32_t width, height, freq_scale;
    float extent_x, extent_y, scale_factor;
    float axis_lo, axis_inv;              // freq_axis = (scale(freq_min), 1 / (scale(freq_max) - scale(freq_min)))  (render.rs:216-224)
    float uv_lo, inv_uv;
    float tilt_db;
    float* accum;                         // [n_streams][width][height] (frequency fastest: a time column is contiguous)
};
struct SplatTiling {
    uint32_t tile_cols, margin_cols, window_width, band_rows;
};
// force_form: 0 = choose, 1 = global atomics, 2 = LDS-tiled (OMX_SPLAT_FORM, tuning / tests)
void launch_splat(const SplatArgs& a, float* db, float power_scale, hipStream_t stream, int force_form = 0);
}  // namespace omx
