// Reassigned-splat accumulation + dB resolve: launcher of splat_kernels.hip.
#pragma once
#include "common.hpp"

namespace omx {
struct SplatArgs {
    const omx_spectrogram_point* points;  // [n_streams][n_columns][column_stride]
    const uint32_t* counts;               // [n_streams][n_columns]
    uint32_t n_streams, n_columns, column_stride;
    // ring mode (omx_spectrogram_history_splat): column c (0 = oldest visible) lives in slot (slot0 + c) % ring_slots of a
    // [n_streams][ring_slots][column_stride] ring; ring_slots == 0: columns are stored in time order ([n_columns] slots)
    uint32_t ring_slots, slot0;
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

// ---- column history ring (history.cpp): device-side data movement
// columns [n_streams][n_cols][src_stride elems] -> ring slots (slot0 + c) % ring_slots, element size `elem` bytes; for
// reassigned columns only counts[s][c] elements are copied and slot_counts is updated, classic columns are zero-filled to
// ring_stride.  Only the newest min(n_cols, ring_slots) columns are touched (older ones would be overwritten).
struct HistoryScatterArgs {
    const unsigned char* src;
    const uint32_t* src_counts;  // nullptr: classic
    unsigned char* ring;
    uint32_t* slot_counts;       // [n_streams][ring_slots] or nullptr
    uint32_t n_streams, n_cols, first_col, src_stride, ring_stride, ring_slots, slot0, elem;
};
void launch_history_scatter(const HistoryScatterArgs& a, hipStream_t stream);
// remap_retained + resize copy (state.rs:150-174, render.rs:457-504): slot src of the old ring -> (src + old_slots - start) % old_slots
// of the new one when that is < keep
struct HistoryRemapArgs {
    const unsigned char* old_ring;
    unsigned char* new_ring;
    const uint32_t* old_counts;
    uint32_t* new_counts;
    uint32_t n_streams, old_slots, new_slots, start, keep, stride_bytes;
};
void launch_history_remap(const HistoryRemapArgs& a, hipStream_t stream);
// fit_reassigned_slot_capacity (state.rs:131-148) on the device: state[0] = reassigned_points_per_slot (stream 0's counts)
void launch_history_fit(const uint32_t* slot_counts, uint32_t ring_slots, uint32_t* state, hipStream_t stream);
}  // namespace omx
