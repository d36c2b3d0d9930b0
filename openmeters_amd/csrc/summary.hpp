// State-side summary reductions: launchers of summary_kernels.hip.
#pragma once
#include "common.hpp"

namespace omx {
void launch_spectrum_peaks(const float* bins, const float* db, uint64_t n_bins, uint64_t n_rows, uint64_t row_stride, float min_f,
                           float max_f, omx_spectrum_peak* out, hipStream_t stream);
void launch_peak_holds_reset(omx_peak_hold* holds, uint64_t n, double now, hipStream_t stream);
void launch_loudness_meters(const omx_loudness_snapshot* snapshots, uint64_t n_streams, uint64_t n_blocks, uint32_t left_mode,
                            uint32_t right_mode, double t0, double dt, omx_peak_hold* holds, omx_meter_row* rows, hipStream_t stream);
// summary rows of a capture group: each launch writes its own columns of rows[n_streams][OMX_STATS_COLUMNS]
void launch_stats_loudness(const omx_loudness_snapshot* snapshots, const omx_meter_row* meters, uint64_t n_streams, uint64_t n_blocks,
                           uint32_t channels, float* rows, hipStream_t stream);
void launch_stats_stereometer(const float* correlations, uint64_t n_streams, uint64_t n_blocks, float* rows, hipStream_t stream);
void launch_stats_spectrogram(const uint32_t* counts, uint64_t n_streams, uint64_t n_columns, float* rows, hipStream_t stream);
}  // namespace omx
