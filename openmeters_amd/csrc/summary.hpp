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
// columns per visual: loudness 0-2, 10, 11; stereometer 3-6; spectrogram 7-9
constexpr uint32_t kStatsLoudnessColumns = 0x0C07u, kStatsStereometerColumns = 0x0078u, kStatsSpectrogramColumns = 0x0380u;
void launch_stats_clear_columns(float* rows, uint64_t n_streams, uint32_t columns, hipStream_t stream);
void launch_stats_spectrogram(const uint32_t* counts, uint64_t n_streams, uint64_t n_columns, float* rows, hipStream_t stream);
// the same for per-capture calls (omx_capture_group_ingest_ragged): capture s ran n_blocks_v[s] <= max_blocks blocks of block_frames_v[s]
// (or block_frames) frames — rows / holds of captures that ran none stay as they are; reset_v[s] restarts the capture's holds on a fresh
// sample clock (LoudnessState::reset_audio, loudness/state.rs:153-160).  clocks[s] = the sample clock of capture s's next snapshot.
void launch_stats_loudness_ragged(const omx_loudness_snapshot* snapshots, uint64_t n_streams, uint64_t max_blocks, const uint32_t* n_blocks_v,
                                  const uint32_t* block_frames_v, uint32_t block_frames, float sample_rate, const uint8_t* reset_v,
                                  uint32_t left_mode, uint32_t right_mode, uint32_t channels, omx_peak_hold* holds, double* clocks, float* rows,
                                  hipStream_t stream);
void launch_stats_stereometer_ragged(const float* correlations, uint64_t n_streams, uint64_t max_blocks, const uint32_t* n_blocks_v, float* rows,
                                     hipStream_t stream);
void launch_stats_spectrogram_ragged(const uint32_t* counts, uint64_t n_streams, uint64_t max_columns, const uint32_t* n_columns_v, float* rows,
                                     hipStream_t stream);
void launch_fill_f64(double* dst, uint64_t n, double value, hipStream_t stream);
}  // namespace omx
