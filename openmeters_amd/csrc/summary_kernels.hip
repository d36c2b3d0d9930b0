// K9: state-side summary reductions over device-resident snapshots.
//   spectrum_peaks_kernel   one wavefront per dB row: arg-max over the interior bins inside [min_f, max_f]
//                           (f32::total_cmp order, the LAST maximum wins like Iterator::max_by), then the
//                           parabolic interpolation of spectrum/state.rs:326-356.  HBM-bound: 4 B per bin.
//   loudness_meters_kernel  one lane per stream: visible_values + the three PeakHold recurrences
//                           (loudness/state.rs:36-60, 121-184, 211-246) over the blocks of a call.
#include "summary.hpp"

namespace omx {

namespace {
constexpr float PEAK_EPSILON = 1e-6f;                   // spectrum/state.rs:20
constexpr float METER_DB_LO = -60.0f, METER_DB_HI = 4.0f;  // loudness/render.rs:11
constexpr double PEAK_HOLD_SECONDS = 2.0;               // loudness/state.rs:20
constexpr float PEAK_DECAY_DB_PER_SEC = 60.0f;          // :21

__device__ __forceinline__ bool finite_f(float v) { return fabsf(v) <= 3.4028234663852886e38f; }  // false for NaN / inf

// (total_cmp key, index) packed so that an unsigned 64-bit max picks the larger level and, among equal levels,
// the larger index
__device__ __forceinline__ unsigned long long peak_key(float v, uint32_t i) {
    int32_t b = __float_as_int(v);
    b ^= (int32_t)((uint32_t)(b >> 31) >> 1);
    return ((unsigned long long)((uint32_t)b ^ 0x80000000u) << 32) | (unsigned long long)i;
}
}  // namespace

__global__ __launch_bounds__(256) void spectrum_peaks_kernel(const float* __restrict__ bins, const float* __restrict__ db,
                                                            uint64_t n_bins, uint64_t n_rows, uint64_t row_stride, float min_f,
                                                            float max_f, omx_spectrum_peak* __restrict__ out) {
    const uint64_t row = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const float* d = db + row * row_stride;
    unsigned long long best = 0;  // no finite level maps to key 0 with index 0 (index 0 is never a candidate)
    for (uint64_t i = 1 + (uint64_t)lane; i + 1 < n_bins; i += 64) {
        const float f = bins[i], v = d[i];
        if (f >= min_f && f <= max_f && finite_f(v)) {
            const unsigned long long k = peak_key(v, (uint32_t)i);
            best = k > best ? k : best;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if (lane != 0) return;
    omx_spectrum_peak p{0u, 0u, 0.0f, 0.0f};
    const uint32_t bin = (uint32_t)(best & 0xffffffffull);
    if (best != 0 && n_bins >= 3) {
        const float bin_hz = bins[1] - bins[0];
        const float center_freq = bins[bin], center = d[bin];
        if (finite_f(bin_hz) && bin_hz > 0.0f && finite_f(center_freq)) {
            const float left = d[bin - 1], right = d[bin + 1];
            float offset = 0.0f;
            if (finite_f(left) && finite_f(right)) {
                const float denom = left - 2.0f * center + right;
                if (denom < -PEAK_EPSILON) {
                    offset = 0.5f * (left - right) / denom;
                    offset = offset < -0.5f ? -0.5f : (offset > 0.5f ? 0.5f : offset);
                }
            }
            float level = center;
            if (offset != 0.0f) {
                level = center - 0.25f * (left - right) * offset;
                level = level > center ? level : center;
            }
            const float f = center_freq + offset * bin_hz;
            p.found = 1u;
            p.bin = bin;
            p.freq_hz = f > 0.0f ? f : 0.0f;
            p.level_db = level;
        }
    }
    out[row] = p;
}

void launch_spectrum_peaks(const float* bins, const float* db, uint64_t n_bins, uint64_t n_rows, uint64_t row_stride, float min_f,
                           float max_f, omx_spectrum_peak* out, hipStream_t stream) {
    if (n_rows == 0) return;
    hipLaunchKernelGGL(spectrum_peaks_kernel, dim3((uint32_t)((n_rows + 3) / 4)), dim3(256), 0, stream, bins, db, n_bins, n_rows,
                       row_stride, min_f, max_f, out);
}

__global__ void peak_holds_reset_kernel(omx_peak_hold* holds, uint64_t n, double now) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) holds[i] = omx_peak_hold{METER_DB_LO, 0u, now};
}
void launch_peak_holds_reset(omx_peak_hold* holds, uint64_t n, double now, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(peak_holds_reset_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, holds, n, now);
}

namespace {
enum { SIDE_LEFT = 0, SIDE_RIGHT = 1, SIDE_BOTH = 2, SIDE_NEITHER = 3 };
__device__ __forceinline__ uint32_t fallback_position(uint32_t total, uint32_t index) {  // dsp.rs:36-47
    if (index >= total) return OMX_POS_UNKNOWN;
    if (total == 1) return OMX_POS_MONO;
    if (total == 4 && index >= 2) return index == 2 ? OMX_POS_REAR_LEFT : OMX_POS_REAR_RIGHT;
    if (total == 5 && index >= 3) return index == 3 ? OMX_POS_REAR_LEFT : OMX_POS_REAR_RIGHT;
    return index;  // SURROUND order == enum order
}
__device__ __forceinline__ int channel_side(uint32_t position, uint32_t index, uint32_t total) {  // loudness/state.rs:222-246
    if (position >= OMX_POS_AUX0 || position == OMX_POS_UNKNOWN) position = fallback_position(total, index);
    switch (position) {
        case OMX_POS_FRONT_LEFT: case OMX_POS_REAR_LEFT: case OMX_POS_SIDE_LEFT: return SIDE_LEFT;
        case OMX_POS_FRONT_RIGHT: case OMX_POS_REAR_RIGHT: case OMX_POS_SIDE_RIGHT: return SIDE_RIGHT;
        case OMX_POS_FRONT_CENTER: case OMX_POS_MONO: return SIDE_BOTH;
        default: return SIDE_NEITHER;
    }
}
__device__ __forceinline__ float meter_value(const omx_loudness_snapshot& s, uint32_t mode, uint32_t ch) {  // :121-131
    switch (mode) {
        case OMX_METER_LUFS_SHORT_TERM: return s.short_term_loudness;
        case OMX_METER_LUFS_MOMENTARY: return s.momentary_loudness;
        case OMX_METER_RMS_FAST: return s.rms_fast_db[ch];
        case OMX_METER_RMS_SLOW: return s.rms_slow_db[ch];
        default: return s.true_peak_db[ch];
    }
}
__device__ __forceinline__ float aggregate(const omx_loudness_snapshot& s, uint32_t mode, int wanted) {  // :153-169
    if (mode <= OMX_METER_LUFS_MOMENTARY) return meter_value(s, mode, 0);
    float acc = METER_DB_LO;
    const uint32_t n = s.channel_count < OMX_MAX_CHANNELS ? s.channel_count : OMX_MAX_CHANNELS;
    for (uint32_t ch = 0; ch < n; ++ch) {
        const int side = channel_side(s.positions[ch], ch, n);
        if (side != SIDE_BOTH && side != wanted) continue;
        acc = fmaxf(acc, meter_value(s, mode, ch));
    }
    return acc;
}
}  // namespace

__global__ __launch_bounds__(64) void loudness_meters_kernel(const omx_loudness_snapshot* __restrict__ snapshots, uint64_t n_streams,
                                                            uint64_t n_blocks, uint32_t left_mode, uint32_t right_mode, double t0,
                                                            double dt, omx_peak_hold* __restrict__ holds,
                                                            omx_meter_row* __restrict__ rows) {
    const uint64_t s = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (s >= n_streams) return;
    omx_peak_hold h[3] = {holds[3 * s], holds[3 * s + 1], holds[3 * s + 2]};
    for (uint64_t k = 0; k < n_blocks; ++k) {
        const omx_loudness_snapshot snap = snapshots[s * n_blocks + k];
        const double now = t0 + (double)k * dt;
        omx_meter_row r;
        r.values[0] = aggregate(snap, left_mode, SIDE_LEFT);
        r.values[1] = aggregate(snap, left_mode, SIDE_RIGHT);
        r.values[2] = meter_value(snap, right_mode, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float v = r.values[i];
            v = v < METER_DB_LO ? METER_DB_LO : (v > METER_DB_HI ? METER_DB_HI : v);
            if (v > h[i].db) {  // PeakHold::update :49-59
                h[i].db = v;
                h[i].decay_from = now + PEAK_HOLD_SECONDS;
            } else if (now > h[i].decay_from) {
                const float decay_dt = (float)(now - h[i].decay_from);
                const float d = h[i].db - PEAK_DECAY_DB_PER_SEC * decay_dt;
                h[i].db = d > v ? d : v;
                h[i].decay_from = now;
            }
            r.peaks[i] = h[i].db;
        }
        rows[s * n_blocks + k] = r;
    }
    holds[3 * s] = h[0];
    holds[3 * s + 1] = h[1];
    holds[3 * s + 2] = h[2];
}

void launch_loudness_meters(const omx_loudness_snapshot* snapshots, uint64_t n_streams, uint64_t n_blocks, uint32_t left_mode,
                            uint32_t right_mode, double t0, double dt, omx_peak_hold* holds, omx_meter_row* rows, hipStream_t stream) {
    if (n_streams == 0) return;
    hipLaunchKernelGGL(loudness_meters_kernel, dim3((uint32_t)((n_streams + 63) / 64)), dim3(64), 0, stream, snapshots, n_streams,
                       n_blocks, left_mode, right_mode, t0, dt, holds, rows);
}

// ---- per-stream summary rows of a capture group (the table that is gathered over RCCL once per epoch, sharding.STATS_COLUMNS):
// three launches, each on the stream of the bank whose output it reads, each writing its own columns of rows[s][OMX_STATS_COLUMNS]
__global__ __launch_bounds__(64) void stats_loudness_kernel(const omx_loudness_snapshot* __restrict__ snapshots, const omx_meter_row* __restrict__ meters,
                                                           uint64_t n_streams, uint64_t n_blocks, uint32_t channels, float* __restrict__ rows) {
    const uint64_t s = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (s >= n_streams) return;
    const omx_loudness_snapshot& snap = snapshots[s * n_blocks + n_blocks - 1];  // the newest block
    float* r = rows + s * OMX_STATS_COLUMNS;
    r[0] = snap.momentary_loudness;
    r[1] = snap.short_term_loudness;
    float peak = snap.true_peak_db[0];
    for (uint32_t c = 1; c < channels && c < OMX_MAX_CHANNELS; ++c) peak = fmaxf(peak, snap.true_peak_db[c]);
    r[2] = peak;
    const omx_meter_row& m = meters[s * n_blocks + n_blocks - 1];
    r[10] = m.peaks[0];  // held true-peak bars, left / right (loudness/state.rs:178-217)
    r[11] = m.peaks[1];
}
__global__ __launch_bounds__(64) void stats_stereometer_kernel(const float* __restrict__ correlations, uint64_t n_streams, uint64_t n_blocks,
                                                              float* __restrict__ rows) {
    const uint64_t s = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (s >= n_streams) return;
    const float* c = correlations + (s * n_blocks + n_blocks - 1) * 4;
    float* r = rows + s * OMX_STATS_COLUMNS;
    r[3] = c[0];
    r[4] = c[1];
    r[5] = c[2];
    r[6] = c[3];
}
__global__ __launch_bounds__(256) void stats_spectrogram_kernel(const uint32_t* __restrict__ counts, uint64_t n_streams, uint64_t n_columns,
                                                               float* __restrict__ rows) {
    const uint64_t s = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);  // one wavefront per stream
    if (s >= n_streams) return;
    const int lane = threadIdx.x & 63;
    unsigned long long sum = 0;  // point counts are integers: their sum is exact, the mean is one f32 division
    for (uint64_t c = lane; c < n_columns; c += 64) sum += counts[s * n_columns + c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane != 0) return;
    float* r = rows + s * OMX_STATS_COLUMNS;
    r[7] = (float)n_columns;
    r[8] = (float)sum / (float)n_columns;
    r[9] = (float)counts[s * n_columns + n_columns - 1];
}
// ---- per-capture calls: one lane per capture walks ITS blocks (PeakHold::update on the capture's own sample clock) and rewrites its
// row when it ran any; everything else stays
__global__ __launch_bounds__(64) void stats_loudness_ragged_kernel(const omx_loudness_snapshot* __restrict__ snapshots, uint64_t n_streams,
                                                                  uint64_t max_blocks, const uint32_t* __restrict__ n_blocks_v,
                                                                  const uint32_t* __restrict__ block_frames_v, uint32_t block_frames,
                                                                  float sample_rate, const uint8_t* __restrict__ reset_v, uint32_t left_mode,
                                                                  uint32_t right_mode, uint32_t channels, omx_peak_hold* __restrict__ holds,
                                                                  double* __restrict__ clocks, float* __restrict__ rows) {
    const uint64_t s = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (s >= n_streams) return;
    omx_peak_hold h[3] = {holds[3 * s], holds[3 * s + 1], holds[3 * s + 2]};
    double now = clocks[s];
    if (reset_v && reset_v[s]) {  // LoudnessState::reset_audio: PeakHold::new(DB_RANGE.0, 0) x 3 on a fresh clock
        now = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) h[i] = omx_peak_hold{METER_DB_LO, 0, now};
    }
    const uint32_t n = n_blocks_v[s];
    const double dt = (double)(block_frames_v ? block_frames_v[s] : block_frames) / (double)sample_rate;
    float peaks[3] = {h[0].db, h[1].db, h[2].db};
    for (uint32_t k = 0; k < n; ++k) {
        const omx_loudness_snapshot snap = snapshots[s * max_blocks + k];
        float values[3] = {aggregate(snap, left_mode, SIDE_LEFT), aggregate(snap, left_mode, SIDE_RIGHT), meter_value(snap, right_mode, 0)};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float v = values[i];
            v = v < METER_DB_LO ? METER_DB_LO : (v > METER_DB_HI ? METER_DB_HI : v);
            if (v > h[i].db) {  // PeakHold::update :49-59
                h[i].db = v;
                h[i].decay_from = now + PEAK_HOLD_SECONDS;
            } else if (now > h[i].decay_from) {
                const float decay_dt = (float)(now - h[i].decay_from);
                const float d = h[i].db - PEAK_DECAY_DB_PER_SEC * decay_dt;
                h[i].db = d > v ? d : v;
                h[i].decay_from = now;
            }
            peaks[i] = h[i].db;
        }
        now += dt;
    }
    holds[3 * s] = h[0];
    holds[3 * s + 1] = h[1];
    holds[3 * s + 2] = h[2];
    clocks[s] = now;
    if (n == 0) return;
    const omx_loudness_snapshot& snap = snapshots[s * max_blocks + n - 1];  // the capture's newest block
    float* r = rows + s * OMX_STATS_COLUMNS;
    r[0] = snap.momentary_loudness;
    r[1] = snap.short_term_loudness;
    float peak = snap.true_peak_db[0];
    for (uint32_t c = 1; c < channels && c < OMX_MAX_CHANNELS; ++c) peak = fmaxf(peak, snap.true_peak_db[c]);
    r[2] = peak;
    r[10] = peaks[0];
    r[11] = peaks[1];
}
__global__ __launch_bounds__(64) void stats_stereometer_ragged_kernel(const float* __restrict__ correlations, uint64_t n_streams, uint64_t max_blocks,
                                                                     const uint32_t* __restrict__ n_blocks_v, float* __restrict__ rows) {
    const uint64_t s = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (s >= n_streams || n_blocks_v[s] == 0) return;
    const float* c = correlations + (s * max_blocks + n_blocks_v[s] - 1) * 4;
    float* r = rows + s * OMX_STATS_COLUMNS;
    r[3] = c[0];
    r[4] = c[1];
    r[5] = c[2];
    r[6] = c[3];
}
__global__ __launch_bounds__(256) void stats_spectrogram_ragged_kernel(const uint32_t* __restrict__ counts, uint64_t n_streams, uint64_t max_columns,
                                                                      const uint32_t* __restrict__ n_columns_v, float* __restrict__ rows) {
    const uint64_t s = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);  // one wavefront per capture
    if (s >= n_streams) return;
    const int lane = threadIdx.x & 63;
    const uint64_t n = n_columns_v[s];
    unsigned long long sum = 0;
    for (uint64_t c = lane; c < n; c += 64) sum += counts[s * max_columns + c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane != 0) return;
    float* r = rows + s * OMX_STATS_COLUMNS;
    r[7] = (float)n;   // columns of THIS call (0: the capture's other spectrogram columns keep its last call's values)
    if (n == 0) return;
    r[8] = (float)sum / (float)n;
    r[9] = (float)counts[s * max_columns + n - 1];
}
__global__ void fill_f64_kernel(double* dst, uint64_t n, double value) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = value;
}
void launch_fill_f64(double* dst, uint64_t n, double value, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(fill_f64_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, dst, n, value);
}
void launch_stats_loudness_ragged(const omx_loudness_snapshot* snapshots, uint64_t n_streams, uint64_t max_blocks, const uint32_t* n_blocks_v,
                                  const uint32_t* block_frames_v, uint32_t block_frames, float sample_rate, const uint8_t* reset_v,
                                  uint32_t left_mode, uint32_t right_mode, uint32_t channels, omx_peak_hold* holds, double* clocks, float* rows,
                                  hipStream_t stream) {
    if (n_streams == 0) return;
    hipLaunchKernelGGL(stats_loudness_ragged_kernel, dim3((uint32_t)((n_streams + 63) / 64)), dim3(64), 0, stream, snapshots, n_streams, max_blocks,
                       n_blocks_v, block_frames_v, block_frames, sample_rate, reset_v, left_mode, right_mode, channels, holds, clocks, rows);
}
void launch_stats_stereometer_ragged(const float* correlations, uint64_t n_streams, uint64_t max_blocks, const uint32_t* n_blocks_v, float* rows,
                                     hipStream_t stream) {
    if (n_streams == 0 || max_blocks == 0) return;
    hipLaunchKernelGGL(stats_stereometer_ragged_kernel, dim3((uint32_t)((n_streams + 63) / 64)), dim3(64), 0, stream, correlations, n_streams,
                       max_blocks, n_blocks_v, rows);
}
void launch_stats_spectrogram_ragged(const uint32_t* counts, uint64_t n_streams, uint64_t max_columns, const uint32_t* n_columns_v, float* rows,
                                     hipStream_t stream) {
    if (n_streams == 0) return;
    hipLaunchKernelGGL(stats_spectrogram_ragged_kernel, dim3((uint32_t)((n_streams + 3) / 4)), dim3(256), 0, stream, counts, n_streams, max_columns,
                       n_columns_v, rows);
}

void launch_stats_loudness(const omx_loudness_snapshot* snapshots, const omx_meter_row* meters, uint64_t n_streams, uint64_t n_blocks,
                           uint32_t channels, float* rows, hipStream_t stream) {
    if (n_streams == 0 || n_blocks == 0) return;
    hipLaunchKernelGGL(stats_loudness_kernel, dim3((uint32_t)((n_streams + 63) / 64)), dim3(64), 0, stream, snapshots, meters, n_streams,
                       n_blocks, channels, rows);
}
void launch_stats_stereometer(const float* correlations, uint64_t n_streams, uint64_t n_blocks, float* rows, hipStream_t stream) {
    if (n_streams == 0 || n_blocks == 0) return;
    hipLaunchKernelGGL(stats_stereometer_kernel, dim3((uint32_t)((n_streams + 63) / 64)), dim3(64), 0, stream, correlations, n_streams,
                       n_blocks, rows);
}
// the columns of a visual that produced nothing this call (bit c of `columns` = column c): zero, as a cleared table would have left them
__global__ __launch_bounds__(256) void stats_clear_columns_kernel(float* __restrict__ rows, uint64_t n_streams, uint32_t columns) {
    const uint64_t s = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (s >= n_streams) return;
    float* r = rows + s * OMX_STATS_COLUMNS;
#pragma unroll
    for (uint32_t c = 0; c < OMX_STATS_COLUMNS; ++c)
        if (columns & (1u << c)) r[c] = 0.0f;
}
void launch_stats_clear_columns(float* rows, uint64_t n_streams, uint32_t columns, hipStream_t stream) {
    if (n_streams == 0 || columns == 0) return;
    hipLaunchKernelGGL(stats_clear_columns_kernel, dim3((uint32_t)((n_streams + 255) / 256)), dim3(256), 0, stream, rows, n_streams, columns);
}
void launch_stats_spectrogram(const uint32_t* counts, uint64_t n_streams, uint64_t n_columns, float* rows, hipStream_t stream) {
    if (n_streams == 0 || n_columns == 0) return;
    hipLaunchKernelGGL(stats_spectrogram_kernel, dim3((uint32_t)((n_streams + 3) / 4)), dim3(256), 0, stream, counts, n_streams, n_columns,
                       rows);
}

}  // namespace omx
