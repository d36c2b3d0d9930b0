// K5 threeband_correlator: per stereo frame 16 f32 biquads (LR4 three-band split of L and R) and four
// f64 EMA correlators.  reference src/dsp.rs:422-432, :489-495 and
// src/visuals/stereometer/processor.rs:40-61, :115-140.
// Four lanes per stream run the same straight-line code with per-lane coefficient sets:
//   lane 0  full band      (no filter)
//   lane 1  low   = LP_low(x)
//   lane 2  mid   = LP_high(HP_low(x))
//   lane 3  high  = HP_high(HP_low(x))      (CASCADE_HIGH = true)
// HP_low is evaluated by lanes 2 and 3 on identical inputs, so both see bit-identical `above_low`.
#include "stereometer.hpp"

namespace omx {

__device__ __forceinline__ float biquad_step(const BiquadCoef& c, float (&z)[2], float x) {  // dsp.rs:422-432
    const float out = c.b[0] * x + z[0];
    z[0] = c.b[1] * x - c.a[0] * out + z[1];
    z[1] = c.b[2] * x - c.a[1] * out;
    if (isfinite(out)) return out;
    z[0] = 0.0f;
    z[1] = 0.0f;
    return 0.0f;
}

__global__ __launch_bounds__(64) void stereometer_kernel(StereometerArgs a) {
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;  // stream * 4 + lane
    const uint32_t s = gid >> 2, band = gid & 3;
    if (s >= a.n_streams) return;
    const bool active = band == 0 || a.analyze_bands != 0;
    StereoLaneState st = a.state[gid];
    const BiquadCoef ca = a.stage_a[band], cb = a.stage_b[band];
    const bool use_a = a.use_a[band] != 0, use_b = a.use_b[band] != 0;
    const bool push_history = band == 0 || a.emit_band_points != 0;
    const float* pcm = a.pcm + (uint64_t)s * a.frames_total * a.fmt.channels;
    float* hist = a.history + ((uint64_t)s * 4 + band) * a.hist_frames * 2;
    uint64_t pos = a.hist_pos[band];

    for (uint32_t blk = 0; blk < a.n_blocks; ++blk) {
        if (active) {
            for (uint32_t f = 0; f < a.block_frames; ++f) {
                const float* frame = pcm + ((uint64_t)blk * a.block_frames + f) * a.fmt.channels;
                float left = 0.0f, right = 0.0f;  // dsp.rs:223-249 stereo fold
                for (uint32_t c = 0; c < a.fmt.channels; ++c) {
                    const float v = frame[c];
                    left = left + v * a.fmt.m[c][0];
                    right = right + v * a.fmt.m[c][1];
                }
                float l = left, r = right;
                if (use_a) {  // Cascade<Biquad,2> per channel (dsp.rs:447-451)
                    l = biquad_step(ca, st.z[0][0][0], l);
                    l = biquad_step(ca, st.z[0][1][0], l);
                    r = biquad_step(ca, st.z[0][0][1], r);
                    r = biquad_step(ca, st.z[0][1][1], r);
                }
                if (use_b) {
                    l = biquad_step(cb, st.z[1][0][0], l);
                    l = biquad_step(cb, st.z[1][1][0], l);
                    r = biquad_step(cb, st.z[1][0][1], r);
                    r = biquad_step(cb, st.z[1][1][1], r);
                }
                const double ld = (double)l, rd = (double)r;  // Correlator::update (:40-46)
                st.moments[0] += a.alpha * (ld * rd - st.moments[0]);
                st.moments[1] += a.alpha * (ld * ld - st.moments[1]);
                st.moments[2] += a.alpha * (rd * rd - st.moments[2]);
                if (push_history) {
                    const uint64_t slot = pos % a.hist_frames;
                    hist[slot * 2] = l;
                    hist[slot * 2 + 1] = r;
                    ++pos;
                }
            }
            // flush_denormals once per block (:134-140)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (fabs(st.moments[i]) < 1.0e-30) st.moments[i] = 0.0;
            if (band != 0) {
                float* z = &st.z[0][0][0][0];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (fabsf(z[i]) < 1.0e-20f) z[i] = 0.0f;
            }
        }
        float value = 0.0f;  // Correlator::value (:48-56)
        if (active) {
            const double denom = sqrt(st.moments[1] * st.moments[2]);
            if (denom > 1e-12) {
                const double v = st.moments[0] / denom;
                if (isfinite(v)) value = (float)fmin(fmax(v, -1.0), 1.0);
            }
        }
        a.correlations[((uint64_t)s * a.n_blocks + blk) * 4 + band] = value;
    }
    a.state[gid] = st;
}

void launch_stereometer(const StereometerArgs& a, hipStream_t stream) {
    if (a.n_streams == 0 || a.n_blocks == 0) return;
    const uint32_t threads = a.n_streams * 4;
    hipLaunchKernelGGL(stereometer_kernel, dim3((threads + 63) / 64), dim3(64), 0, stream, a);
}

struct PointsArgs {
    const float* history;
    uint32_t n_streams, hist_frames, target;
    uint64_t hist_pos[4];
    uint32_t band_valid[4];
    float* points;
};
__global__ __launch_bounds__(256) void stereometer_points_kernel(PointsArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t band = blockIdx.y, s = blockIdx.z;
    if (i >= a.target || !a.band_valid[band]) return;
    // data = the newest hist_frames pairs, oldest first; pick data[i * frames / target] (:163-168)
    const uint64_t frames = a.hist_frames;
    const uint64_t idx = (uint64_t)i * frames / a.target;
    const uint64_t oldest = a.hist_pos[band] - frames;
    const float* hist = a.history + ((uint64_t)s * 4 + band) * frames * 2;
    const uint64_t slot = (oldest + idx) % frames;
    float l = hist[slot * 2], r = hist[slot * 2 + 1];
    if (band != 0) {  // BAND_DISPLAY_GAIN (:8, :165-168)
        l *= 0.8f;
        r *= 0.8f;
    }
    float* out = a.points + (((uint64_t)s * 4 + band) * a.target + i) * 2;
    out[0] = l;
    out[1] = r;
}

void launch_stereometer_points(const float* history, uint32_t n_streams, uint32_t hist_frames, const uint64_t hist_pos[4],
                               const uint32_t band_valid[4], uint32_t target, float* points, hipStream_t stream) {
    if (n_streams == 0 || target == 0) return;
    PointsArgs a{};
    a.history = history;
    a.n_streams = n_streams;
    a.hist_frames = hist_frames;
    a.target = target;
    for (int b = 0; b < 4; ++b) {
        a.hist_pos[b] = hist_pos[b];
        a.band_valid[b] = band_valid[b];
    }
    a.points = points;
    hipLaunchKernelGGL(stereometer_points_kernel, dim3((target + 255) / 256, 4, n_streams), dim3(256), 0, stream, a);
}

}  // namespace omx
