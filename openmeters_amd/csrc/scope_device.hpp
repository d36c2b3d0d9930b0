// Device-side pieces shared by the oscilloscope kernels (oscilloscope_kernels.hip: the single-pass kernel of the configurations
// whose autocorrelation is not an 8192-point transform; scope_fast_kernels.hip: the estimate / trigger kernels of the ones that are).
// reference src/visuals/oscilloscope/processor.rs — the scalar helpers (:14-19, :184-204, :238-247), the uniform trigger
// bookkeeping (StableTrigger::unlock / stabilize, :298-304, :336-356) and the wave-level reductions.
#pragma once
#include "oscilloscope.hpp"

namespace omx {
namespace {

constexpr float F32_EPS = 1.1920929e-7f;
constexpr float NEG_INF = -__builtin_huge_valf();
// PeriodEstimator (:86-91)
constexpr float MIN_HZ = 20.0f, MAX_HZ = 8000.0f, MIN_SIGNAL_PEAK = 0.001f, MIN_PERIODICITY = 0.5f, PEAK_CUTOFF = 0.93f;
// StableTrigger (:285-296)
constexpr float SEARCH_PERIODS = 1.5f, NORMALIZE_FLOOR = 0.01f, MEAN_RESPONSIVENESS = 0.25f, EDGE_STRENGTH = 1.0f,
                BUFFER_RESPONSIVENESS = 0.5f, BUFFER_FALLOFF_PERIODS = 0.5f, BUFFER_RETUNE_SEMITONES = 1.0f,
                SLOPE_WIDTH_PERIODS = 0.25f, RESET_BELOW_MATCH = 0.3f, WINDOW_SECONDS = 0.04f, MIN_CYCLES = 2.0f;
constexpr uint32_t MAX_MISSED_PERIODS = 4;

struct View {  // a contiguous logical slice of a trace ring (positions modulo the ring: 32-bit index arithmetic per access)
    const float* ring;
    uint32_t start, mask;
    uint32_t n;
    __device__ __forceinline__ float at(uint32_t i) const { return ring[(start + i) & mask]; }
    __device__ __forceinline__ View sub(uint32_t off, uint32_t len) const { return View{ring, (start + off) & mask, mask, len}; }
};

__device__ __forceinline__ float rclamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ uint32_t f2u(float x) { return !(x > 0.0f) ? 0u : (x >= 4294967040.0f ? 0xFFFFFFFFu : (uint32_t)x); }
__device__ __forceinline__ uint32_t total_order_key(float x) {  // monotone under f32::total_cmp
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float parabolic_refine(float y_prev, float y_curr, float y_next, uint32_t tau) {  // :14-19
    const float denom = y_prev - 2.0f * y_curr + y_next;
    if (fabsf(denom) < F32_EPS) return (float)tau;
    const float delta = 0.5f * (y_prev - y_next) / denom;
    return fmaxf((float)tau + rclamp(delta, -1.0f, 1.0f), 1.0f);
}
__device__ __forceinline__ uint32_t trigger_kernel_len(float period, float rate) {  // :184-189
    return f2u(fmaxf(roundf(fmaxf(rate * WINDOW_SECONDS, period * MIN_CYCLES)), 2.0f));
}
__device__ __forceinline__ float gaussian(uint32_t len, uint32_t index, float std_) {  // :199-204
    if (len <= 1 || std_ <= F32_EPS) return 0.0f;
    const float center = (float)(len - 1) * 0.5f;
    const float r = ((float)index - center) / std_;
    return expf(-0.5f * (r * r));
}
__device__ __forceinline__ float sample_linear_zero(const float* data, uint32_t n, float pos) {  // :238-247
    if (n == 0 || pos < 0.0f || pos > (float)(n - 1)) return 0.0f;
    const uint32_t idx = f2u(pos);
    const float frac = pos - (float)idx;
    if (frac > F32_EPS && idx + 1 < n) return data[idx] + (data[idx + 1] - data[idx]) * frac;
    return data[idx];
}
__device__ __forceinline__ float sample_linear_zero_view(const View& v, float pos) {
    if (v.n == 0 || pos < 0.0f || pos > (float)(v.n - 1)) return 0.0f;
    const uint32_t idx = f2u(pos);
    const float frac = pos - (float)idx;
    if (frac > F32_EPS && idx + 1 < v.n) {
        const float a = v.at(idx), b = v.at(idx + 1);
        return a + (b - a) * frac;
    }
    return v.at(idx);
}

struct Capture {
    int some;
    float span;
    uint32_t start;
    float frac_offset;
};

struct Estimate {
    int some;
    float period, confidence;
};

template <class TS>
__device__ __forceinline__ void trigger_unlock(TS& t) {  // :298-304
    t.has_period = 0;
    t.missed_periods = 0;
    t.ref_len = 0;
    t.reference_period = 0.0f;
    t.mean = 0.0f;
}

template <class TS>
__device__ __forceinline__ Estimate stabilize(TS& t, Estimate detected) {  // :336-356 (uniform scalar code)
    if (!detected.some) {
        if (!t.has_period) return Estimate{0, 0.0f, 0.0f};
        const float p = t.period;
        t.missed_periods = t.missed_periods >= 255 ? 255 : t.missed_periods + 1;
        if (t.missed_periods > MAX_MISSED_PERIODS) {
            trigger_unlock(t);
            return Estimate{0, 0.0f, 0.0f};
        }
        return Estimate{1, p, 0.0f};
    }
    t.missed_periods = 0;
    if (t.has_period) {
        const float prev = t.period, r = detected.period / prev;
        if (r >= 0.9f && r <= 1.1f) detected.period = prev + 0.35f * (detected.period - prev);
    }
    t.has_period = 1;
    t.period = detected.period;
    return detected;
}

// The projected value of a stereo frame for one trace (util/audio/channel.rs:13-21)
__device__ __forceinline__ float scope_project(uint32_t channel, float left, float right) {
    switch (channel) {
        case OMX_CHANNEL_LEFT: return left;
        case OMX_CHANNEL_RIGHT: return right;
        case OMX_CHANNEL_MID: return (left + right) * 0.5f;
        case OMX_CHANNEL_SIDE: return (left - right) * 0.5f;
        default: return 0.0f;
    }
}

}  // namespace
}  // namespace omx
