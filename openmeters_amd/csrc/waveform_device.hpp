// Device helpers shared by the waveform kernels (waveform_kernels.hip: one wavefront per four streams; waveform_roles_kernels.hip:
// one role per wavefront).  reference src/dsp.rs:264-371 (CompensatedPair, WindowedMeans), :422-432 (Biquad::process).
#pragma once
#include "waveform.hpp"

namespace omx {
namespace wf {
__device__ __forceinline__ void kbn_add(double& sum, double& corr, double v) {  // dsp.rs:277-285
    // (big, small) picked first: one branch's two f64 operations instead of both branches' four — same operands, same order
    const double next = sum + v;
    const bool sum_is_big = fabs(sum) >= fabs(v);
    const double big = sum_is_big ? sum : v, small = sum_is_big ? v : sum;
    corr += (big - next) + small;
    sum = next;
}
// the `since refresh` pair only ever adds values >= +0.0 to a sum that starts at +0.0: |sum| >= |v| is sum >= v and
// (big, small) = (max, min) — same operands again
__device__ __forceinline__ void kbn_add_nonneg(double& sum, double& corr, double v) {
    const double next = sum + v;
    corr += (fmax(sum, v) - next) + fmin(sum, v);
    sum = next;
}
// Biquad::process (dsp.rs:422-432) with the non-finite reset as selects: the lanes of a wavefront carry different channels and
// bands, so a branch here is a divergent one per element and sample
__device__ __forceinline__ float biquad_step(const BiquadCoef& c, float (&z)[2], float x) {
    const float out = c.b[0] * x + z[0];
    const float n0 = c.b[1] * x - c.a[0] * out + z[1];
    const float n1 = c.b[2] * x - c.a[1] * out;
    const bool ok = isfinite(out);
    z[0] = ok ? n0 : 0.0f;
    z[1] = ok ? n1 : 0.0f;
    return ok ? out : 0.0f;
}
__device__ __forceinline__ float power_to_db_f(float power, float floor) {  // level.rs:28-34
    return power > 0.0f ? fmaxf(logf(power) * 4.3429448f, floor) : floor;
}
struct Window {  // one WindowedMeans window of one value
    double s0, s1, c0, c1;
    uint32_t cap, refresh, unfilled;
    // dsp.rs:335-352 for one (window, value).  CHECK = false: the caller has established that CompensatedPair::refresh cannot fire
    // in this batch (refresh + batch < cap), the common case, and gets straight-line code
    template <bool CHECK>
    __device__ __forceinline__ void push(double v, double old) {
        kbn_add(s0, c0, v);
        kbn_add_nonneg(s1, c1, v);  // v is |band value| x gain or a squared band value, NaN / inf already zeroed (:108-121)
        kbn_add(s0, c0, -old);  // old == 0.0 until the window is full
        unfilled -= unfilled != 0u ? 1u : 0u;
        ++refresh;
        if constexpr (CHECK) {
            if (refresh == cap) {
                s0 = s1;
                s1 = 0.0;
                c0 = c1;
                c1 = 0.0;
                refresh = 0;
            }
        }
    }
    __device__ __forceinline__ double mean(uint64_t pushes, uint32_t ring_len) const {  // dsp.rs:367-370
        const uint64_t count = max(min(min(pushes, (uint64_t)ring_len), (uint64_t)cap), (uint64_t)1);
        return (s0 + c0) / (double)count;
    }
    __device__ __forceinline__ void init(const double (&st)[4], uint32_t cap_, uint64_t pushes) {
        s0 = st[0]; s1 = st[1]; c0 = st[2]; c1 = st[3];
        cap = cap_;
        refresh = (uint32_t)(pushes % cap_);
        unfilled = pushes >= cap_ ? 0u : (uint32_t)(cap_ - pushes);
    }
    __device__ __forceinline__ void save(double (&st)[4]) const { st[0] = s0; st[1] = s1; st[2] = c0; st[3] = c1; }
};
// derived_frame (waveform/processor.rs:123-125) for a lane's channel: Left, Right, Mid = (l + r) / 2, Side = (l - r) / 2.  The
// selection is spelled with per-lane bit masks: as nested `ch == 0 ? ... :` the compiler turns it into a switch, i.e. three levels
// of divergent branches per frame (saveexec / cbranch / restore around one or two instructions each), and the frame's code falls
// apart into a dozen basic blocks.  Pure selection of the same four candidates: bit-identical.
struct ChannelPick {
    uint32_t m_right, m_side, m_pair;  // all ones: take right over left / side over mid / the (mid, side) pair over (left, right)
    __device__ __forceinline__ explicit ChannelPick(uint32_t ch) : m_right(ch == 1 ? ~0u : 0u), m_side(ch == 3 ? ~0u : 0u), m_pair(ch >= 2 ? ~0u : 0u) {
        // opaque to the optimiser: with visible 0 / ~0 values it rewrites the bit selects as compare + v_cndmask pairs (ten instructions
        // per pick instead of seven)
        asm volatile("" : "+v"(m_right), "+v"(m_side), "+v"(m_pair));
    }
    static __device__ __forceinline__ uint32_t bits(uint32_t mask, uint32_t if_set, uint32_t if_clear) {  // v_bfi_b32
        return (mask & if_set) | (~mask & if_clear);
    }
    __device__ __forceinline__ float operator()(float left, float right) const {
        const float mid = (left + right) * 0.5f, side = (left - right) * 0.5f;
        const uint32_t lr = bits(m_right, __float_as_uint(right), __float_as_uint(left));
        const uint32_t ms = bits(m_side, __float_as_uint(side), __float_as_uint(mid));
        return __uint_as_float(bits(m_pair, ms, lr));
    }
};
// fminf / fmaxf of two values that are known not to be NaN (the min/max column state only ever takes finite samples): the library
// forms canonicalise both operands first (v_max_f32 x, x, x), three instructions per call
__device__ __forceinline__ float min_finite(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_finite(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t expiring_index(uint32_t head, uint32_t k, uint32_t len, uint32_t cap) {
    uint32_t pos = head + k;
    pos = pos >= len ? pos - len : pos;
    return pos >= cap ? pos - cap : pos + len - cap;
}
}  // namespace wf
}  // namespace omx
