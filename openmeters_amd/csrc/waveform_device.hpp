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
__device__ __forceinline__ uint32_t expiring_index(uint32_t head, uint32_t k, uint32_t len, uint32_t cap) {
    uint32_t pos = head + k;
    pos = pos >= len ? pos - len : pos;
    return pos >= cap ? pos - cap : pos + len - cap;
}
}  // namespace wf
}  // namespace omx
