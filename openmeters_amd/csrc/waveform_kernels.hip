// Waveform kernel: min/max columns of L/R/Mid/Side with fractional column phase, 12 dB/oct
// three-band split, f32 sliding means with Kahan-Babuska-Neumaier f64 sums for colour and RMS history.
// reference src/visuals/waveform/processor.rs:92-121, :213-298 and src/dsp.rs:264-371, :422-432, :489-495.
// 16 lanes per stream (12 live): lane = channel * 3 + band.  Every lane evaluates the L and R band filters of its
// band (identical inputs -> bit-identical outputs across the four channel lanes) and forms its channel's value,
// so the Mid/Side trackers need no cross-lane traffic.  Built with -ffp-contract=off.
#include <type_traits>

#include "waveform.hpp"

namespace omx {

namespace {
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
}  // namespace

// ANALYZE / HISTORY: the configuration's band analysis and RMS history, compile-time so that a frame's code is one basic block
template <int B, bool ANALYZE, bool HISTORY>
__global__ __launch_bounds__(64) void waveform_kernel(WaveformArgs a) {
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;  // stream * 16 + lane
    const uint32_t s = gid >> 4, ln = gid & 15;
    const bool live = s < a.n_streams && ln < 12;
    const uint32_t ch = ln / 3, band = ln % 3;
    const uint32_t row = a.n_streams * 16;
    WaveLaneState st;
    memset(&st, 0, sizeof(st));
    if (live) st = a.state[gid];
    const bool analyze = a.analyze != 0 && live, history = a.track_history != 0 && analyze;
    const BiquadCoef cb = band == 0 ? a.lp_lo : (band == 1 ? a.lp_hi : a.hp_hi);
    const bool use_a = band == 1;
    const float gain = band == 0 ? 1.0f : (band == 1 ? 0.7f : 2.0f);  // BAND_COLOR_GAINS (:22)
    Window wc, wh0, wh1;
    wc.init(st.color, a.color_len, a.pushes);
    wh0.init(st.hist[0], a.color_len, a.pushes);
    wh1.init(st.hist[1], a.slow_len, a.pushes);
    uint32_t head_c = (uint32_t)(a.pushes % a.color_len), head_h = (uint32_t)(a.pushes % a.slow_len);
    uint64_t pushes = a.pushes;
    float* cring = a.color_ring + (s < a.n_streams ? gid : 0u);  // lanes past the last stream read column 0 (discarded), store nothing
    float* hring = a.hist_ring + (s < a.n_streams ? gid : 0u);
    const float* pcm = a.pcm + (uint64_t)(live ? s : 0) * a.frames * a.fmt.channels;
    double phase = a.column_phase;
    uint64_t col = 0;
    const bool minmax_lane = live && band == 0;
    const bool two_channels = a.fmt.channels == 2;

    auto write_column = [&](omx_wave_column* dst) {  // column_for (:213-235) for this lane's fields
        if (minmax_lane) {
            float mn = 0.0f, mx = 0.0f;
            if (st.cur_some) {
                mn = st.cur_min;
                mx = st.cur_max;
                if (st.last_valid) {
                    mn = fminf(mn, st.last_sample);
                    mx = fmaxf(mx, st.last_sample);
                }
            }
            dst->min = mn;
            dst->max = mx;
        }
        float color = 0.0f, rms0 = -140.0f, rms1 = -140.0f;
        if (analyze) {
            color = (float)fmax(wc.mean(pushes, a.color_len), 0.0);
            if (history) {
                rms0 = power_to_db_f((float)fmax(wh0.mean(pushes, a.slow_len), 0.0), -140.0f);
                rms1 = power_to_db_f((float)fmax(wh1.mean(pushes, a.slow_len), 0.0), -140.0f);
            }
        }
        dst->color_bands[band] = color;
        dst->rms_db[0][band] = rms0;
        dst->rms_db[1][band] = rms1;
    };

    // hp_lo runs on every lane and is selected by band (a divergent `if (use_a)` costs the other bands the same instructions
    // anyway); its state only matters on the mid-band lanes
    for (uint64_t f0 = 0; f0 < a.frames; f0 += B) {
        const uint32_t nb = (uint32_t)min((uint64_t)B, a.frames - f0);
        float lr[B][2];
        float old_c[B], old_h0[B], old_h1[B];
        // every load of the batch is unconditional (clamped frame index, always-valid ring slots; non-live lanes point at
        // stream 0 / their own padding column): a conditional load waits for its data on the spot and serialises the batch.
        // What must not be used is discarded where it is consumed.
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const uint32_t kc = (uint32_t)k < nb ? (uint32_t)k : nb - 1u;
            const float* frame = pcm + (f0 + kc) * a.fmt.channels;
            float left = 0.0f, right = 0.0f;  // dsp.rs:223-249 stereo fold
            if (two_channels) {  // uniform; the common shape without a runtime trip count
                const float2 x = *reinterpret_cast<const float2*>(frame);
                left = 0.0f + x.x * a.fmt.m[0][0] + x.y * a.fmt.m[1][0];
                right = 0.0f + x.x * a.fmt.m[0][1] + x.y * a.fmt.m[1][1];
            } else {
                for (uint32_t c = 0; c < a.fmt.channels; ++c) {
                    const float v = frame[c];
                    left = left + v * a.fmt.m[c][0];
                    right = right + v * a.fmt.m[c][1];
                }
            }
            lr[k][0] = left;
            lr[k][1] = right;
            // expiring values, read before this batch's stores (windows >= B samples long)
            old_c[k] = cring[(uint64_t)expiring_index(head_c, k, a.color_len, a.color_len) * row];
            old_h0[k] = hring[(uint64_t)expiring_index(head_h, k, a.slow_len, a.color_len) * row];
            old_h1[k] = hring[(uint64_t)expiring_index(head_h, k, a.slow_len, a.slow_len) * row];
        }
        // samples of this batch that precede a window's first expiring value (dsp.rs:336-338), fixed before the pushes move them
        const uint32_t unf_c = wc.unfilled, unf_h0 = wh0.unfilled, unf_h1 = wh1.unfilled;
        auto samples = [&](auto check_c, auto tail_c, auto emit_c) {
            constexpr bool CHECK = decltype(check_c)::value, TAIL = decltype(tail_c)::value, EMIT = decltype(emit_c)::value;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                if constexpr (TAIL) {
                    if ((uint32_t)k >= nb) break;
                }
                const float left = lr[k][0], right = lr[k][1];
                // derived_frame (:123-125): Left, Right, Mid, Side
                const float derived = ch == 0 ? left : (ch == 1 ? right : (ch == 2 ? (left + right) * 0.5f : (left - right) * 0.5f));
                const bool fin = isfinite(derived);
                if constexpr (ANALYZE) {  // :258-272 (non-live lanes compute on a neighbour's frames and store nothing)
                    float xl = isfinite(left) ? left : 0.0f, xr = isfinite(right) ? right : 0.0f;
                    // mid = LP_high(HP_low(x))  (CASCADE_HIGH = false: the high band takes the raw sample)
                    const float hl = biquad_step(a.hp_lo, st.za[0], xl), hr = biquad_step(a.hp_lo, st.za[1], xr);
                    xl = use_a ? hl : xl;
                    xr = use_a ? hr : xr;
                    const float bl = biquad_step(cb, st.zb[0], xl), br = biquad_step(cb, st.zb[1], xr);
                    float v = ch == 0 ? bl : (ch == 1 ? br : (ch == 2 ? (bl + br) * 0.5f : (bl - br) * 0.5f));
                    v = fin ? v : 0.0f;
                    // BandTracker::process (:108-121)
                    float cv = fabsf(v) * gain;
                    cv = isfinite(cv) ? cv : 0.0f;
                    wc.template push<CHECK>((double)cv, (uint32_t)k >= unf_c ? (double)old_c[k] : 0.0);
                    if (analyze) cring[(uint64_t)head_c * row] = cv;
                    head_c = head_c + 1 == a.color_len ? 0 : head_c + 1;
                    if constexpr (HISTORY) {
                        float pw = v * v;
                        pw = isfinite(pw) ? pw : 0.0f;
                        wh0.template push<CHECK>((double)pw, (uint32_t)k >= unf_h0 ? (double)old_h0[k] : 0.0);
                        wh1.template push<CHECK>((double)pw, (uint32_t)k >= unf_h1 ? (double)old_h1[k] : 0.0);
                        if (history) hring[(uint64_t)head_h * row] = pw;
                        head_h = head_h + 1 == a.slow_len ? 0 : head_h + 1;
                    }
                    ++pushes;
                }
                // ingest_derived (:275-291), as selects
                const bool some = st.cur_some != 0;
                st.cur_min = fin ? (some ? fminf(st.cur_min, derived) : derived) : st.cur_min;
                st.cur_max = fin ? (some ? fmaxf(st.cur_max, derived) : derived) : st.cur_max;
                st.cur_last = fin ? derived : st.cur_last;
                st.cur_has_last = fin ? 1 : (some ? 0 : st.cur_has_last);
                st.cur_some = fin ? 1 : st.cur_some;
                st.last_valid = fin ? st.last_valid : 0;
                phase += a.step;
                if constexpr (!EMIT) continue;  // the caller has replayed the phase additions: no column ends in this batch
                if (phase >= 1.0) {  // emit_column (:237-250); uniform over the wavefront
                    if (live && col >= a.first_kept)
                        write_column(a.columns + ((uint64_t)s * (a.n_emit - a.first_kept) + (col - a.first_kept)) * 4 + ch);
                    if (st.cur_some && st.cur_has_last) {
                        st.last_valid = 1;
                        st.last_sample = st.cur_last;
                    }
                    st.cur_some = 0;
                    st.cur_has_last = 0;
                    ++col;
                    phase -= 1.0;
                }
            }
        };
        const bool may_refresh = wc.refresh + (uint32_t)B >= wc.cap || wh0.refresh + (uint32_t)B >= wh0.cap || wh1.refresh + (uint32_t)B >= wh1.cap;
        // does a column end inside this batch?  Replay the f64 phase additions (they are the reference's, bit for bit)
        bool emits = false;
        {
            double ph = phase;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                ph += a.step;
                emits = emits || ph >= 1.0;
            }
        }
        using T = std::true_type;
        using F = std::false_type;
        if (nb == (uint32_t)B && !emits && !may_refresh) samples(F{}, F{}, F{});  // the straight-line batch
        else if (nb == (uint32_t)B && !emits) samples(T{}, F{}, F{});
        else samples(T{}, T{}, T{});
    }
    // BandFilter::flush_denormals once per block (:321-323)
    if (analyze) {
        float* z = &st.za[0][0];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (fabsf(z[i]) < 1.0e-20f) z[i] = 0.0f;
    }
    if (live && a.write_preview) write_column(a.preview + (uint64_t)s * 4 + ch);  // preview (:300-306)
    if (live) {
        wc.save(st.color);
        wh0.save(st.hist[0]);
        wh1.save(st.hist[1]);
        a.state[gid] = st;
    }
}

void launch_waveform(const WaveformArgs& a, hipStream_t stream) {
    if (a.n_streams == 0) return;
    const uint32_t threads = a.n_streams * 16;
    const dim3 grid((threads + 63) / 64);
    const bool analyze = a.analyze != 0, history = analyze && a.track_history != 0;
    auto launch = [&](auto b_c) {
        constexpr int B = decltype(b_c)::value;
        if (history) hipLaunchKernelGGL((waveform_kernel<B, true, true>), grid, dim3(64), 0, stream, a);
        else if (analyze) hipLaunchKernelGGL((waveform_kernel<B, true, false>), grid, dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((waveform_kernel<B, false, false>), grid, dim3(64), 0, stream, a);
    };
    if (a.color_len >= 8 && a.slow_len >= 8) launch(std::integral_constant<int, 8>{});
    else launch(std::integral_constant<int, 1>{});
}

}  // namespace omx
