// Waveform kernel: min/max columns of L/R/Mid/Side with fractional column phase, 12 dB/oct
// three-band split, f32 sliding means with Kahan-Babuska-Neumaier f64 sums for colour and RMS history.
// reference src/visuals/waveform/processor.rs:92-121, :213-298 and src/dsp.rs:264-371, :422-432, :489-495.
// 16 lanes per stream (12 live): lane = channel * 3 + band.  Every lane evaluates the L and R band filters of its
// band (identical inputs -> bit-identical outputs across the four channel lanes) and forms its channel's value,
// so the Mid/Side trackers need no cross-lane traffic.  Built with -ffp-contract=off.
// This one-wavefront form serves what the role-per-wavefront kernel (waveform_roles_kernels.hip) does not: band analysis off, windows
// shorter than 32 samples and >= 2048 streams without RMS history (ragged banks: one stream per workgroup here).  It is paced by one wavefront's instruction issue
// (a dependent VALU instruction every 9 - 10 cycles) and by one memory wait per batch: the next batch's loads ARE issued a batch
// ahead, but the compiler still places an `s_waitcnt vmcnt(0)` right after them (measured: 150 us per 256-frame block with or
// without the prefetch) — which is what the role kernel's memory wavefront removes.
#include <cstdlib>
#include <type_traits>

#include "waveform_device.hpp"

namespace omx {

using namespace wf;

// ANALYZE / HISTORY: the configuration's band analysis and RMS history, compile-time so that a frame's code is one basic block
template <int B, bool ANALYZE, bool HISTORY>
__global__ __launch_bounds__(64) void waveform_kernel(WaveformArgs a) {
    if (a.run_if && *a.run_if == 0u) return;  // fallback launch of the chunk-parallel path: the PCM was finite
    // ragged banks: one stream per workgroup (lanes 0 ... 15), so that its frame count, push count and column phase are
    // workgroup-uniform like the lock-step kernel arguments they replace
    const bool ragged = a.frames_v != nullptr;
    const uint32_t gid = ragged ? blockIdx.x * 16 + (threadIdx.x & 15) : blockIdx.x * 64 + threadIdx.x;  // stream * 16 + lane
    const uint32_t s = gid >> 4, ln = gid & 15;
    const bool live = s < a.n_streams && ln < 12 && (!ragged || threadIdx.x < 16);
    const bool reset_stream = ragged && a.reset_v != nullptr && a.reset_v[s] != 0;
    const uint64_t frames_s = ragged ? a.frames_v[s] : a.frames;                       // frames of this call
    const uint64_t pushes0 = ragged ? (reset_stream ? 0ull : a.pushes_v[s]) : a.pushes;
    const double phase0 = ragged ? (reset_stream ? 0.0 : a.phase_v[s]) : a.column_phase;
    const uint32_t ch = ln / 3, band = ln % 3;
    const uint32_t row = a.n_streams * 16;
    WaveLaneState st;
    memset(&st, 0, sizeof(st));
    if (live && !reset_stream) st = a.state[gid];
    const bool analyze = a.analyze != 0 && live, history = a.track_history != 0 && analyze;
    const BiquadCoef cb = band == 0 ? a.lp_lo : (band == 1 ? a.lp_hi : a.hp_hi);
    const bool use_a = band == 1;
    const ChannelPick pick(ch);
    const float gain = band == 0 ? 1.0f : (band == 1 ? 0.7f : 2.0f);  // BAND_COLOR_GAINS (:22)
    Window wc, wh0, wh1;
    wc.init(st.color, a.color_len, pushes0);
    wh0.init(st.hist[0], a.color_len, pushes0);
    wh1.init(st.hist[1], a.slow_len, pushes0);
    uint32_t head_c = (uint32_t)(pushes0 % a.color_len), head_h = (uint32_t)(pushes0 % a.slow_len);
    uint64_t pushes = pushes0;
    float* cring = a.color_ring + (s < a.n_streams ? gid : 0u);  // lanes past the last stream read column 0 (discarded), store nothing
    float* hring = a.hist_ring + (s < a.n_streams ? gid : 0u);
    const float* pcm = a.pcm + (uint64_t)(s < a.n_streams ? s : 0) * a.frames * a.fmt.channels;
    // PCM goes through LDS, kStage frames at a time, loaded by the stream's 16 lanes in one burst: per-frame global loads would
    // pay the memory latency — PCIe for a single-stream handle, whose block sits in pinned host memory — once per batch
    constexpr uint32_t kStage = 128;
    __shared__ float stage[4][kStage * OMX_MAX_CHANNELS];
    float* my_stage = stage[threadIdx.x >> 4];
    auto refill = [&](uint64_t f) {
        const uint32_t n = (uint32_t)min((uint64_t)kStage, frames_s - f) * a.fmt.channels;
        const float* src = pcm + f * a.fmt.channels;
        const uint32_t l16 = threadIdx.x & 15;
        for (uint32_t e0 = 0; e0 < n; e0 += 16 * 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[min(e0 + (uint32_t)j * 16 + l16, n - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (e0 + (uint32_t)j * 16 + l16 < n) my_stage[e0 + (uint32_t)j * 16 + l16] = v[j];
        }
        __syncthreads();  // one wavefront per workgroup: orders the LDS writes before the other lanes' reads
    };
    double phase = phase0;
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

    // The expiring values of batch i + 1 are requested before batch i is computed (see the header for what that does and does not
    // buy).  Batch i + 1 expires slots that batches i - 1 and i do not store (windows >= 4 B samples long, launch_waveform).
    float nold_c[B], nold_h0[B], nold_h1[B];
    uint32_t fetch_head_c = head_c, fetch_head_h = head_h;
    // Ring stores trail the pushes by one batch: issued right after a batch's pushes they would be the newest memory operations
    // when the next batch waits for its (older) loads, and the wait would cover their round trip too.
    float pend_c[B] = {}, pend_h[B] = {};
    uint32_t store_head_c = head_c, store_head_h = head_h;
    auto flush_stores = [&](uint32_t n) {
        if constexpr (ANALYZE) {
#pragma unroll
            for (int k = 0; k < B; ++k) {
                if ((uint32_t)k < n) {  // n == B except after the call's last batch
                    uint32_t slot = store_head_c + (uint32_t)k;
                    slot = slot >= a.color_len ? slot - a.color_len : slot;
                    if (analyze) cring[(uint64_t)slot * row] = pend_c[k];
                    if constexpr (HISTORY) {
                        uint32_t hs = store_head_h + (uint32_t)k;
                        hs = hs >= a.slow_len ? hs - a.slow_len : hs;
                        if (history) hring[(uint64_t)hs * row] = pend_h[k];
                    }
                }
            }
            store_head_c = (store_head_c + n) % a.color_len;
            store_head_h = (store_head_h + n) % a.slow_len;
        }
    };
    auto fetch = [&]() {
        // every load of the batch is unconditional (clamped frame index, always-valid ring slots; non-live lanes point at
        // stream 0 / their own padding column): a conditional load waits for its data on the spot and serialises the batch.
        // What must not be used is discarded where it is consumed.
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if constexpr (ANALYZE) {
                nold_c[k] = cring[(uint64_t)expiring_index(fetch_head_c, k, a.color_len, a.color_len) * row];
                if constexpr (HISTORY) {
                    nold_h0[k] = hring[(uint64_t)expiring_index(fetch_head_h, k, a.slow_len, a.color_len) * row];
                    nold_h1[k] = hring[(uint64_t)expiring_index(fetch_head_h, k, a.slow_len, a.slow_len) * row];
                }
            }
        }
        if constexpr (ANALYZE) {  // the ring heads after this batch (a whole batch: a short one is the call's last)
            fetch_head_c = (fetch_head_c + (uint32_t)B) % a.color_len;
            fetch_head_h = (fetch_head_h + (uint32_t)B) % a.slow_len;
        }
    };
    if (B > 1 && frames_s) fetch();  // B == 1 serves windows shorter than 16 samples: each frame's loads follow the previous frame's stores
    // hp_lo runs on every lane and is selected by band (a divergent `if (use_a)` costs the other bands the same instructions
    // anyway); its state only matters on the mid-band lanes
    for (uint64_t f0 = 0; f0 < frames_s; f0 += B) {
        const uint32_t nb = (uint32_t)min((uint64_t)B, frames_s - f0);
        if (B == 1) fetch();
        if (f0 % kStage == 0) refill(f0);
        float lr[B][2];
        float old_c[B] = {}, old_h0[B] = {}, old_h1[B] = {};
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const uint32_t kc = (uint32_t)k < nb ? (uint32_t)k : nb - 1u;
            const float* frame = my_stage + ((uint32_t)(f0 % kStage) + kc) * a.fmt.channels;
            float left = 0.0f, right = 0.0f;  // dsp.rs:223-249 stereo fold
            if (two_channels) {  // uniform; the common shape without a runtime trip count
                left = 0.0f + frame[0] * a.fmt.m[0][0] + frame[1] * a.fmt.m[1][0];
                right = 0.0f + frame[0] * a.fmt.m[0][1] + frame[1] * a.fmt.m[1][1];
            } else {
                for (uint32_t c = 0; c < a.fmt.channels; ++c) {
                    const float v = frame[c];
                    left = left + v * a.fmt.m[c][0];
                    right = right + v * a.fmt.m[c][1];
                }
            }
            lr[k][0] = left;
            lr[k][1] = right;
            if constexpr (ANALYZE) old_c[k] = nold_c[k];
            if constexpr (HISTORY) {
                old_h0[k] = nold_h0[k];
                old_h1[k] = nold_h1[k];
            }
        }
        if constexpr (B > 1) {
            if (f0) flush_stores((uint32_t)B);  // the previous batch's ring values, after the wait for this batch's loads (above) and not before it
            if (f0 + B < frames_s) fetch();
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
                const float derived = pick(left, right);
                const bool fin = isfinite(derived);
                if constexpr (ANALYZE) {  // :258-272 (non-live lanes compute on a neighbour's frames and store nothing)
                    float xl = isfinite(left) ? left : 0.0f, xr = isfinite(right) ? right : 0.0f;
                    // mid = LP_high(HP_low(x))  (CASCADE_HIGH = false: the high band takes the raw sample)
                    const float hl = biquad_step(a.hp_lo, st.za[0], xl), hr = biquad_step(a.hp_lo, st.za[1], xr);
                    xl = use_a ? hl : xl;
                    xr = use_a ? hr : xr;
                    const float bl = biquad_step(cb, st.zb[0], xl), br = biquad_step(cb, st.zb[1], xr);
                    float v = pick(bl, br);
                    v = fin ? v : 0.0f;
                    // BandTracker::process (:108-121)
                    float cv = fabsf(v) * gain;
                    cv = isfinite(cv) ? cv : 0.0f;
                    wc.template push<CHECK>((double)cv, (uint32_t)k >= unf_c ? (double)old_c[k] : 0.0);
                    pend_c[k] = cv;
                    head_c = head_c + 1 == a.color_len ? 0 : head_c + 1;
                    if constexpr (HISTORY) {
                        float pw = v * v;
                        pw = isfinite(pw) ? pw : 0.0f;
                        wh0.template push<CHECK>((double)pw, (uint32_t)k >= unf_h0 ? (double)old_h0[k] : 0.0);
                        wh1.template push<CHECK>((double)pw, (uint32_t)k >= unf_h1 ? (double)old_h1[k] : 0.0);
                        pend_h[k] = pw;
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
                    if (live && ragged) {
                        if (col < a.max_cols) write_column(a.columns + ((uint64_t)s * a.max_cols + col) * 4 + ch);
                    } else if (live && col >= a.first_kept) {
                        write_column(a.columns + ((uint64_t)s * (a.n_emit - a.first_kept) + (col - a.first_kept)) * 4 + ch);
                    }
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
        if constexpr (B == 1) flush_stores(1u);  // short windows: the next frame may expire this very slot
    }
    if constexpr (B > 1) {
        if (frames_s) flush_stores((uint32_t)(frames_s - (frames_s - 1) / B * B));  // the last batch, whole or short
    }
    // BandFilter::flush_denormals once per block (:321-323)
    if (analyze) {
        float* z = &st.za[0][0];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (fabsf(z[i]) < 1.0e-20f) z[i] = 0.0f;
    }
    const float progress = (float)fmin(fmax(phase, 0.0), 1.0);  // preview (:300-306)
    if (live && (ragged ? progress > 0.0f : a.write_preview != 0)) write_column(a.preview + (uint64_t)s * 4 + ch);
    if (ragged && threadIdx.x == 0 && s < a.n_streams) {
        a.pushes_v[s] = pushes;
        a.phase_v[s] = phase;
        a.cols_v[s] = (uint32_t)min(col, a.max_cols);
        a.progress_v[s] = progress;
    }
    if (live) {
        wc.save(st.color);
        wh0.save(st.hist[0]);
        wh1.save(st.hist[1]);
        a.state[gid] = st;
    }
}

void launch_waveform(const WaveformArgs& a, hipStream_t stream) {
    if (a.n_streams == 0) return;
    static const bool pin_single = std::getenv("OMX_WAVEFORM_SINGLE") != nullptr;
    static const bool ragged_single = tuning_env("OMX_WAVEFORM_RAGGED_SINGLE") != nullptr;  // A/B: ragged calls on the one-wavefront kernel
    if (!pin_single && !(a.frames_v && (ragged_single || a.frames > 0xFFFFFFFFull)) && waveform_roles_applicable(a)) {
        launch_waveform_roles(a, stream);
        return;
    }
    const uint32_t threads = a.n_streams * 16;
    const dim3 grid(a.frames_v ? a.n_streams : (threads + 63) / 64);  // ragged banks: one stream per workgroup
    const bool analyze = a.analyze != 0, history = analyze && a.track_history != 0;
    auto launch = [&](auto b_c) {
        constexpr int B = decltype(b_c)::value;
        if (history) hipLaunchKernelGGL((waveform_kernel<B, true, true>), grid, dim3(64), 0, stream, a);
        else if (analyze) hipLaunchKernelGGL((waveform_kernel<B, true, false>), grid, dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((waveform_kernel<B, false, false>), grid, dim3(64), 0, stream, a);
    };
    if (a.color_len >= 32 && a.slow_len >= 32) launch(std::integral_constant<int, 8>{});
    else launch(std::integral_constant<int, 1>{});
}

}  // namespace omx
