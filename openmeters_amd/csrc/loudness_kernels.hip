// K4 loudness_bs1770: BS.1770 K-weighting (one 4th-order f64 TDF-II), four Kahan-Babuska-Neumaier
// sliding sums over a shared f64 ring, 4x / 2x polyphase true peak, per-block snapshot.
// reference src/visuals/loudness/processor.rs:123-162, :253-311 and src/dsp.rs:264-371.
//
// One thread per (stream, channel) — the recurrences are sequential in time — laid out 8 lanes per
// stream so the position-weighted channel sum of a snapshot is an in-order shuffle walk.  The ring
// is [slot][stream*8 + channel]: the slot index is identical for every channel of a lock-step bank
// (lazy activation with leading zeros == eager state fed zeros, loudness/processor.rs:400-417), so
// each expiring-value read and each ring write is one coalesced f64 row per wave.
// Built with -ffp-contract=off: the f64 filter and the KBN sums round exactly like the scalar code.
#include "loudness.hpp"

namespace omx {

__device__ __forceinline__ void kbn_add(double& sum, double& corr, double v) {  // dsp.rs:277-285
    const double next = sum + v;
    corr += (fabs(sum) >= fabs(v)) ? (sum - next) + v : (v - next) + sum;
    sum = next;
}
__device__ __forceinline__ float power_to_db_f(float power, float floor) {  // level.rs:28-34
    return power > 0.0f ? fmaxf(logf(power) * 4.3429448f, floor) : floor;
}
__device__ __forceinline__ float mean_square_to_lufs(double ms, float floor) {  // loudness/processor.rs:57-66
    return ms > 0.0 ? (float)fmax(fma(log10(ms), 10.0, -0.691), (double)floor) : floor;
}

template <int B>  // B = samples whose PCM + expiring ring values are prefetched together
__global__ __launch_bounds__(64) void loudness_kernel(LoudnessArgs a) {
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;  // stream * 8 + channel
    const uint32_t s = gid >> 3, c = gid & 7;
    const bool live = s < a.n_streams && c < a.channels;
    const uint32_t row = a.n_streams * 8;
    LoudnessChannelState st;
    if (live) st = a.state[gid];
    else memset(&st, 0, sizeof(st));
    const float* pcm = a.pcm + ((uint64_t)(live ? s : 0) * a.frames_total) * a.channels + (live ? c : 0);
    uint64_t seen = a.frames_seen;
    uint64_t head = seen % a.ring_len;
    uint64_t refresh[kLoudnessWindows];
#pragma unroll
    for (int w = 0; w < kLoudnessWindows; ++w) refresh[w] = seen % a.capacities[w];  // dsp.rs:363

    for (uint32_t blk = 0; blk < a.n_blocks; ++blk) {
        for (uint32_t f0 = 0; f0 < a.block_frames; f0 += B) {
            float x[B];
            double old[B][kLoudnessWindows];
            const uint32_t nb = min((uint32_t)B, a.block_frames - f0);
#pragma unroll
            for (int k = 0; k < B; ++k) {
                x[k] = 0.0f;
                if (live && (uint32_t)k < nb) {
                    x[k] = pcm[((uint64_t)blk * a.block_frames + f0 + k) * a.channels];
#pragma unroll
                    for (int w = 0; w < kLoudnessWindows; ++w) {  // dsp.rs:336-338 (read before this sample's store)
                        const uint64_t cap = a.capacities[w];
                        old[k][w] = (seen + k >= cap) ? a.ring[((head + k + a.ring_len - cap) % a.ring_len) * row + gid] : 0.0;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                if ((uint32_t)k >= nb) break;
                const float sample = x[k];
                // ---- k_weighted (:153-162)
                const double xd = (double)sample;
                const double y = a.b[0] * xd + st.filter[0];
                st.filter[0] = a.b[1] * xd + st.filter[1] - a.a[1] * y;
                st.filter[1] = a.b[2] * xd + st.filter[2] - a.a[2] * y;
                st.filter[2] = a.b[3] * xd + st.filter[3] - a.a[3] * y;
                st.filter[3] = a.b[4] * xd - a.a[4] * y;
                const double filtered = (double)(float)y;  // rounded to f32 before squaring (:161, :276-277)
                double value = filtered * filtered;
                // ---- WindowedMeans::push (dsp.rs:324-357)
                if (!isfinite(value)) value = 0.0;
#pragma unroll
                for (int w = 0; w < kLoudnessWindows; ++w) {
                    const uint64_t cap = a.capacities[w];
                    kbn_add(st.sums[w][0], st.corrections[w][0], value);
                    kbn_add(st.sums[w][1], st.corrections[w][1], value);
                    if (seen >= cap) kbn_add(st.sums[w][0], st.corrections[w][0], -old[k][w]);
                    if (++refresh[w] == cap) {  // CompensatedPair::refresh (dsp.rs:287-289)
                        st.sums[w][0] = st.sums[w][1];
                        st.sums[w][1] = 0.0;
                        st.corrections[w][0] = st.corrections[w][1];
                        st.corrections[w][1] = 0.0;
                        refresh[w] = 0;
                    }
                }
                if (live) a.ring[head * row + gid] = value;
                head = head + 1 == a.ring_len ? 0 : head + 1;
                ++seen;
                // ---- TruePeakMeter::process (:123-150); delay[0] = newest
                st.peak = fmaxf(st.peak, fabsf(sample));
                if (a.delay_len == 12) {
#pragma unroll
                    for (int i = 11; i > 0; --i) st.delay[i] = st.delay[i - 1];
                    st.delay[0] = sample;
                    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        o0 += st.delay[i] * a.fir4[i][0];
                        o1 += st.delay[i] * a.fir4[i][1];
                        o2 += st.delay[i] * a.fir4[i][2];
                    }
                    st.peak = fmaxf(fmaxf(fmaxf(st.peak, fabsf(o0)), fabsf(o1)), fabsf(o2));
                } else if (a.delay_len == 24) {
#pragma unroll
                    for (int i = 23; i > 0; --i) st.delay[i] = st.delay[i - 1];
                    st.delay[0] = sample;
                    float o = 0.0f;
#pragma unroll
                    for (int i = 0; i < 24; ++i) o += st.delay[i] * a.fir2[i];
                    st.peak = fmaxf(st.peak, fabsf(o));
                }
            }
        }
        // ---- end of block: denormal flush (:281-285) and snapshot (:287-310)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (fabs(st.filter[i]) < 1.0e-30) st.filter[i] = 0.0;
        double mean[kLoudnessWindows];
#pragma unroll
        for (int w = 0; w < kLoudnessWindows; ++w) {  // dsp.rs:367-370
            const uint64_t cnt = max(min(min(seen, a.ring_len), a.capacities[w]), (uint64_t)1);
            mean[w] = (st.sums[w][0] + st.corrections[w][0]) / (double)cnt;
        }
        const float peak = st.peak;
        st.peak = 0.0f;  // std::mem::take (:301)
        // position-weighted channel sums, channel order preserved (:292-296)
        double short_term = 0.0, momentary = 0.0;
        const int lane0 = threadIdx.x & ~7;
        for (uint32_t k = 0; k < a.channels; ++k) {
            const double ms = __shfl(mean[0], lane0 + (int)k);
            const double mm = __shfl(mean[1], lane0 + (int)k);
            short_term += ms * a.weights[k];
            momentary += mm * a.weights[k];
        }
        if (live) {
            omx_loudness_snapshot* snap = a.snapshots + (uint64_t)s * a.n_blocks + blk;
            snap->rms_fast_db[c] = power_to_db_f((float)mean[2], a.floor_db);
            snap->rms_slow_db[c] = power_to_db_f((float)mean[3], a.floor_db);
            snap->true_peak_db[c] = power_to_db_f(peak * peak, a.floor_db);
            if (c == 0) {
                snap->short_term_loudness = mean_square_to_lufs(short_term, a.floor_db);
                snap->momentary_loudness = mean_square_to_lufs(momentary, a.floor_db);
                snap->channel_count = a.channels;
                snap->_pad = 0;
                for (int i = 0; i < OMX_MAX_CHANNELS; ++i) snap->positions[i] = a.positions[i];
                for (uint32_t i = a.channels; i < OMX_MAX_CHANNELS; ++i) {  // LoudnessSnapshot::with_floor (:197-207)
                    snap->rms_fast_db[i] = a.floor_db;
                    snap->rms_slow_db[i] = a.floor_db;
                    snap->true_peak_db[i] = a.floor_db;
                }
            }
        }
    }
    if (live) a.state[gid] = st;
}

void launch_loudness(const LoudnessArgs& a, hipStream_t stream) {
    if (a.n_streams == 0 || a.n_blocks == 0) return;
    const uint32_t threads = a.n_streams * 8;
    const dim3 grid((threads + 63) / 64);
    uint64_t min_cap = a.capacities[0];
    for (int w = 1; w < kLoudnessWindows; ++w) min_cap = std::min(min_cap, a.capacities[w]);
    // prefetching B expiring values is only valid when no window is shorter than the batch
    if (min_cap >= 8) hipLaunchKernelGGL(loudness_kernel<8>, grid, dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(loudness_kernel<1>, grid, dim3(64), 0, stream, a);
}

}  // namespace omx
