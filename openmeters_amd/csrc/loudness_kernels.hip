// K4 loudness_bs1770: BS.1770 K-weighting (one 4th-order f64 TDF-II), four Kahan-Babuska-Neumaier
// sliding sums over a shared f64 ring, 4x / 2x polyphase true peak, per-block snapshot.
// reference src/visuals/loudness/processor.rs:123-162, :253-311 and src/dsp.rs:264-371.
//
// Four lanes per (stream, channel) — the recurrences are sequential in time — 32 lanes per stream, so
// the position-weighted channel sum of a snapshot is an in-order shuffle walk.  The ring
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

// Lane layout: gid = (stream * 8 + channel) * 4 + r.  The four lanes of a channel run the same straight-line code:
//   * the K-weighting filter is evaluated redundantly (17 f64 flops) so every lane has y^2 without a shuffle;
//   * lane r owns sliding window r (its KBN pair, its expiring-value reads), lane 0 also writes the ring;
//   * lane r < 3 owns phase r of the 4x true-peak interpolator (lane 3 runs the same MACs on zero taps).
// That cuts the dependent f64 chain per sample by ~3x and quadruples the resident waves (cfg3: 512 instead of 128).
// Everything inside a batch is branch-free and statically indexed: the true-peak delay line is an extended
// register window (no per-sample shifting), the "window not yet full" case adds -0.0 instead of branching.
template <int DL>
struct LoudLane {
    double sum0, sum1, cor0, cor1, filt[4];
    float hist[DL > 1 ? DL - 1 : 1];  // previous samples, hist[0] = newest
    float taps[DL > 0 ? DL : 1];
    float peak;
    uint32_t head, refresh, unfilled;
};

// Processes NB consecutive samples (x) with their expiring ring values (old, 0.0 while the window is not full).
template <int NB, int DL>
__device__ __forceinline__ void loudness_step(LoudLane<DL>& L, const float (&x)[NB], const double (&old)[NB], const LoudnessArgs& a,
                                              double* ring_col, uint32_t row, uint32_t len, uint32_t cap, bool store_lane) {
    float ext[NB + (DL > 0 ? DL - 1 : 0)];  // ext[NB-1-k] = x[k]; ext[NB + i] = hist[i]
#pragma unroll
    for (int k = 0; k < NB; ++k) ext[NB - 1 - k] = x[k];
    if constexpr (DL > 1) {
#pragma unroll
        for (int i = 0; i < DL - 1; ++i) ext[NB + i] = L.hist[i];
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const float sample = x[k];
        // ---- k_weighted (:153-162)
        const double xd = (double)sample;
        const double y = a.b[0] * xd + L.filt[0];
        L.filt[0] = a.b[1] * xd + L.filt[1] - a.a[1] * y;
        L.filt[1] = a.b[2] * xd + L.filt[2] - a.a[2] * y;
        L.filt[2] = a.b[3] * xd + L.filt[3] - a.a[3] * y;
        L.filt[3] = a.b[4] * xd - a.a[4] * y;
        const double filtered = (double)(float)y;  // rounded to f32 before squaring (:161, :276-277)
        double value = filtered * filtered;
        // ---- WindowedMeans::push for this lane's window (dsp.rs:324-357)
        value = isfinite(value) ? value : 0.0;
        kbn_add(L.sum0, L.cor0, value);
        kbn_add(L.sum1, L.cor1, value);
        kbn_add(L.sum0, L.cor0, -old[k]);  // old[k] == 0.0 until the window is full: adding -0.0 changes nothing
        L.unfilled -= (L.unfilled != 0u) ? 1u : 0u;
        if (++L.refresh == cap) {  // CompensatedPair::refresh (dsp.rs:287-289); rare
            L.sum0 = L.sum1;
            L.sum1 = 0.0;
            L.cor0 = L.cor1;
            L.cor1 = 0.0;
            L.refresh = 0;
        }
        if (store_lane) ring_col[(uint64_t)L.head * row] = value;
        L.head = L.head + 1 == len ? 0 : L.head + 1;
        // ---- TruePeakMeter::process (:123-150): window newest-first = ext[NB-1-k + i]
        L.peak = fmaxf(L.peak, fabsf(sample));
        if constexpr (DL > 0) {
            float o = 0.0f;
#pragma unroll
            for (int i = 0; i < DL; ++i) o += ext[NB - 1 - k + i] * L.taps[i];
            L.peak = fmaxf(L.peak, fabsf(o));
        }
    }
    if constexpr (DL > 1) {
#pragma unroll
        for (int i = 0; i < DL - 1; ++i) L.hist[i] = ext[i];  // newest NB samples first, then what is left of the old history
    }
}

template <int NB, int DL>
__device__ __forceinline__ void loudness_fetch(const LoudLane<DL>& L, float (&x)[NB], double (&old)[NB], const float* pcm,
                                               uint64_t frame0, uint32_t channels, const double* ring_col, uint32_t row, uint32_t len,
                                               uint32_t cap, uint32_t ahead, bool live) {
    // `ahead` = samples between the lane's cursor (head, unfilled) and the first sample fetched here
    uint32_t h = L.head + ahead;
    h = h >= len ? h - len : h;
    const uint32_t unf = L.unfilled > ahead ? L.unfilled - ahead : 0u;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        x[k] = 0.0f;
        old[k] = 0.0;
        if (live) {
            x[k] = pcm[(frame0 + k) * channels];
            if ((uint32_t)k >= unf) {  // dsp.rs:336-338: the value pushed `cap` samples ago, read before this sample's store
                uint32_t pos = h + (uint32_t)k;
                pos = pos >= len ? pos - len : pos;
                const uint32_t idx = pos >= cap ? pos - cap : pos + len - cap;
                old[k] = ring_col[(uint64_t)idx * row];
            }
        }
    }
}

template <int B, int DL>  // B = samples per prefetch batch, DL = true-peak delay length (12: 4x, 24: 2x, 0: off)
__global__ __launch_bounds__(64) void loudness_kernel(LoudnessArgs a) {
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
    const uint32_t r = gid & 3, chan = gid >> 2;      // chan = stream * 8 + channel
    const uint32_t s = chan >> 3, c = chan & 7;
    const bool live = s < a.n_streams && c < a.channels;
    const uint32_t row = a.n_streams * 8;
    LoudLane<DL> L;
    L.sum0 = L.sum1 = L.cor0 = L.cor1 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) L.filt[i] = 0.0;
#pragma unroll
    for (int i = 0; i < (DL > 1 ? DL - 1 : 1); ++i) L.hist[i] = 0.0f;
    if (live) {
        const LoudnessChannelState& st = a.state[chan];
        L.sum0 = st.sums[r][0];
        L.sum1 = st.sums[r][1];
        L.cor0 = st.corrections[r][0];
        L.cor1 = st.corrections[r][1];
#pragma unroll
        for (int i = 0; i < 4; ++i) L.filt[i] = st.filter[i];
        if constexpr (DL > 1) {
#pragma unroll
            for (int i = 0; i < DL - 1; ++i) L.hist[i] = st.delay[i];
        }
    }
    L.peak = 0.0f;  // always 0 at a block boundary: every snapshot takes it (:301)
#pragma unroll
    for (int i = 0; i < (DL > 0 ? DL : 1); ++i) {  // 4x: phase r (zeros for r == 3); 2x: all 24 taps on lane 0
        float t = 0.0f;
        if constexpr (DL == 12) t = r < 3 ? a.fir4[i][r < 3 ? r : 0] : 0.0f;
        if constexpr (DL == 24) t = r == 0 ? a.fir2[i] : 0.0f;
        L.taps[i] = t;
    }
    const float* pcm = a.pcm + ((uint64_t)(live ? s : 0) * a.frames_total) * a.channels + (live ? c : 0);
    const uint32_t len = (uint32_t)a.ring_len, cap = (uint32_t)a.capacities[r];
    L.head = (uint32_t)(a.frames_seen % a.ring_len);
    L.refresh = (uint32_t)(a.frames_seen % a.capacities[r]);                                             // dsp.rs:363
    L.unfilled = a.frames_seen >= a.capacities[r] ? 0u : (uint32_t)(a.capacities[r] - a.frames_seen);  // pushes until count >= cap
    uint64_t seen = a.frames_seen;
    double* ring_col = a.ring + chan;
    const bool store_lane = live && r == 0;
    const uint32_t full = a.block_frames / B, tail = a.block_frames % B;

    for (uint32_t blk = 0; blk < a.n_blocks; ++blk) {
        const uint64_t f_blk = (uint64_t)blk * a.block_frames;
        // full batches, two per iteration so the prefetch buffers swap roles without register copies: the loads of
        // batch n+1 are issued before batch n is computed (HBM round trip hidden behind ~8 samples of f64 work)
        float xa[B], xb[B];
        double oa[B], ob[B];
        if (full > 0) loudness_fetch<B, DL>(L, xa, oa, pcm, f_blk, a.channels, ring_col, row, len, cap, 0, live);
        uint32_t q = 0;
        for (; q + 2 <= full; q += 2) {
            loudness_fetch<B, DL>(L, xb, ob, pcm, f_blk + (uint64_t)(q + 1) * B, a.channels, ring_col, row, len, cap, B, live);
            loudness_step<B, DL>(L, xa, oa, a, ring_col, row, len, cap, store_lane);
            if (q + 2 < full)
                loudness_fetch<B, DL>(L, xa, oa, pcm, f_blk + (uint64_t)(q + 2) * B, a.channels, ring_col, row, len, cap, B, live);
            loudness_step<B, DL>(L, xb, ob, a, ring_col, row, len, cap, store_lane);
        }
        if (q < full) loudness_step<B, DL>(L, xa, oa, a, ring_col, row, len, cap, store_lane);
        for (uint32_t k = 0; k < tail; ++k) {  // block_frames % B leftover samples, one at a time
            float x1[1];
            double o1[1];
            loudness_fetch<1, DL>(L, x1, o1, pcm, f_blk + (uint64_t)full * B + k, a.channels, ring_col, row, len, cap, 0, live);
            loudness_step<1, DL>(L, x1, o1, a, ring_col, row, len, cap, store_lane);
        }
        seen += a.block_frames;

        // ---- end of block: denormal flush (:281-285) and snapshot (:287-310)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (fabs(L.filt[i]) < 1.0e-30) L.filt[i] = 0.0;
        const uint64_t cnt = max(min(min(seen, a.ring_len), a.capacities[r]), (uint64_t)1);  // dsp.rs:367-370
        const double mean_r = (L.sum0 + L.cor0) / (double)cnt;
        // channel-level values: means of the four windows (lanes 4q..4q+3) and the max of the lane peaks
        const int lane = threadIdx.x, chan_lane0 = lane & ~3;
        const double mean_fast = __shfl(mean_r, chan_lane0 + 2), mean_slow = __shfl(mean_r, chan_lane0 + 3);
        float pk = fmaxf(L.peak, __shfl_xor(L.peak, 1));
        pk = fmaxf(pk, __shfl_xor(pk, 2));
        L.peak = 0.0f;  // std::mem::take (:301)
        // position-weighted channel sums in channel order (:292-296): lane (stream_lane0 + 4k + w) holds window w of channel k
        double short_term = 0.0, momentary = 0.0;
        const int stream_lane0 = lane & ~31;
        for (uint32_t k = 0; k < a.channels; ++k) {
            const double ms = __shfl(mean_r, stream_lane0 + 4 * (int)k + 0);
            const double mm = __shfl(mean_r, stream_lane0 + 4 * (int)k + 1);
            short_term += ms * a.weights[k];
            momentary += mm * a.weights[k];
        }
        if (live && r == 0) {
            omx_loudness_snapshot* snap = a.snapshots + (uint64_t)s * a.n_blocks + blk;
            snap->rms_fast_db[c] = power_to_db_f((float)mean_fast, a.floor_db);
            snap->rms_slow_db[c] = power_to_db_f((float)mean_slow, a.floor_db);
            snap->true_peak_db[c] = power_to_db_f(pk * pk, a.floor_db);
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
    if (live) {
        LoudnessChannelState& st = a.state[chan];
        st.sums[r][0] = L.sum0;
        st.sums[r][1] = L.sum1;
        st.corrections[r][0] = L.cor0;
        st.corrections[r][1] = L.cor1;
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) st.filter[i] = L.filt[i];
            if constexpr (DL > 1) {
#pragma unroll
                for (int i = 0; i < DL - 1; ++i) st.delay[i] = L.hist[i];
            }
            st.peak = 0.0f;
        }
    }
}

template <int DL>
static void launch_loudness_dl(const LoudnessArgs& a, dim3 grid, bool batched, hipStream_t stream) {
    if (batched) hipLaunchKernelGGL((loudness_kernel<8, DL>), grid, dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((loudness_kernel<1, DL>), grid, dim3(64), 0, stream, a);
}

void launch_loudness(const LoudnessArgs& a, hipStream_t stream) {
    if (a.n_streams == 0 || a.n_blocks == 0) return;
    const uint32_t threads = a.n_streams * 32;  // 8 channels x 4 lanes per stream
    const dim3 grid((threads + 63) / 64);
    uint64_t min_cap = a.capacities[0];
    for (int w = 1; w < kLoudnessWindows; ++w) min_cap = std::min(min_cap, a.capacities[w]);
    // prefetching two batches of expiring values is only valid when no window is shorter than two batches
    const bool batched = min_cap >= 16;
    if (a.delay_len == 12) launch_loudness_dl<12>(a, grid, batched, stream);
    else if (a.delay_len == 24) launch_loudness_dl<24>(a, grid, batched, stream);
    else launch_loudness_dl<0>(a, grid, batched, stream);
}

}  // namespace omx
