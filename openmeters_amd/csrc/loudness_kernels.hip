// K4 loudness_bs1770: BS.1770 K-weighting (one 4th-order f64 TDF-II), four Kahan-Babuska-Neumaier
// sliding sums over a shared ring of the f32 K-weighted samples (squared, exactly, by every reader), 4x / 2x polyphase true peak,
// per-block snapshot.
// reference src/visuals/loudness/processor.rs:123-162, :253-311 and src/dsp.rs:264-371.
//
// Four lanes per (stream, channel) — the recurrences are sequential in time — 32 lanes per stream, so
// the position-weighted channel sum of a snapshot is an in-order shuffle walk.  The ring
// is [slot][stream*8 + channel]: the slot index is identical for every channel of a lock-step bank
// (lazy activation with leading zeros == eager state fed zeros, loudness/processor.rs:400-417), so
// each expiring-value read and each ring write is one coalesced f32 row per wave.
// Built with -ffp-contract=off: the f64 filter and the KBN sums round exactly like the scalar code.
#include <type_traits>

#include "loudness.hpp"

namespace omx {

// Ring layout: [group of 64 (stream, channel) slots][ring slot][64].  A workgroup's expiring-value reads and its ring writes then
// walk five sequential 256-byte-row streams inside one contiguous region (the first layout, [ring slot][all slots], put every
// access of a workgroup 64 KiB x n_streams / 1024 apart: one DRAM page and one TLB reach per 512 bytes).
constexpr uint32_t kRingRow = 64;
__device__ __forceinline__ RingT* ring_column(RingT* ring, uint32_t chan, uint64_t ring_len) {
    return ring + (uint64_t)(chan >> 6) * ring_len * kRingRow + (chan & 63u);
}

__device__ __forceinline__ void kbn_add(double& sum, double& corr, double v) {  // dsp.rs:277-285
    // corr += |sum| >= |v| ? (sum - next) + v : (v - next) + sum;  picking (big, small) first evaluates one branch's two f64
    // operations instead of both branches' four (f64 issues at half rate): same operands, same order, same bits
    const double next = sum + v;
    const bool sum_is_big = fabs(sum) >= fabs(v);
    const double big = sum_is_big ? sum : v, small = sum_is_big ? v : sum;
    corr += (big - next) + small;
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
// MODE 0: everything in one lane (K-weighting + window + true-peak phase);  MODE 1: K-weighting + window only (the true
// peak of the call is computed by the true-peak workgroups of the same launch);  MODE 2: true peak only.
template <int NB, int DL, int MODE>
__device__ __forceinline__ void loudness_step(LoudLane<DL>& L, const float (&xr)[NB], const RingT (&oldr)[NB], uint32_t unf, bool live,
                                              const LoudnessArgs& a, RingT* ring_col, uint32_t row, uint32_t len, uint32_t cap,
                                              bool store_lane) {
    float x[NB];
    double old[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {  // the selects that loudness_fetch left out
        x[k] = live ? xr[k] : 0.0f;
        old[k] = (live && (uint32_t)k >= unf) ? ring_square(oldr[k]) : 0.0;
    }
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
        if constexpr (MODE != 2) {
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
        const bool finite = isfinite(value);
        value = finite ? value : 0.0;
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
        if (store_lane) ring_col[(uint64_t)L.head * row] = finite ? (RingT)filtered : (RingT)0;
        L.head = L.head + 1 == len ? 0 : L.head + 1;
        }
        // ---- TruePeakMeter::process (:123-150): window newest-first = ext[NB-1-k + i]
        if constexpr (MODE != 1) L.peak = fmaxf(L.peak, fabsf(sample));
        if constexpr (DL > 0 && MODE != 1) {
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

template <int NB, int DL, int MODE>
__device__ __forceinline__ uint32_t loudness_fetch(const LoudLane<DL>& L, float (&x)[NB], RingT (&old)[NB], const float* pcm,
                                               uint64_t frame0, uint32_t channels, const RingT* ring_col, uint32_t row, uint32_t len,
                                               uint32_t cap, uint32_t ahead, bool live) {
    // `ahead` = samples between the lane's cursor (head, unfilled) and the first sample fetched here
    uint32_t h = L.head + ahead;
    h = h >= len ? h - len : h;
    const uint32_t unf = L.unfilled > ahead ? L.unfilled - ahead : 0u;
    // Branch-free on purpose: a conditional load sits in its own basic block and the compiler drains vmcnt at every join,
    // which serialises the whole prefetch.  Non-live lanes point at stream 0 / column 0 (always valid), the ring index is
    // always inside the ring; what must not be used is discarded by the selects.
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const float xv = pcm[(frame0 + k) * channels];
        uint32_t pos = h + (uint32_t)k;
        pos = pos >= len ? pos - len : pos;
        const uint32_t idx = pos >= cap ? pos - cap : pos + len - cap;
        RingT ov = 0;
        if constexpr (MODE != 2) ov = ring_col[(uint64_t)idx * row];
        x[k] = xv;   // raw: the selects happen where the values are consumed (a select here would wait for the load)
        old[k] = ov;
    }
    return unf;  // samples of this batch that precede the first expiring value (dsp.rs:336-338)
}

template <int B, int DL, int MODE>
__device__ __forceinline__ void loudness_body(const LoudnessArgs& a, uint32_t gid) {
    if (a.run_if && *a.run_if == 0u) return;  // fallback launch of the chunk-parallel path: the PCM was finite
    const uint32_t r = gid & 3, chan = gid >> 2;      // chan = stream * 8 + channel
    const uint32_t s = chan >> a.slot_shift, c = chan & ((1u << a.slot_shift) - 1u);
    const bool live = s < a.n_streams && c < a.channels;
    const uint32_t row = kRingRow;
    // ragged banks: the stream's own block count, sample counter and reset flag (every lane of a stream sits in one wavefront and
    // sees the same values; the block loop below then has a per-lane trip count and its shuffles stay inside the stream's lanes)
    const bool ragged = a.blocks_v != nullptr, in_bank = s < a.n_streams;
    const bool reset = ragged && in_bank && a.reset_v != nullptr && a.reset_v[s] != 0;
    const uint64_t seen0 = ragged ? ((in_bank && !reset) ? a.seen_v[s] : 0ull) : a.frames_seen;
    const uint32_t n_blocks_s = ragged ? (in_bank ? a.blocks_v[s] : 0u) : a.n_blocks;
    // chunk calls: the stream's own block length (per-lane trip counts in the batch loops below; nothing in them crosses lanes)
    const uint32_t block_frames_s = (a.frames_v != nullptr && in_bank) ? a.frames_v[s] : a.block_frames;
    LoudLane<DL> L;
    L.sum0 = L.sum1 = L.cor0 = L.cor1 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) L.filt[i] = 0.0;
#pragma unroll
    for (int i = 0; i < (DL > 1 ? DL - 1 : 1); ++i) L.hist[i] = 0.0f;
    if (live && !reset) {
        const LoudnessChannelState& st = a.state[chan];
        if constexpr (MODE != 2) {
            L.sum0 = st.sums[r][0];
            L.sum1 = st.sums[r][1];
            L.cor0 = st.corrections[r][0];
            L.cor1 = st.corrections[r][1];
#pragma unroll
            for (int i = 0; i < 4; ++i) L.filt[i] = st.filter[i];
        }
        if constexpr (DL > 1 && MODE != 1) {
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
    L.head = (uint32_t)(seen0 % a.ring_len);
    L.refresh = (uint32_t)(seen0 % a.capacities[r]);                                   // dsp.rs:363
    L.unfilled = seen0 >= a.capacities[r] ? 0u : (uint32_t)(a.capacities[r] - seen0);  // pushes until count >= cap
    uint64_t seen = seen0;
    RingT* ring_col = ring_column(a.ring, live ? chan : 0, a.ring_len);  // dead lanes read column 0 (discarded) and never store
    const bool store_lane = live && r == 0;
    const uint32_t full = block_frames_s / B, tail = block_frames_s % B;

    for (uint32_t blk = 0; blk < n_blocks_s; ++blk) {
        const uint64_t f_blk = (uint64_t)blk * block_frames_s;
        // full batches, two per iteration so the prefetch buffers swap roles without register copies: the loads of
        // batch n+1 are issued before batch n is computed (HBM round trip hidden behind ~8 samples of f64 work)
        float xa[B], xb[B];
        RingT oa[B], ob[B];
        uint32_t ua = 0, ub = 0;
        if (full > 0) ua = loudness_fetch<B, DL, MODE>(L, xa, oa, pcm, f_blk, a.channels, ring_col, row, len, cap, 0, live);
        uint32_t q = 0;
        for (; q + 2 <= full; q += 2) {
            ub = loudness_fetch<B, DL, MODE>(L, xb, ob, pcm, f_blk + (uint64_t)(q + 1) * B, a.channels, ring_col, row, len, cap, B, live);
            loudness_step<B, DL, MODE>(L, xa, oa, ua, live, a, ring_col, row, len, cap, store_lane);
            if (q + 2 < full)
                ua = loudness_fetch<B, DL, MODE>(L, xa, oa, pcm, f_blk + (uint64_t)(q + 2) * B, a.channels, ring_col, row, len, cap, B, live);
            loudness_step<B, DL, MODE>(L, xb, ob, ub, live, a, ring_col, row, len, cap, store_lane);
        }
        if (q < full) loudness_step<B, DL, MODE>(L, xa, oa, ua, live, a, ring_col, row, len, cap, store_lane);
        for (uint32_t k = 0; k < tail; ++k) {  // block_frames % B leftover samples, one at a time
            float x1[1];
            RingT o1[1];
            const uint32_t u1 = loudness_fetch<1, DL, MODE>(L, x1, o1, pcm, f_blk + (uint64_t)full * B + k, a.channels, ring_col, row, len,
                                                            cap, 0, live);
            loudness_step<1, DL, MODE>(L, x1, o1, u1, live, a, ring_col, row, len, cap, store_lane);
        }
        seen += block_frames_s;

        // ---- end of block: denormal flush (:281-285) and snapshot (:287-310)
        const int lane = threadIdx.x;
        if constexpr (MODE != 1) {  // true peak: max over the lanes (phases) of the channel, taken at every block (:301)
            float pk = fmaxf(L.peak, __shfl_xor(L.peak, 1));
            pk = fmaxf(pk, __shfl_xor(pk, 2));
            L.peak = 0.0f;  // std::mem::take (:301)
            if (live && r == 0) {
                omx_loudness_snapshot* snap = a.snapshots + (uint64_t)s * a.n_blocks + blk;
                snap->true_peak_db[c] = power_to_db_f(pk * pk, a.floor_db);
                if (c == 0)
                    for (uint32_t i = a.channels; i < OMX_MAX_CHANNELS; ++i) snap->true_peak_db[i] = a.floor_db;  // with_floor (:197-207)
            }
        }
        if constexpr (MODE != 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (fabs(L.filt[i]) < 1.0e-30) L.filt[i] = 0.0;
            const uint64_t cnt = max(min(min(seen, a.ring_len), a.capacities[r]), (uint64_t)1);  // dsp.rs:367-370
            const double mean_r = (L.sum0 + L.cor0) / (double)cnt;
            // channel-level values: means of the four windows (lanes 4q..4q+3)
            const int chan_lane0 = lane & ~3;
            const double mean_fast = __shfl(mean_r, chan_lane0 + 2), mean_slow = __shfl(mean_r, chan_lane0 + 3);
            // position-weighted channel sums in channel order (:292-296): lane (stream_lane0 + 4k + w) holds window w of channel k
            double short_term = 0.0, momentary = 0.0;
            const int stream_lane0 = lane & ~(int)((4u << a.slot_shift) - 1u);
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
                if (c == 0) {
                    snap->short_term_loudness = mean_square_to_lufs(short_term, a.floor_db);
                    snap->momentary_loudness = mean_square_to_lufs(momentary, a.floor_db);
                    snap->channel_count = a.channels;
                    snap->_pad = 0;
                    for (int i = 0; i < OMX_MAX_CHANNELS; ++i) snap->positions[i] = a.positions[i];
                    for (uint32_t i = a.channels; i < OMX_MAX_CHANNELS; ++i) {  // LoudnessSnapshot::with_floor (:197-207)
                        snap->rms_fast_db[i] = a.floor_db;
                        snap->rms_slow_db[i] = a.floor_db;
                    }
                }
            }
        }
    }
    // the stream's counter is written by its K-weighting / window lanes only (they read it above, in the same wavefront); the
    // true-peak lanes of a split launch sit in other workgroups and use nothing that derives from it
    if (MODE != 2 && ragged && live && r == 0 && c == 0) a.seen_v[s] = seen;
    if (live) {
        LoudnessChannelState& st = a.state[chan];
        if constexpr (MODE != 2) {
            st.sums[r][0] = L.sum0;
            st.sums[r][1] = L.sum1;
            st.corrections[r][0] = L.cor0;
            st.corrections[r][1] = L.cor1;
            if (r == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) st.filter[i] = L.filt[i];
            }
        }
        if constexpr (MODE != 1) {
            if (r == 0) {
                if constexpr (DL > 1) {
#pragma unroll
                    for (int i = 0; i < DL - 1; ++i) st.delay[i] = L.hist[i];
                }
                st.peak = 0.0f;
            }
        }
    }
}

template <int B, int DL, bool SPLIT>
__global__ __launch_bounds__(64) void loudness_kernel(LoudnessArgs a) {
    if constexpr (!SPLIT) {
        loudness_body<B, DL, 0>(a, blockIdx.x * 64 + threadIdx.x);
    } else {
        if (blockIdx.x < a.n_meter_blocks) loudness_body<B, DL, 1>(a, blockIdx.x * 64 + threadIdx.x);
        else loudness_body<B, DL, 2>(a, (blockIdx.x - a.n_meter_blocks) * 64 + threadIdx.x);
    }
}

// ---- role-per-wavefront form -------------------------------------------------------------------------------------------
// Meter workgroup = 5 wavefronts over 64 (stream, channel) slots (lane = slot): wavefront 0 runs the K-weighting filter and
// hands (f32-rounded y)^2 to the others through a double-buffered LDS batch; wavefronts 1..4 each own one sliding window
// (its KBN pair, its expiring-value reads, its fields of the snapshot); window 0 — the longest, cap == ring length — also
// stores the new values to the ring right after it has read the expiring ones, exactly like lane 0 of the lane-quad form.
// No wavefront executes another role's instructions, so the per-sample cost is the window's (3 KBN adds), not the sum of
// filter + window + interpolator.  (Five wavefronts on a CU's four SIMDs: one SIMD carries two of them, K-weighting 28 + a
// window's 34 VALU per sample, and paces the workgroup — the meter workgroups alone take the whole 2.5 ms of the launch, the
// true-peak ones 1.2 ms.  Packing two channel groups into one workgroup makes that worse (3.6 ms): the 128 workgroups already
// have a CU each.)  True peak runs in separate workgroups (MODE 2 of loudness_body).  Every per-window
// recurrence is still strictly sequential in time: results are bit-identical to the lane-quad form.
// Workgroup barrier for data handed over through LDS only: waits for this wavefront's LDS traffic, not for its outstanding
// global loads (__syncthreads would drain the expiring-value prefetches of the window wavefronts at every round).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// SIX: six wavefronts instead of five.  Wavefront w runs on SIMD w % 4, so the five-wavefront form puts the K-weighting
// wavefront and a window wavefront (28 + 34 VALU per sample) on one SIMD, and that SIMD paces the workgroup.  Here the `since
// last refresh` Kahan pairs of all four windows move to a wavefront of their own (one round behind the filter, one ahead of the
// windows, which take the pair over through a mailbox when CompensatedPair::refresh fires); the four windows then sit on four
// different SIMDs and the filter / pair wavefronts are the lighter second tenant of two of them.
// Every accumulator still sees the same additions in the same order.
template <int B, int NSUB, int DL, bool SIX>  // B samples per prefetch batch, NSUB batches per barrier round
__global__ __launch_bounds__(SIX ? 384 : 320) void loudness_roles_kernel(LoudnessArgs a) {
    static_assert(NSUB == 4, "four rotating prefetch buffers");
    // one latency-bound wavefront per SIMD: when another kernel's wavefronts share the SIMD (the shard pipeline runs the fused
    // STFT beside this bank) this one should win the issue arbitration — it has nothing else to hide its latency with
    __builtin_amdgcn_s_setprio(3);
    constexpr uint32_t THREADS = SIX ? 384 : 320;
    constexpr uint32_t NBUF = SIX ? 3 : 2, LAG = SIX ? 2 : 1;  // value buffers; rounds between the filter and the windows
    if (a.run_if && *a.run_if == 0u) return;  // fallback launch of the chunk-parallel path (workgroup-uniform)
    if (blockIdx.x >= a.n_meter_blocks) {  // true-peak workgroups: 4 phase lanes per channel
        loudness_body<8, DL, 2>(a, (blockIdx.x - a.n_meter_blocks) * THREADS + threadIdx.x);
        return;
    }
    __shared__ double vals[NBUF][NSUB * B][64];
    __shared__ RingT vals_ring[NBUF][NSUB * B][64];  // the same samples as the ring stores them (read by the storing window wavefront only)
    __shared__ double mailbox[SIX ? 2 : 1][4][2][64];  // [round parity][window][sum, correction][lane]
    // wave-uniform by construction; readfirstlane tells the compiler, so that everything derived from the role (window length,
    // ring slots, refresh counters) lives in SGPRs and the ring accesses take the scalar-base + lane-offset form
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // roles: five-wave form 0 = filter, 1..4 = windows; six-wave form 0..3 = windows (one per SIMD: wavefronts w and w + 4 share
    // a SIMD, tools/microbench/wave_simd.hip), 4 = filter, 5 = `since refresh` pairs — measured over all 30 assignments: 2.13 ms
    // for this one, 2.25-2.5 ms for those that put two windows on one SIMD
    const bool k_role = SIX ? wave == 4 : wave == 0;
    const bool s1_role = SIX && wave == 5;
    const uint32_t chan = blockIdx.x * 64 + lane;  // stream * 8 + channel
    const uint32_t s = chan >> a.slot_shift, c = chan & ((1u << a.slot_shift) - 1u);
    const bool live = s < a.n_streams && c < a.channels;
    const uint32_t full = a.block_frames / B;           // batches per block (block_frames % (B * NSUB) == 0, host-checked)
    const uint64_t total = (uint64_t)a.n_blocks * full;   // batches of the call
    const uint64_t rounds = total / NSUB;                 // barrier rounds carrying data
    const float* pcm = a.pcm + ((uint64_t)(live ? s : 0) * a.frames_total) * a.channels + (live ? c : 0);
    // [ring slot][64] of this workgroup: a uniform base (SGPR pair) + unsigned 32-bit byte offsets (ring slot x 512 + lane x 8,
    // < 2^32 for any ring the host accepts) select the scalar-base + vector-offset addressing form — one VALU per access
    // instead of the seven of a per-lane 64-bit address
    char* group_bytes = reinterpret_cast<char*>(a.ring + (uint64_t)blockIdx.x * a.ring_len * kRingRow);
    const uint32_t lane_bytes = lane * (uint32_t)sizeof(RingT);
    constexpr uint32_t kRowBytes = kRingRow * (uint32_t)sizeof(RingT);
    const uint32_t len = (uint32_t)a.ring_len;

    if (k_role) {
        // ---------------- K-weighting wavefront ----------------
        double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
        if (live) {
            const LoudnessChannelState& st = a.state[chan];
            f0 = st.filter[0];
            f1 = st.filter[1];
            f2 = st.filter[2];
            f3 = st.filter[3];
        }
        float xa[B], xb[B];
        auto fetch = [&](float (&x)[B], uint64_t batch) {
#pragma unroll
            for (int k = 0; k < B; ++k)  // unconditional raw loads (see loudness_fetch); past the end of the call: batch 0 again
                x[k] = pcm[((batch < total ? batch : 0) * B + k) * a.channels];
        };
        auto produce = [&](const float (&x)[B], uint32_t buf, uint32_t sub, uint64_t batch) {
#pragma unroll
            for (int k = 0; k < B; ++k) {  // k_weighted (:153-162)
                const double xd = (double)(live ? x[k] : 0.0f);
                const double y = a.b[0] * xd + f0;
                f0 = a.b[1] * xd + f1 - a.a[1] * y;
                f1 = a.b[2] * xd + f2 - a.a[2] * y;
                f2 = a.b[3] * xd + f3 - a.a[3] * y;
                f3 = a.b[4] * xd - a.a[4] * y;
                const double filtered = (double)(float)y;  // rounded to f32 before squaring (:161, :276-277)
                double value = filtered * filtered;
                const bool finite = isfinite(value);
                value = finite ? value : 0.0;              // WindowedMeans::push (dsp.rs:325)
                vals[buf][sub * B + k][lane] = value;
                vals_ring[buf][sub * B + k][lane] = finite ? (RingT)filtered : (RingT)0;
            }
            if ((batch + 1) % full == 0) {  // denormal flush after a block's last sample (:281-285)
                if (fabs(f0) < 1.0e-30) f0 = 0.0;
                if (fabs(f1) < 1.0e-30) f1 = 0.0;
                if (fabs(f2) < 1.0e-30) f2 = 0.0;
                if (fabs(f3) < 1.0e-30) f3 = 0.0;
            }
        };
        fetch(xa, 0);
        // round i: this wavefront fills buffer i & 1 with batches [i NSUB, (i + 1) NSUB) while the window wavefronts consume
        // round i - 1 from the other buffer; one barrier per round
        for (uint64_t i = 0; i < rounds + LAG; ++i) {
            if (i < rounds) {
                const uint64_t b0 = i * NSUB;
                const uint32_t buf = (uint32_t)(i % NBUF);
#pragma unroll
                for (int sb = 0; sb < NSUB; sb += 2) {
                    fetch(xb, b0 + sb + 1);
                    produce(xa, buf, sb, b0 + sb);
                    fetch(xa, b0 + sb + 2);
                    produce(xb, buf, sb + 1, b0 + sb + 1);
                }
            }
            lds_barrier();
        }
        if (live) {
            LoudnessChannelState& st = a.state[chan];
            st.filter[0] = f0;
            st.filter[1] = f1;
            st.filter[2] = f2;
            st.filter[3] = f3;
        }
        return;
    }

    if (s1_role) {
        // ---------------- `since last refresh` pairs of the four windows (six-wave form) ----------------
        double s1[4], c1[4];
        uint32_t refresh4[4], cap4[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            s1[w] = c1[w] = 0.0;
            cap4[w] = (uint32_t)a.capacities[w];
            refresh4[w] = (uint32_t)(a.frames_seen % a.capacities[w]);
            if (live) {
                s1[w] = a.state[chan].sums[w][1];
                c1[w] = a.state[chan].corrections[w][1];
            }
        }
        for (uint64_t i = 0; i < rounds + LAG; ++i) {
            if (i >= 1 && i <= rounds) {  // round i - 1
                const uint32_t buf = (uint32_t)((i - 1) % NBUF), par = (uint32_t)((i - 1) & 1);
#pragma unroll
                for (int sb = 0; sb < NSUB; ++sb) {
                    double batch[B];
#pragma unroll
                    for (int k = 0; k < B; ++k) batch[k] = vals[buf][sb * B + k][lane];
                    bool any = false;
#pragma unroll
                    for (int w = 0; w < 4; ++w) any = any || refresh4[w] + (uint32_t)B >= cap4[w];
                    auto sweep = [&](auto check_c) {  // instantiated without the refresh test for the (usual) batch that cannot fire it
                        constexpr bool CHECK = decltype(check_c)::value;
#pragma unroll
                        for (int k = 0; k < B; ++k) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                // the pair only ever adds values >= +0.0 to a sum that starts at +0.0: |sum| >= |v| is sum >= v, and
                                // (big, small) = (max, min) — same operands as kbn_add, three instructions less
                                const double next = s1[w] + batch[k];
                                const double big = fmax(s1[w], batch[k]), small = fmin(s1[w], batch[k]);
                                c1[w] += (big - next) + small;
                                s1[w] = next;
                                if constexpr (CHECK) {
                                    if (refresh4[w] + (uint32_t)k + 1u == cap4[w]) {  // CompensatedPair::refresh (dsp.rs:287-289)
                                        mailbox[par][w][0][lane] = s1[w];
                                        mailbox[par][w][1][lane] = c1[w];
                                        s1[w] = 0.0;
                                        c1[w] = 0.0;
                                    }
                                }
                            }
                        }
                    };
                    if (any) sweep(std::true_type{});
                    else sweep(std::false_type{});
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        refresh4[w] += (uint32_t)B;
                        refresh4[w] = refresh4[w] >= cap4[w] ? refresh4[w] - cap4[w] : refresh4[w];
                    }
                }
            }
            lds_barrier();
        }
        if (live) {
            LoudnessChannelState& st = a.state[chan];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                st.sums[w][1] = s1[w];
                st.corrections[w][1] = c1[w];
            }
        }
        return;
    }

    // ---------------- window wavefronts ----------------
    const uint32_t r = SIX ? wave : wave - 1;
    const uint32_t cap = (uint32_t)a.capacities[r];
    double sum0 = 0.0, sum1 = 0.0, cor0 = 0.0, cor1 = 0.0;
    if (live) {
        const LoudnessChannelState& st = a.state[chan];
        sum0 = st.sums[r][0];
        sum1 = st.sums[r][1];
        cor0 = st.corrections[r][0];
        cor1 = st.corrections[r][1];
    }
    uint32_t head = (uint32_t)(a.frames_seen % a.ring_len);
    uint32_t refresh = (uint32_t)(a.frames_seen % a.capacities[r]);
    uint32_t unfilled = a.frames_seen >= a.capacities[r] ? 0u : (uint32_t)(a.capacities[r] - a.frames_seen);
    uint64_t seen = a.frames_seen;
    RingT o0[B], o1[B], o2[B], o3[B];  // expiring samples, fetched four batches (32 samples, several HBM round trips) ahead
    // expiring values of the batch that starts `ahead` samples from the cursor (dsp.rs:336-338); `batch` only gates the tail
    auto fetch_old = [&](RingT (&old)[B], uint32_t ahead, uint64_t batch) -> uint32_t {
        uint32_t h = head + ahead;
        h = h >= len ? h - len : h;
        const uint32_t unf = unfilled > ahead ? unfilled - ahead : 0u;
#pragma unroll
        for (int k = 0; k < B; ++k) {  // unconditional raw loads (see loudness_fetch): the selects happen in consume()
            uint32_t pos = h + (uint32_t)k;
            pos = pos >= len ? pos - len : pos;
            const uint32_t idx = pos >= cap ? pos - cap : pos + len - cap;
            // the ring slot is wave-uniform and the workgroup's 64 columns are contiguous: scalar row address + lane offset
            // (the per-lane 64-bit address arithmetic was 7 VALU per load, a fifth of this wavefront's instructions)
            old[k] = *reinterpret_cast<const RingT*>(group_bytes + (idx * kRowBytes + lane_bytes));
        }
        return batch < total ? unf : (uint32_t)B;  // first sample of the batch that has an expiring value
    };
    auto snapshot = [&](uint32_t blk) {
        const uint64_t cnt = max(min(min(seen, a.ring_len), a.capacities[r]), (uint64_t)1);  // dsp.rs:367-370
        const double mean_r = (sum0 + cor0) / (double)cnt;
        omx_loudness_snapshot* snap = a.snapshots + (uint64_t)(live ? s : 0) * a.n_blocks + blk;
        if (r < 2) {  // position-weighted channel sum in channel order (:292-296): the 8 lanes of a stream
            double acc = 0.0;
            const int stream_lane0 = (int)(lane & ~((1u << a.slot_shift) - 1u));
            for (uint32_t k = 0; k < a.channels; ++k) acc += __shfl(mean_r, stream_lane0 + (int)k) * a.weights[k];
            if (live && c == 0) {
                const float lufs = mean_square_to_lufs(acc, a.floor_db);
                if (r == 0) {
                    snap->short_term_loudness = lufs;
                    snap->channel_count = a.channels;
                    snap->_pad = 0;
                    for (int i = 0; i < OMX_MAX_CHANNELS; ++i) snap->positions[i] = a.positions[i];
                } else {
                    snap->momentary_loudness = lufs;
                }
            }
        } else if (live) {
            float* field = r == 2 ? snap->rms_fast_db : snap->rms_slow_db;
            field[c] = power_to_db_f((float)mean_r, a.floor_db);
            if (c == 0)
                for (uint32_t i = a.channels; i < OMX_MAX_CHANNELS; ++i) field[i] = a.floor_db;  // with_floor (:197-207)
        }
    };
    // Window 0 (cap == ring length) also writes the ring.  The role is wave-uniform: the per-sample `if (store_lane)` of the
    // first form cost every other window wavefront a taken branch per sample, so the batch body is instantiated per role
    // (STORE) and per "can CompensatedPair::refresh fire in this batch" (REFRESH; once per `cap` >= 14 400 pushes) — the common
    // instantiation is straight-line code.
    const bool store_wave = __builtin_amdgcn_readfirstlane((int)(r == 0)) != 0;
    uint32_t round_parity = 0;  // six-wave form: parity of the round being consumed (mailbox slot)
    auto consume_as = [&](auto store_c, auto refresh_c, const RingT (&old)[B], uint32_t first_valid, uint32_t buf, uint32_t sub) {
        constexpr bool STORE = decltype(store_c)::value, REFRESH = decltype(refresh_c)::value;
        // all B values of the batch in one burst of LDS reads (read-per-sample exposed the LDS latency at every sample)
        double batch[B];
#pragma unroll
        for (int k = 0; k < B; ++k) batch[k] = vals[buf][sub * B + k][lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const double value = batch[k];
            const double expiring = (live && (uint32_t)k >= first_valid) ? ring_square(old[k]) : 0.0;
            kbn_add(sum0, cor0, value);
            if constexpr (!SIX) kbn_add(sum1, cor1, value);
            kbn_add(sum0, cor0, -expiring);
            if constexpr (REFRESH) {
                if (refresh + (uint32_t)k + 1u == cap) {  // CompensatedPair::refresh (dsp.rs:287-289)
                    if constexpr (SIX) {  // the pair kept by the `since refresh` wavefront, as of this very sample
                        sum0 = mailbox[round_parity][r][0][lane];
                        cor0 = mailbox[round_parity][r][1][lane];
                    } else {
                        sum0 = sum1;
                        sum1 = 0.0;
                        cor0 = cor1;
                        cor1 = 0.0;
                    }
                }
            }
            if constexpr (STORE) {
                if (live)
                    *reinterpret_cast<RingT*>(group_bytes + ((head + (uint32_t)k >= len ? head + (uint32_t)k - len : head + (uint32_t)k) * kRowBytes + lane_bytes)) =
                        vals_ring[buf][sub * B + k][lane];
            }
        }
    };
    auto consume = [&](const RingT (&old)[B], uint32_t first_valid, uint32_t buf, uint32_t sub, uint64_t batch) {
        using T = std::true_type;
        using F = std::false_type;
        const bool may_refresh = refresh + (uint32_t)B >= cap;
        if (store_wave) {
            if (may_refresh) consume_as(T{}, T{}, old, first_valid, buf, sub);
            else consume_as(T{}, F{}, old, first_valid, buf, sub);
        } else {
            if (may_refresh) consume_as(F{}, T{}, old, first_valid, buf, sub);
            else consume_as(F{}, F{}, old, first_valid, buf, sub);
        }
        refresh += (uint32_t)B;
        refresh = refresh >= cap ? refresh - cap : refresh;
        unfilled = unfilled > (uint32_t)B ? unfilled - (uint32_t)B : 0u;
        head += (uint32_t)B;
        head = head >= len ? head - len : head;
        seen += B;
        if ((batch + 1) % full == 0) snapshot((uint32_t)((batch + 1) / full - 1));
    };
    uint32_t u0 = fetch_old(o0, 0, 0), u1 = fetch_old(o1, B, 1), u2 = fetch_old(o2, 2 * B, 2), u3 = fetch_old(o3, 3 * B, 3);
    for (uint64_t i = 0; i < rounds + LAG; ++i) {
        if (i >= LAG) {  // consume round i - LAG; after a batch is consumed its registers take the batch 4 ahead
            const uint64_t b0 = (i - LAG) * NSUB;
            const uint32_t buf = (uint32_t)((i - LAG) % NBUF);
            round_parity = (uint32_t)((i - LAG) & 1);
            consume(o0, u0, buf, 0, b0);
            u0 = fetch_old(o0, 3 * B, b0 + 4);
            consume(o1, u1, buf, 1, b0 + 1);
            u1 = fetch_old(o1, 3 * B, b0 + 5);
            consume(o2, u2, buf, 2, b0 + 2);
            u2 = fetch_old(o2, 3 * B, b0 + 6);
            consume(o3, u3, buf, 3, b0 + 3);
            u3 = fetch_old(o3, 3 * B, b0 + 7);
        }
        lds_barrier();
    }
    if (live) {
        LoudnessChannelState& st = a.state[chan];
        st.sums[r][0] = sum0;
        st.corrections[r][0] = cor0;
        if constexpr (!SIX) {
            st.sums[r][1] = sum1;
            st.corrections[r][1] = cor1;
        }
    }
}

template <int DL>
static void launch_loudness_dl(LoudnessArgs a, uint32_t blocks, bool batched, bool split, hipStream_t stream) {
    a.n_meter_blocks = blocks;
    if (batched && split) hipLaunchKernelGGL((loudness_kernel<8, DL, true>), dim3(2 * blocks), dim3(64), 0, stream, a);
    else if (batched) hipLaunchKernelGGL((loudness_kernel<8, DL, false>), dim3(blocks), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((loudness_kernel<1, DL, false>), dim3(blocks), dim3(64), 0, stream, a);
}

void launch_loudness(const LoudnessArgs& a, hipStream_t stream) {
    if (a.n_streams == 0 || a.n_blocks == 0) return;
    const uint32_t threads = (a.n_streams << a.slot_shift) * 4;  // slots x 4 lanes per stream
    const uint32_t grid = (threads + 63) / 64;
    uint64_t min_cap = a.capacities[0];
    for (int w = 1; w < kLoudnessWindows; ++w) min_cap = std::min(min_cap, a.capacities[w]);
    // prefetching two batches of expiring values is only valid when no window is shorter than two batches
    const bool batched = min_cap >= 16;
    // split roles while the single-role launch would leave SIMDs idle (one 64-lane workgroup per SIMD fills 1024 of them)
    static const int force = [] {
        const char* e = tuning_env("OMX_LOUDNESS_SPLIT");  // 0 / 1 pins the form (A/B and tests)
        return e ? atoi(e) : -1;
    }();
    if (a.blocks_v) {  // ragged banks: the lane-quad kernel (per-lane ring position, block count and reset flag); true peak in its own
                       // workgroups while that fills idle SIMDs, as in the lock-step launch below
        const bool split_r = a.delay_len != 0 && grid <= 4096;
        if (a.delay_len == 12) launch_loudness_dl<12>(a, grid, batched, split_r, stream);
        else if (a.delay_len == 24) launch_loudness_dl<24>(a, grid, batched, split_r, stream);
        else launch_loudness_dl<0>(a, grid, batched, false, stream);
        return;
    }
    const bool split = force >= 0 ? force != 0 : (a.delay_len != 0 && grid <= 4096);
    // role-per-wavefront form (OMX_LOUDNESS_SPLIT=2 pins it, default when it applies): whole batches per block, 4x interpolator
    const bool roles = a.ring_len * (uint64_t)(kRingRow * sizeof(RingT)) <= 0xFFFFFFFFull &&  // 32-bit byte offsets inside a group's ring
                       (force == 2 || force == 3 || (force < 0 && grid <= 4096)) && min_cap >= 64 && a.block_frames % 32 == 0 && a.delay_len == 12;
    if (roles) {
        LoudnessArgs r = a;
        r.n_meter_blocks = ((a.n_streams << a.slot_shift) + 63) / 64;
        if (force == 2) {  // the five-wavefront form (A/B and tests)
            const uint32_t peak_blocks = ((a.n_streams << a.slot_shift) * 4 + 319) / 320;
            hipLaunchKernelGGL((loudness_roles_kernel<8, 4, 12, false>), dim3(r.n_meter_blocks + peak_blocks), dim3(320), 0, stream, r);
        } else {
            const uint32_t peak_blocks = ((a.n_streams << a.slot_shift) * 4 + 383) / 384;
            hipLaunchKernelGGL((loudness_roles_kernel<8, 4, 12, true>), dim3(r.n_meter_blocks + peak_blocks), dim3(384), 0, stream, r);
        }
        return;
    }
    if (a.delay_len == 12) launch_loudness_dl<12>(a, grid, batched, split, stream);
    else if (a.delay_len == 24) launch_loudness_dl<24>(a, grid, batched, split, stream);
    else launch_loudness_dl<0>(a, grid, batched, false, stream);
}

}  // namespace omx
