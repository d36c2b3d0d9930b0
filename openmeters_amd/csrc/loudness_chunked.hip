// K4c: the loudness recurrences evaluated chunk-parallel in time (one chunk = one block of a bank call).
//
// The sequential kernels (loudness_kernels.hip) walk a (stream, channel) slot's samples in order: exact reference order, but
// a 1024 x 2-channel shard (BASELINE configs[4]) is 2048 slots = 32 workgroups on a 256-CU part and takes 2.5 ms per 16 384
// frames whatever the arithmetic costs.  What the path computes is linear in its state or has finite memory:
//   K-weighting      one 4th-order f64 TDF-II (reference src/visuals/loudness/processor.rs:153-162): linear
//                    -> pass A (zero-state end state per block), a wave-parallel scan over the blocks with the block
//                       transition T (host, f64), pass B from the TRUE start state
//   WindowedMeans    (src/dsp.rs:264-371) the mean of the newest cap_w squared samples.  The reference keeps it as two
//                    Kahan-Babuska-Neumaier running sums with a periodic exact refresh — an accurate SLIDING evaluation of a
//                    window sum.  Here the same sums come from a prefix over 64-sample sub-block sums (KBN inside a
//                    sub-block, f64 running total Q since the last reset, kept in a per-slot ring): W = Q[end] - Q[start - 1].
//                    Both are accurate to ~1e-15 of the window sum; they differ in the last bits only (bar: 1e-4 dB).
//   true peak        (:123-151) 12- / 24-sample FIR memory: every block is computed from its own samples and the DL - 1
//                    before them, in the reference's accumulation order -> bit-identical to the sequential kernels.
// The K-weighted samples still go to the ring in the sequential kernels' layout (f32, RingT: loudness.hpp), and the channel state (filter, KBN pairs,
// delay line) is written back in their form, so calls may alternate between the two forms.
// Shapes: any channel count (1, 2, 4, 8 through LDS tiles of whole streams; 3, 5, 6, 7 by per-lane reads), block_frames and the
// sample counter multiples of 64; window lengths on the 64-sample grid (48 / 96 / 192 kHz) or off it (44.1 / 88.2 kHz: `tails`).  Non-finite PCM is detected in pass A; every later kernel then leaves the state alone and the caller runs
// the sequential kernel instead (processor.rs has no reset in k_weighted: a NaN poisons the filter for good — order matters).
#include "loudness.hpp"

#include <type_traits>

namespace omx {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));  // two independent f32 lanes of a v_pk_*_f32 instruction
constexpr uint32_t kRow = 64;        // ring row = 64 slots (loudness_kernels.hip)
constexpr int STEP = 16;             // frames per staged PCM tile
constexpr uint32_t SUB = 64;         // samples per sub-block sum

// slot = (stream << slot_shift) + channel, the sequential kernels' numbering (1, 2, 4 or 8 slots per stream: channel counts 3, 5, 6, 7
// leave the slots past `channels` idle), so the state, the rings and every per-slot array here are shared between the two forms
__device__ __forceinline__ bool slot_live(const LoudChunkArgs& a, uint32_t slot) {
    return (slot >> a.slot_shift) < a.n_streams && (slot & ((1u << a.slot_shift) - 1u)) < a.channels;
}
// Ragged calls (omx_loudness_bank_process_ragged; registry.rs:396-418 feeds every capture on its own): stream s runs blocks_v[s] of
// the call's n_blocks block slots from its own sample counter seen_v[s], from a cleared state when reset_v[s] is set.  Lock-step
// calls (blocks_v == nullptr) take the bank's common counter.  One scalar load each per kernel; the block body is the same.
struct SlotCall {
    uint64_t seen;
    uint32_t blocks;
    bool reset;
};
template <bool RAGGED = true>
__device__ __forceinline__ SlotCall slot_call(const LoudChunkArgs& a, uint32_t s) {  // s < n_streams
    if (!RAGGED || !a.blocks_v) return {a.frames_seen, a.n_blocks, false};  // (RAGGED = false: wave-uniform values, scalar registers)
    const bool reset = a.reset_v && a.reset_v[s] != 0;
    return {reset ? 0ull : a.seen_v[s], a.blocks_v[s], reset};
}
// Slot of sub-block `g` in a per-slot running-total ring of q_len entries (q_ring, q_lo, tails).  Four interleaved sub-rings by g % 4:
// a snapshot reads the totals at block boundaries minus per-window constants — with 256-frame blocks every 4th entry — and lane =
// block, so consecutive lanes read consecutive entries of ONE sub-ring (in natural order they were 32 bytes apart: 8 of every 32 bytes
// fetched were used, 164 MB per cfg3 call).  The writers (lane = sub-block) cover four 128-byte runs per store instead of one 512-byte run.
__device__ __forceinline__ uint64_t q_slot(uint64_t g, uint64_t q_len) {
    const uint64_t i = g & (q_len - 1u);
    return (i & 3u) * (q_len >> 2) + (i >> 2);
}
__device__ __forceinline__ void kbn(double& sum, double& corr, double v) {  // dsp.rs:277-285
    const double next = sum + v;
    const bool big_sum = fabs(sum) >= fabs(v);
    const double big = big_sum ? sum : v, small = big_sum ? v : sum;
    corr += (big - next) + small;
    sum = next;
}
__device__ __forceinline__ float power_to_db(float power, float floor) {  // level.rs:28-34
    return power > 0.0f ? fmaxf(logf(power) * 4.3429448f, floor) : floor;
}
__device__ __forceinline__ float ms_to_lufs(double ms, float floor) {  // loudness/processor.rs:57-66
    return ms > 0.0 ? (float)fmax(fma(log10(ms), 10.0, -0.691), (double)floor) : floor;
}
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    return __hiloint2double(__shfl_up(__double2hiint(v), d), __shfl_up(__double2loint(v), d));
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}

// PCM tile of one slot group and one 16-frame step: rows = the 64 / C streams of the group, row stride 17 C floats (the pad
// keeps the per-lane reads conflict-free for C = 1, 2, 4, 8).  One wavefront = one workgroup: LDS traffic is in order.
struct Tile {
    const float* src[4];
    uint32_t dst[4];
    bool live[4];
    float4 pre[4], pre2[4];  // pre2: the second tile in flight (pass A, TWO tiles ahead; see loud_chunk_peak_kernel)
    template <bool RAGGED>
    __device__ __forceinline__ void setup(const LoudChunkArgs& a, uint32_t group, uint32_t c, uint32_t lane) {
        const uint32_t C = a.channels, row_bytes = 64u * C, per_group = 64u / C;
        const uint64_t frame0 = (uint64_t)c * a.block_frames;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const uint32_t byte = (lane + 64u * (uint32_t)n) * 16u;
            const uint32_t row = byte / row_bytes, inrow = byte % row_bytes;
            const uint64_t s = (uint64_t)group * per_group + row;
            live[n] = s < a.n_streams;
            if constexpr (RAGGED) live[n] = live[n] && c < a.blocks_v[s];  // (a stream's unused block slots hold anything)
            src[n] = a.pcm + ((live[n] ? s : 0) * a.frames_total + frame0) * C + inrow / 4u;
            dst[n] = row * 17u * C + inrow / 4u;
        }
    }
    __device__ __forceinline__ void issue(uint32_t step, uint32_t C) {
#pragma unroll
        for (int n = 0; n < 4; ++n) pre[n] = *reinterpret_cast<const float4*>(src[n] + (uint64_t)step * STEP * C);
    }
    __device__ __forceinline__ void issue2(uint32_t step, uint32_t C) {
#pragma unroll
        for (int n = 0; n < 4; ++n) pre2[n] = *reinterpret_cast<const float4*>(src[n] + (uint64_t)step * STEP * C);
    }
    template <bool SECOND = false>
    __device__ __forceinline__ uint32_t stage(float* tile) {  // returns 1 when a staged sample is not finite
        uint32_t bad = 0;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const float4 p = live[n] ? (SECOND ? pre2[n] : pre[n]) : float4{0.0f, 0.0f, 0.0f, 0.0f};
            bad |= (!isfinite(p.x) || !isfinite(p.y) || !isfinite(p.z) || !isfinite(p.w)) ? 1u : 0u;
            tile[dst[n]] = p.x;
            tile[dst[n] + 1] = p.y;
            tile[dst[n] + 2] = p.z;
            tile[dst[n] + 3] = p.w;
        }
        return bad;
    }
};

// Channel counts 3, 5, 6, 7 (the reference's own 5- and 6-channel cases, loudness/processor.rs:366-398): eight slots per stream of
// which `channels` are live, so a group's PCM is not one rectangular tile of floats; every live lane reads its own channel's 16
// samples of the step straight from the interleaved PCM (stride C floats; a stream's lanes share the cache lines of its 16 C
// contiguous floats), the next step's loads in flight behind the current arithmetic.
struct Direct {
    const float* src;
    bool live;
    float pre[STEP];
    __device__ __forceinline__ void setup(const LoudChunkArgs& a, uint32_t chan, bool live_, uint64_t frame0) {
        const uint32_t C = a.channels, s = chan >> a.slot_shift, ch = chan & ((1u << a.slot_shift) - 1u);
        live = live_;
        src = a.pcm + ((uint64_t)(live ? s : 0) * a.frames_total + frame0) * C + (live ? ch : 0);
    }
    __device__ __forceinline__ void issue(uint32_t step, uint32_t C) {
#pragma unroll
        for (int f = 0; f < STEP; ++f) pre[f] = live ? src[((uint64_t)step * STEP + f) * C] : 0.0f;
    }
    __device__ __forceinline__ uint32_t take(float (&x)[STEP]) {  // returns 1 when a sample of this slot is not finite
        uint32_t bad = 0;
#pragma unroll
        for (int f = 0; f < STEP; ++f) {
            x[f] = pre[f];
            bad |= isfinite(x[f]) ? 0u : 1u;
        }
        return bad;
    }
};


// ---- true peak of STEP samples (TruePeakMeter::process, :123-151).  ext[STEP - 1 - k] = x[k] on entry; hist[i] = the i-th newest sample
// before them (updated on return).  The interpolator is 36 (4x) / 24 (2x) multiply-THEN-add pairs per sample — the reference does not
// fuse them (`output[phase] += sample * coefficients[phase]`, :139-143), so neither does this — i.e. 72 / 48 VALU instructions per sample
// when issued one float at a time (round 4: 79 measured).  Two samples half a tile apart are independent sums over the same taps: as
// lanes .x / .y of v_pk_mul_f32 / v_pk_add_f32 each keeps its own products and its own accumulation order (bit-identical peaks) at
// half the instruction count.  The leading `0.0 + p` of every sum is dropped: it can only turn -0.0 into +0.0, and only |o| is used.
template <int DL>
struct TruePeak {
    static constexpr int H = DL > 1 ? DL - 1 : 1;
    // the DL - 1 samples before block c: from the carried delay line (first block) or from the PCM of the call
    static __device__ __forceinline__ void load_history(float (&hist)[H], const LoudChunkArgs& a, uint32_t chan, uint32_t s, uint32_t ch, uint32_t c,
                                                        bool reset, bool live) {
#pragma unroll
        for (int i = 0; i < H; ++i) hist[i] = 0.0f;
        if (DL > 1 && live) {
            if (c == 0) {
#pragma unroll
                for (int i = 0; i < H; ++i) hist[i] = reset ? 0.0f : a.state[chan].delay[i];
            } else {
                const uint32_t C = a.channels;
                const float* p = a.pcm + ((uint64_t)s * a.frames_total + (uint64_t)c * a.block_frames) * C + ch;
#pragma unroll
                for (int i = 0; i < H; ++i) hist[i] = *(p - (int64_t)(i + 1) * C);  // L >= 64 > DL: inside the call
            }
        }
    }
    // (A/B of where the taps live, round 5: as kernel arguments they are scalars and a packed operand needs the splat {t, t} in an SGPR
    // pair — 72 SGPRs for the 4x interpolator, which beside pass B's K-weighting coefficients overflows the scalar file: 208
    // v_readlane_b32 per 16 samples fetch spilled SGPRs back, 158 VGPRs.  The alternatives measured worse: all 36 taps as VGPR pairs
    // read through op_sel 226 VGPRs; one phase's 12 taps at a time over eight position chains 183 VGPRs and dependent-issue stalls.)
    static __device__ __forceinline__ void step(float (&ext)[STEP + H], float (&hist)[H], float& peak, const LoudChunkArgs& a) {
#pragma unroll
        for (int i = 0; i < H; ++i) ext[STEP + i] = hist[i];
        constexpr int HALF = STEP / 2;
        v2f pair[HALF + H];  // pair[j] = {ext[j], ext[j + HALF]}: window position j of the newer and of the older half
#pragma unroll
        for (int j = 0; j < HALF + H; ++j) pair[j] = v2f{ext[j], ext[j + HALF]};
        // (three phases of one window position = three independent chains, written side by side: a dependent v_pk_*_f32 waits
        // 8 cycles, an independent one issues after 4 — tools/microbench/issue_rate.hip — and the scheduler keeps source order)
#pragma unroll
        for (int m = 0; m < HALF; ++m) {
            peak = fmaxf(fmaxf(peak, fabsf(ext[m])), fabsf(ext[m + HALF]));  // (max(max(a, |b|), |c|): one v_max3_f32)
            if constexpr (DL == 12) {
                v2f o0 = pair[m] * v2f{a.fir4[0][0], a.fir4[0][0]};
                v2f o1 = pair[m] * v2f{a.fir4[0][1], a.fir4[0][1]};
                v2f o2 = pair[m] * v2f{a.fir4[0][2], a.fir4[0][2]};
#pragma unroll
                for (int i = 1; i < 12; ++i) {
                    const v2f p0 = pair[m + i] * v2f{a.fir4[i][0], a.fir4[i][0]};
                    const v2f p1 = pair[m + i] * v2f{a.fir4[i][1], a.fir4[i][1]};
                    const v2f p2 = pair[m + i] * v2f{a.fir4[i][2], a.fir4[i][2]};
                    o0 = o0 + p0;
                    o1 = o1 + p1;
                    o2 = o2 + p2;
                }
                peak = fmaxf(fmaxf(peak, fabsf(o0.x)), fabsf(o0.y));
                peak = fmaxf(fmaxf(peak, fabsf(o1.x)), fabsf(o1.y));
                peak = fmaxf(fmaxf(peak, fabsf(o2.x)), fabsf(o2.y));
            } else if constexpr (DL == 24) {
                v2f o = pair[m] * v2f{a.fir2[0], a.fir2[0]};
#pragma unroll
                for (int i = 1; i < 24; ++i) o = o + pair[m + i] * v2f{a.fir2[i], a.fir2[i]};
                peak = fmaxf(fmaxf(peak, fabsf(o.x)), fabsf(o.y));
            }
        }
#pragma unroll
        for (int i = 0; i < H; ++i) hist[i] = ext[i];
    }
    static __device__ __forceinline__ void store(const LoudChunkArgs& a, uint32_t s, uint32_t ch, uint32_t c, float peak) {
        omx_loudness_snapshot* snap = a.snapshots + (uint64_t)s * a.n_blocks + c;
        snap->true_peak_db[ch] = power_to_db(peak * peak, a.floor_db);
        if (ch == 0)
            for (uint32_t i = a.channels; i < OMX_MAX_CHANNELS; ++i) snap->true_peak_db[i] = a.floor_db;  // with_floor (:197-207)
    }
};

}  // namespace

// ---- K-weighting: PASS 0 = zero-state end state of the block; PASS 1 = from the true start state: samples -> ring,
// sub-block sums.  grid (slot groups, blocks), 64 threads: lane = slot of the group.
// PK >= 0 (PASS 1 only): the block's true peak is computed here as well, on the tile this pass has in registers, with delay length PK (0,
// 12 or 24).  Pass B is bound by its memory traffic (1.18 GB per cfg3 call at 5.3 TB/s, VALU 40 % busy) and the interpolator is pure
// VALU work on data already in registers: inside this kernel the two overlap, where in pass A the interpolator was a VALU-bound kernel
// of its own (0.23 ms) in front of a memory-bound one (0.23 ms).
template <int PASS, bool TILED, bool TAILS, bool RAGGED, int PK = -1>
__global__ __launch_bounds__(64) void loud_chunk_filter_kernel(LoudChunkArgs a) {
    __shared__ float tile[TILED ? 2 : 1][TILED ? 64 * 17 : 1];
    if (PASS == 1 && *a.bad != 0u) return;
    const uint32_t lane = threadIdx.x, group = blockIdx.x, c = blockIdx.y;
    const uint32_t C = a.channels, L = a.block_frames, steps = L / STEP;
    const uint32_t chan = group * 64u + lane;
    SlotCall sc = slot_call<false>(a, 0);
    bool live = slot_live(a, chan);
    if constexpr (RAGGED) {
        sc = live ? slot_call(a, chan >> a.slot_shift) : SlotCall{0ull, 0u, false};
        live = c < sc.blocks;
        if (__ballot(live) == 0ull) return;  // (one wavefront per workgroup)
    }
    Tile t;
    Direct dl;
    if constexpr (TILED) t.template setup<RAGGED>(a, group, c, lane);
    else dl.setup(a, chan, live, (uint64_t)c * L);
    const uint32_t rd = (lane / C) * 17u * C + (lane % C);
    double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
    double* cf = a.chunk_filter + ((uint64_t)chan * a.n_blocks + c) * 4u;
    if (PASS == 1 && live) {
        f0 = cf[0];
        f1 = cf[1];
        f2 = cf[2];
        f3 = cf[3];
    }
    double b0 = a.b[0], b1 = a.b[1], b2 = a.b[2], b3 = a.b[3], b4 = a.b[4], a1 = a.a[1], a2 = a.a[2], a3 = a.a[3], a4 = a.a[4];
#ifndef LOUD_COEF_VGPR
#define LOUD_COEF_VGPR 0
#endif
    if constexpr (PK > 0 && LOUD_COEF_VGPR) {  // beside the interpolator's 72 tap SGPRs the nine coefficients go to vector registers (18 fewer scalars to spill)
        asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4));
    }
    RingT* ring_col = a.ring + (uint64_t)group * a.ring_len * kRow + lane;
    const uint32_t ring_len = (uint32_t)a.ring_len;  // (window_length of a 3 s window: < 2^32 at any rate)
    uint32_t pos = (uint32_t)((sc.seen + (uint64_t)c * L) % a.ring_len);  // ring slot of the block's first sample
    // a call longer than the ring: only its newest ring_len samples are stored (an earlier block must not race a later one for a slot).
    // keep_from = the first sample of THIS block that is stored (0 in every call that fits the ring)
    const uint64_t idx0 = (uint64_t)c * L, frames_s = (uint64_t)sc.blocks * L, first_kept = frames_s > a.ring_len ? frames_s - a.ring_len : 0u;
    const uint32_t keep_from = first_kept > idx0 ? (uint32_t)min(first_kept - idx0, (uint64_t)L) : 0u;
    double* sub = a.sub_sums + ((uint64_t)chan * a.n_blocks + c) * (L / SUB);
    // Sub-block sums: 64 non-negative squares, two plain f64 accumulators (even / odd samples).  Plain summation is exact to 64 ulp of the
    // SUB-BLOCK's own sum (7e-15) — what the window sums need: they are differences of double-double running totals of these values, so
    // a loud passage's rounding never reaches a quiet window (until round 5 a Kahan-Babuska-Neumaier pair: 10 instructions per sample
    // for digits below the f32 rounding of the samples themselves).
    double acc0 = 0.0, acc1 = 0.0;
    // window lengths off the 64-sample grid (44.1 / 88.2 kHz): the sum of the LAST cap_w % 64 samples of every sub-block, so that a
    // window starting inside sub-block g is Q[end] - (Q[g] - tail_w[g])
    double tl[kLoudnessWindows] = {0.0, 0.0, 0.0, 0.0};
    const uint64_t g_first = sc.seen / SUB + (uint64_t)c * (L / SUB);
    double* tails = TAILS && live ? a.tails + (uint64_t)chan * kLoudnessWindows * a.q_len : nullptr;
    uint32_t bad = 0;
    using TP = TruePeak<(PK > 0 ? PK : 0)>;
    float hist[TP::H];
    float peak = 0.0f;
    if constexpr (PK >= 0) TP::load_history(hist, a, chan, chan >> a.slot_shift, chan & ((1u << a.slot_shift) - 1u), c, sc.reset, live);
    if constexpr (TILED) t.issue(0, C);
    else dl.issue(0, C);
    for (uint32_t step = 0; step < steps; ++step) {
        float x[STEP];
        if constexpr (TILED) {
            bad |= t.stage(tile[step & 1u]);
            if (step + 1u < steps) t.issue(step + 1u, C);
            __syncthreads();
            const float* row = tile[step & 1u] + rd;
#pragma unroll
            for (int f = 0; f < STEP; ++f) x[f] = row[f * C];
        } else {
            bad |= dl.take(x);
            if (step + 1u < steps) dl.issue(step + 1u, C);
        }
        if constexpr (PK >= 0) {
            float ext[STEP + TP::H];
#pragma unroll
            for (int k = 0; k < STEP; ++k) ext[STEP - 1 - k] = x[k];
            TP::step(ext, hist, peak, a);
            __builtin_amdgcn_sched_barrier(0);  // the interpolator's 70 registers are dead from here: keep the recurrence below out of its live range
        }
        RingT out[STEP];
#pragma unroll
        for (int f = 0; f < STEP; ++f) {  // k_weighted (:153-162)
            // Fused multiply-adds (9 operations per sample where the reference's unfused statement order takes 17): this form is not
            // bit-identical to the sequential order anyway (block-boundary states come from the scan), and an FMA only drops
            // intermediate roundings of a recurrence whose own f64 rounding noise the bars already carry (1e-4 dB; measured 1.5e-5).
            // The sequential kernels keep the reference's order.
            const double xd = (double)x[f];
            const double y = fma(b0, xd, f0);
            f0 = fma(-a1, y, fma(b1, xd, f1));
            f1 = fma(-a2, y, fma(b2, xd, f2));
            f2 = fma(-a3, y, fma(b3, xd, f3));
            f3 = fma(-a4, y, b4 * xd);
            if constexpr (PASS == 1) {
                const float y32 = (float)y;                // rounded to f32 before squaring (:161, :276-277)
                const bool finite = isfinite(y32);         // the square of a finite f32 is a finite f64 (WindowedMeans::push, dsp.rs:325)
                out[f] = finite ? (RingT)y32 : (RingT)0;
                const double filtered = (double)out[f];
                const double value = filtered * filtered;
                if (f & 1) acc1 += value;
                else acc0 += value;
                if constexpr (TAILS) {
                    const int lim = (int)SUB - STEP * (int)(step % (SUB / STEP)) - f;  // samples from this one to the sub-block's end
#pragma unroll
                    for (int w = 0; w < kLoudnessWindows; ++w) tl[w] += (int)a.tail_len[w] >= lim ? value : 0.0;
                }
            }
        }
        if constexpr (PASS == 1) {
            // ring stores of the step: one branch for the common case (every sample kept, no wrap inside the step)
            const uint32_t k0 = step * STEP;
            if (live) {
                if (k0 >= keep_from && pos + (uint32_t)STEP <= ring_len) {
                    RingT* dst = ring_col + (uint64_t)pos * kRow;
#pragma unroll
                    for (int f = 0; f < STEP; ++f) dst[(uint32_t)f * kRow] = out[f];
                } else {
#pragma unroll
                    for (int f = 0; f < STEP; ++f) {
                        const uint32_t pf = pos + (uint32_t)f >= ring_len ? pos + (uint32_t)f - ring_len : pos + (uint32_t)f;
                        if (k0 + (uint32_t)f >= keep_from) ring_col[(uint64_t)pf * kRow] = out[f];
                    }
                }
            }
            pos = pos + (uint32_t)STEP >= ring_len ? pos + (uint32_t)STEP - ring_len : pos + (uint32_t)STEP;
            if ((step + 1u) % (SUB / STEP) == 0u) {
                const uint32_t j = (step + 1u) / (SUB / STEP) - 1u;
                if (live) sub[j] = acc0 + acc1;
                acc0 = acc1 = 0.0;
                if constexpr (TAILS) {
#pragma unroll
                    for (int w = 0; w < kLoudnessWindows; ++w) {
                        if (live) tails[(uint64_t)w * a.q_len + q_slot(g_first + j, a.q_len)] = tl[w];
                        tl[w] = 0.0;
                    }
                }
            }
        }
    }
    if constexpr (PASS == 0) {
        if (__ballot(bad != 0u) != 0ull && lane == 0) atomicOr(a.bad, 1u);
        if (live) {
            cf[0] = f0;
            cf[1] = f1;
            cf[2] = f2;
            cf[3] = f3;
        }
    } else {
        // (the true-peak delay line the next call starts from is written by the snapshot kernel, the last of the call: block 0 of THIS
        // pass still reads the carried one)
        if constexpr (PK >= 0) {
            if (live) TP::store(a, chan >> a.slot_shift, chan & ((1u << a.slot_shift) - 1u), c, peak);
        }
    }
}

// ---- scan of the filter states over the blocks: wavefront = slot, lane = block (see stereometer_chunked.hip for the scheme).
// x_c+1 = T x_c + z_c in double-double arithmetic (T as hi / lo pairs from the host, loudness.cpp: k_weighting_transitions): the
// products T x cancel to ~1e-6 of their size at 192 kHz, and the start states must reach the f64 recurrence's own accuracy.
struct DD {
    double h, l;
};
__device__ __forceinline__ DD dd_madd(DD acc, double th, double tl, DD u) {  // acc + (th + tl) * (u.h + u.l)
    const double p = th * u.h;
    const double pe = fma(th, u.h, -p) + (th * u.l + tl * u.h);
    const double s = acc.h + p, bb = s - acc.h;
    const double e = ((acc.h - (s - bb)) + (p - bb)) + (acc.l + pe);
    const double h = s + e;
    return {h, e - (h - s)};
}
// USE_DD = false (rates up to 96 kHz): the same scan in plain f64 on the high parts — there the products cancel to ~1e-3 of their size
// and the f64 scan is 5e-6 dB or better (measured through the 1e-4 dB bar at 44.1 / 48 / 96 kHz); three times cheaper.
template <bool USE_DD>
__global__ __launch_bounds__(256) void loud_scan_filter_kernel(LoudChunkArgs a, const double* __restrict__ Tp /* [2][6][4][4] */) {
    if (*a.bad != 0u) return;
    const uint32_t chan = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (!slot_live(a, chan)) return;
    const SlotCall sc = slot_call(a, chan >> a.slot_shift);
    LoudnessChannelState& st = a.state[chan];
    if (sc.blocks == 0u) {
        if (sc.reset && lane == 0) st = LoudnessChannelState{};  // a reset without samples: ChannelState::default (:234-236)
        return;
    }
    DD carry[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) carry[k] = {sc.reset ? 0.0 : st.filter[k], 0.0};
    const double* Tl = Tp + 6 * 16;
    for (uint32_t c0 = 0; c0 < sc.blocks; c0 += 64u) {
        const uint32_t c = c0 + lane;
        const bool live = c < sc.blocks;
        double* cf = a.chunk_filter + ((uint64_t)chan * a.n_blocks + (live ? c : c0)) * 4u;
        DD x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = {live ? cf[k] : 0.0, 0.0};
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if constexpr (USE_DD) x[k] = dd_madd(x[k], Tp[k * 4 + m], Tl[k * 4 + m], carry[m]);
                    else x[k].h += Tp[k * 4 + m] * carry[m].h;
                }
            }
        }
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int d = 1 << step;
            const double *Th = Tp + step * 16, *Tw = Tl + step * 16;
            DD up[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) up[k] = {shfl_up_f64(x[k].h, d), USE_DD ? shfl_up_f64(x[k].l, d) : 0.0};
            if ((int)lane >= d) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if constexpr (USE_DD) x[k] = dd_madd(x[k], Th[k * 4 + m], Tw[k * 4 + m], up[m]);
                        else x[k].h += Th[k * 4 + m] * up[m].h;
                    }
                }
            }
        }
        const uint32_t last = min(sc.blocks - c0, 64u) - 1u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double start = shfl_up_f64(x[k].h, 1);
            if (lane == 0) start = carry[k].h;
            if (live) cf[k] = fabs(start) < 1.0e-30 ? 0.0 : start;  // denormal flush once per block (:281-285)
            const double eh = shfl_f64(x[k].h, (int)last), el = USE_DD ? shfl_f64(x[k].l, (int)last) : 0.0;
            carry[k] = fabs(eh) < 1.0e-30 ? DD{0.0, 0.0} : DD{eh, el};
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) st.filter[k] = carry[k].h;
    }
}

// ---- pass A: everything about a block that needs no state — grid (slot groups, blocks), lane = slot.
//   true peak of the block (TruePeakMeter::process, :123-151), bit-identical to the sequential kernels: same samples, same tap order;
//     the DL - 1 samples before the block come from the PCM of the call or, for its first block, from the carried delay line;
//   zero-state end state of the K-weighting filter (k_weighted, :153-162) as four dot products with the host's weights
//     W[k] = A^(L-1-k) B (loudness.cpp: k_weighting_zero_state_weights) — until round 5 a K-weighting pass of its own over the PCM
//     (a 9-FMA dependent chain per sample, 0.15 of the call's 0.69 ms at cfg3); here four independent FMAs per sample on the tile the
//     interpolator has in registers anyway;
//   the non-finite flag (a call with a NaN / Inf sample is redone by the sequential kernels).
// The carried delay line is written by pass B (which runs only when the flag stayed clear).
template <int DL, bool TILED, bool RAGGED, bool PEAK>
__global__ __launch_bounds__(64) void loud_chunk_peak_kernel(LoudChunkArgs a) {
    __shared__ float tile[TILED ? 2 : 1][TILED ? 64 * 17 : 1];
    const uint32_t lane = threadIdx.x, group = blockIdx.x, c = blockIdx.y;
    const uint32_t C = a.channels, L = a.block_frames, steps = L / STEP;
    const uint32_t chan = group * 64u + lane;
    const uint32_t s = chan >> a.slot_shift, ch = chan & ((1u << a.slot_shift) - 1u);
    SlotCall sc = slot_call<false>(a, 0);
    bool live = slot_live(a, chan);
    if constexpr (RAGGED) {
        sc = live ? slot_call(a, s) : SlotCall{0ull, 0u, false};
        live = c < sc.blocks;
        if (__ballot(live) == 0ull) return;
    }
    Tile t;
    Direct dl;
    if constexpr (TILED) t.template setup<RAGGED>(a, group, c, lane);
    else dl.setup(a, chan, live, (uint64_t)c * L);
    const uint32_t rd = (lane / C) * 17u * C + (lane % C);
    using TP = TruePeak<DL>;
    constexpr int H = TP::H;
    float hist[H];  // hist[0] = newest sample before the block
    if constexpr (PEAK) TP::load_history(hist, a, chan, s, ch, c, sc.reset, live);
    float peak = 0.0f;
    double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;  // zero-state end state of the block
    uint32_t bad = 0;
    // Two tiles in flight per wavefront (round 6): with one, a wavefront asks for tile k + 1 when tile k has landed and been staged, i.e. it
    // keeps 4 KiB in flight — 18 wavefronts per CU, 18 MB on the chip, ~3.4 TB/s at the loaded latency (160 us for the 537 MB of cfg3's
    // PCM under the profiler).  Steps are taken in pairs: even steps come from `pre`, odd ones from `pre2`.
    auto body = [&](uint32_t step, auto second) {
        constexpr bool SECOND = decltype(second)::value;
        float ext[STEP + H];  // ext[STEP - 1 - k] = x[k]; ext[STEP + i] = hist[i]
        if constexpr (TILED) {
            bad |= t.template stage<SECOND>(tile[step & 1u]);
            if (step + 2u < steps) {
                if constexpr (SECOND) t.issue2(step + 2u, C);
                else t.issue(step + 2u, C);
            }
            __syncthreads();
            const float* row = tile[step & 1u] + rd;
#pragma unroll
            for (int k = 0; k < STEP; ++k) ext[STEP - 1 - k] = row[k * C];  // (no select: Tile::stage zeroes the rows of dead streams, and a
                                                                             //  dead lane's results are never written)
        } else {
            float x[STEP];
            bad |= dl.take(x);
            if (step + 1u < steps) dl.issue(step + 1u, C);
#pragma unroll
            for (int k = 0; k < STEP; ++k) ext[STEP - 1 - k] = x[k];
        }
        {   // zero-state end state: z += W[k] x[k] (wave-uniform weights: scalar loads)
            const double* w = a.zs_weights + (uint64_t)step * (STEP * 4);
#pragma unroll
            for (int k = 0; k < STEP; ++k) {
                const double xd = (double)ext[STEP - 1 - k];
                z0 = fma(w[k * 4 + 0], xd, z0);
                z1 = fma(w[k * 4 + 1], xd, z1);
                z2 = fma(w[k * 4 + 2], xd, z2);
                z3 = fma(w[k * 4 + 3], xd, z3);
            }
        }
        if constexpr (PEAK) TP::step(ext, hist, peak, a);
    };
    if constexpr (TILED) {
        t.issue(0, C);
        if (steps > 1u) t.issue2(1, C);
    } else {
        dl.issue(0, C);
    }
    for (uint32_t step = 0; step < steps; step += 2u) {  // (steps = block_frames / 16 is even: block_frames is a multiple of 64)
        body(step, std::false_type{});
        if (step + 1u < steps) body(step + 1u, std::true_type{});
    }
    if (__ballot(bad != 0u) != 0ull && lane == 0) atomicOr(a.bad, 1u);
    if (!live) return;
    if constexpr (PEAK) TP::store(a, s, ch, c, peak);
    double* cf = a.chunk_filter + ((uint64_t)chan * a.n_blocks + c) * 4u;
    cf[0] = z0;
    cf[1] = z1;
    cf[2] = z2;
    cf[3] = z3;
}

// ---- prefix of the sub-block sums: Q[g] = sum of every squared sample up to the end of sub-block g since the last reset.
// wavefront = slot, lane = sub-block (64 per sweep)
__global__ __launch_bounds__(256) void loud_scan_q_kernel(LoudChunkArgs a) {
    if (*a.bad != 0u) return;
    const uint32_t chan = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (!slot_live(a, chan)) return;
    const SlotCall sc = slot_call(a, chan >> a.slot_shift);
    const uint64_t n_sub = (uint64_t)sc.blocks * (a.block_frames / SUB), g0 = sc.seen / SUB;  // first new sub-block
    double* q = a.q_ring + (uint64_t)chan * a.q_len;
    double* ql = a.q_lo + (uint64_t)chan * a.q_len;
    // the total so far as a double-double pair; the prefix of a sweep's 64 sub-block sums (<= 4096) is plain f64
    double carry = g0 == 0 ? 0.0 : q[q_slot(g0 - 1u, a.q_len)], carry_lo = g0 == 0 ? 0.0 : ql[q_slot(g0 - 1u, a.q_len)];
    const double* sub = a.sub_sums + (uint64_t)chan * a.n_blocks * (a.block_frames / SUB);
    for (uint64_t j0 = 0; j0 < n_sub; j0 += 64u) {
        const uint64_t j = j0 + lane;
        double x = j < n_sub ? sub[j] : 0.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double up = shfl_up_f64(x, d);
            if ((int)lane >= d) x += up;
        }
        const double hi = carry + x, bb = hi - carry;                 // two-sum (Knuth): hi + err == carry + x exactly
        const double lo = ((carry - (hi - bb)) + (x - bb)) + carry_lo;
        if (j < n_sub) {
            q[q_slot(g0 + j, a.q_len)] = hi;
            ql[q_slot(g0 + j, a.q_len)] = lo;
        }
        carry = shfl_f64(hi, 63);
        carry_lo = shfl_f64(lo, 63);
    }
}

// ---- rebuild Q from the sample ring (after calls that went through the sequential kernels): sub-block sums of the
// newest min(seen, ring_len) samples of every slot, then the same prefix.  grid (slot groups, sub-blocks), lane = slot.
constexpr uint32_t kRebuildRows = 16;  // sub-blocks per workgroup of loud_rebuild_sub_kernel
struct RebuildSpan {
    uint64_t first_sub, n;  // first whole sub-block in the ring, number of whole sub-blocks
    uint32_t avail;         // samples the ring still holds of sub-block first_sub - 1 (a ring length off the 64-sample grid)
};
__device__ __forceinline__ RebuildSpan rebuild_span(const LoudChunkArgs& a, uint32_t s) {
    const uint64_t seen = slot_call(a, s).seen;  // (0 for a stream about to be reset: nothing to rebuild)
    const uint64_t have = min(seen, a.ring_len), oldest = seen - have;
    const uint64_t first_sub = (oldest + SUB - 1u) / SUB, total = seen / SUB;
    return {first_sub, total > first_sub ? total - first_sub : 0u, (uint32_t)(first_sub * SUB - oldest)};
}
__global__ __launch_bounds__(64) void loud_rebuild_sub_kernel(LoudChunkArgs a, double* out /* [chan][stride] */, uint64_t stride, const uint32_t* only_if) {
    if (only_if && *only_if == 0u) return;  // (the launch after every chunk-parallel call: kRebuildRows rows per workgroup keep it cheap)
    const uint32_t lane = threadIdx.x, group = blockIdx.x;
    const uint32_t chan = group * 64u + lane;
    if (!slot_live(a, chan)) return;
    const RebuildSpan sp = rebuild_span(a, chan >> a.slot_shift);
    const RingT* ring_col = a.ring + (uint64_t)group * a.ring_len * kRow + lane;
    for (uint64_t j = (uint64_t)blockIdx.y * kRebuildRows; j < ((uint64_t)blockIdx.y + 1u) * kRebuildRows; ++j) {
        // j == n: the sub-block before the first whole one; only its tails are needed (the longest window starts inside it)
        const bool partial = j == sp.n;
        if (j > sp.n || (partial && (sp.avail == 0u || !a.tails || sp.first_sub == 0u))) break;
        const uint64_t g = partial ? sp.first_sub - 1u : sp.first_sub + j;
        const uint32_t i0 = partial ? SUB - sp.avail : 0u;
        double sum = 0.0, c = 0.0, tl[kLoudnessWindows] = {0.0, 0.0, 0.0, 0.0};
        for (uint32_t i = i0; i < SUB; ++i) {
            const double v = ring_square(ring_col[((g * SUB + i) % a.ring_len) * kRow]);
            kbn(sum, c, v);
#pragma unroll
            for (int w = 0; w < kLoudnessWindows; ++w) tl[w] += a.tail_len[w] >= SUB - i ? v : 0.0;
        }
        if (!partial) out[(uint64_t)chan * stride + j] = sum + c;
        if (a.tails) {
#pragma unroll
            for (int w = 0; w < kLoudnessWindows; ++w) a.tails[((uint64_t)chan * kLoudnessWindows + w) * a.q_len + q_slot(g, a.q_len)] = tl[w];
        }
    }
}
__global__ __launch_bounds__(256) void loud_rebuild_q_kernel(LoudChunkArgs a, const double* sub, uint64_t stride, const uint32_t* only_if) {
    if (only_if && *only_if == 0u) return;
    const uint32_t chan = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (!slot_live(a, chan)) return;
    const RebuildSpan sp = rebuild_span(a, chan >> a.slot_shift);
    double* q = a.q_ring + (uint64_t)chan * a.q_len;
    double* ql = a.q_lo + (uint64_t)chan * a.q_len;
    double carry = 0.0;  // only differences of Q are used once the windows are full; before that first_sub == 0
    for (uint64_t j0 = 0; j0 < sp.n; j0 += 64u) {
        const uint64_t j = j0 + lane;
        double x = j < sp.n ? sub[(uint64_t)chan * stride + j] : 0.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double up = shfl_up_f64(x, d);
            if ((int)lane >= d) x += up;
        }
        x += carry;  // (at most ring_len samples: plain f64 is exact enough here, the low words restart at 0)
        if (j < sp.n) {
            q[q_slot(sp.first_sub + j, a.q_len)] = x;
            ql[q_slot(sp.first_sub + j, a.q_len)] = 0.0;
        }
        carry = shfl_f64(x, 63);
    }
    if (sp.first_sub > 0 && lane == 0) {
        q[q_slot(sp.first_sub - 1u, a.q_len)] = 0.0;
        ql[q_slot(sp.first_sub - 1u, a.q_len)] = 0.0;
    }
}

// ---- ragged calls: the per-stream sample counters move once every kernel above has read them
__global__ __launch_bounds__(256) void loud_chunk_advance_kernel(LoudChunkArgs a) {
    if (*a.bad != 0u) return;
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= a.n_streams) return;
    const SlotCall sc = slot_call(a, s);
    if (sc.blocks != 0u || sc.reset) a.seen_v[s] = sc.seen + (uint64_t)sc.blocks * a.block_frames;
}

// ---- snapshots (loudness/processor.rs:287-310) and the write-back of the KBN pairs: thread = (stream, block)
__global__ __launch_bounds__(512) void loud_chunk_snapshot_kernel(LoudChunkArgs a) {
    // workgroup = one stream x 64 consecutive blocks; wavefront = channel slot, lane = block (until round 5: one thread per (stream, block)
    // walking the channels one after the other — 1024 wavefronts, each a chain of 8 x 10 dependent-latency loads: 67 us at cfg3)
    __shared__ double part[OMX_MAX_CHANNELS][64][2];
    if (*a.bad != 0u) return;
    const uint32_t lane = threadIdx.x & 63u, ch = threadIdx.x >> 6;
    const uint32_t s = blockIdx.x, c = blockIdx.y * 64u + lane;
    const uint32_t C = a.channels;
    const SlotCall sc = slot_call(a, s);
    const bool on = c < sc.blocks && c < a.n_blocks;
    const uint64_t P = sc.seen + (uint64_t)(c + 1u) * a.block_frames;  // pushes at the end of this block
    omx_loudness_snapshot* snap = a.snapshots + (uint64_t)s * a.n_blocks + c;
    const bool last = c + 1u == sc.blocks;
    if (on && ch < C) {
        const uint32_t chan = (s << a.slot_shift) + ch;
        const double* q = a.q_ring + (uint64_t)chan * a.q_len;
        const double* ql = a.q_lo + (uint64_t)chan * a.q_len;
        const double q_end = q[q_slot(P / SUB - 1u, a.q_len)], q_end_lo = ql[q_slot(P / SUB - 1u, a.q_len)];
        double mean[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t m = min(P, a.capacities[w]);  // dsp.rs:367-370: min(count, cap), count itself saturates at the ring
            const uint64_t st = P - m, start_sub = st / SUB;  // st % 64 == (64 - tail_len[w]) % 64 once the window is full
            // window sum = difference of two double-double totals: high words first (exact to the difference's own last bit), then the low words
            double base = 0.0, base_lo = 0.0, tail = 0.0;
            if (st % SUB == 0u) {
                if (start_sub != 0) {
                    base = q[q_slot(start_sub - 1u, a.q_len)];
                    base_lo = ql[q_slot(start_sub - 1u, a.q_len)];
                }
            } else {
                base = q[q_slot(start_sub, a.q_len)];
                base_lo = ql[q_slot(start_sub, a.q_len)];
                tail = a.tails[((uint64_t)chan * kLoudnessWindows + w) * a.q_len + q_slot(start_sub, a.q_len)];
            }
            const double W = ((q_end - base) + (q_end_lo - base_lo)) + tail;
            mean[w] = W / (double)max(m, (uint64_t)1);
            if (last) {  // the sequential kernels' state: live pair = the window sum, `since refresh` pair = sum since the last
                         // multiple of cap pushes (CompensatedPair::refresh, dsp.rs:287-289, :363); corrections folded in
                LoudnessChannelState& st = a.state[chan];
                const uint64_t refresh = (P / a.capacities[w]) * a.capacities[w], refresh_sub = refresh / SUB;
                double rbase, rbase_lo = 0.0, radd = 0.0;
                if (refresh % SUB == 0u) {
                    rbase = refresh_sub == 0 ? 0.0 : q[q_slot(refresh_sub - 1u, a.q_len)];
                    rbase_lo = refresh_sub == 0 ? 0.0 : ql[q_slot(refresh_sub - 1u, a.q_len)];
                } else {  // off the grid: the rest of that sub-block is still in the sample ring (refresh > P - cap >= P - ring_len)
                    const RingT* ring_col = a.ring + (uint64_t)(chan / kRow) * a.ring_len * kRow + chan % kRow;
                    double rest = 0.0;
                    for (uint64_t i = refresh; i < (refresh_sub + 1u) * SUB; ++i) rest += ring_square(ring_col[(i % a.ring_len) * kRow]);
                    rbase = q[q_slot(refresh_sub, a.q_len)];
                    radd = rest;
                    rbase_lo = ql[q_slot(refresh_sub, a.q_len)];
                }
                st.sums[w][0] = W;
                st.corrections[w][0] = 0.0;
                st.sums[w][1] = ((q_end - rbase) + (q_end_lo - rbase_lo)) + radd;
                st.corrections[w][1] = 0.0;
            }
        }
        if (last) {
            // the true-peak delay line the next call starts from (TruePeakMeter::delay, :123-133): the newest delay_len - 1 samples of the
            // stream's last block, newest first, straight from the PCM (block_frames >= 64 > delay_len); `peak` is taken at every
            // snapshot (:301).  Written here, in the call's last kernel: the first block of pass A / B reads the carried line.
            LoudnessChannelState& st = a.state[chan];
            const float* p = a.pcm + ((uint64_t)s * a.frames_total + (uint64_t)(c + 1u) * a.block_frames) * C + ch;
            for (uint32_t i = 0; i + 1u < a.delay_len; ++i) st.delay[i] = *(p - (int64_t)(i + 1u) * C);
            st.peak = 0.0f;
        }
        part[ch][lane][0] = mean[0];
        part[ch][lane][1] = mean[1];
        snap->rms_fast_db[ch] = power_to_db((float)mean[2], a.floor_db);
        snap->rms_slow_db[ch] = power_to_db((float)mean[3], a.floor_db);
    }
    __syncthreads();
    if (!on || ch != 0) return;
    double short_term = 0.0, momentary = 0.0;
    for (uint32_t k = 0; k < C; ++k) {  // position-weighted, channel order (:292-296)
        short_term += part[k][lane][0] * a.weights[k];
        momentary += part[k][lane][1] * a.weights[k];
    }
    snap->short_term_loudness = ms_to_lufs(short_term, a.floor_db);
    snap->momentary_loudness = ms_to_lufs(momentary, a.floor_db);
    snap->channel_count = C;
    snap->_pad = 0;
    for (int k = 0; k < OMX_MAX_CHANNELS; ++k) snap->positions[k] = a.positions[k];
    for (uint32_t k = C; k < OMX_MAX_CHANNELS; ++k) {  // LoudnessSnapshot::with_floor (:197-207)
        snap->rms_fast_db[k] = a.floor_db;
        snap->rms_slow_db[k] = a.floor_db;
    }
}

// a.frames_seen / a.seen_v = the sample counters the ring content corresponds to; only_if: run only when *only_if != 0 (after a
// fallback).  scratch: [slots][ring_len / 64 + 1]
void launch_loudness_rebuild_q(const LoudChunkArgs& a, double* scratch, const uint32_t* only_if, hipStream_t stream) {
    const uint64_t stride = a.ring_len / SUB + 1u;
    const uint32_t slots = a.n_streams << a.slot_shift, groups = (slots + 63u) / 64u;
    if (!a.blocks_v && a.frames_seen < SUB) return;
    const uint64_t rows = a.blocks_v ? stride + 1u : std::min<uint64_t>(a.frames_seen / SUB, stride) + 1u;
    hipLaunchKernelGGL(loud_rebuild_sub_kernel, dim3(groups, (uint32_t)((rows + kRebuildRows - 1u) / kRebuildRows)), dim3(64), 0, stream, a, scratch,
                       stride, only_if);
    hipLaunchKernelGGL(loud_rebuild_q_kernel, dim3((slots + 3u) / 4u), dim3(256), 0, stream, a, scratch, stride, only_if);
}

void launch_loudness_chunked(const LoudChunkArgs& a, const double* d_T, hipStream_t stream) {
    const uint32_t slots = a.n_streams << a.slot_shift, groups = (slots + 63u) / 64u;
    const dim3 grid(groups, a.n_blocks);
    const bool tiled = a.channels == 1 || a.channels == 2 || a.channels == 4 || a.channels == 8;  // 64 slots = whole streams
    const bool ragged = a.blocks_v != nullptr;
    auto with_shape = [&](auto kernel_of) {  // kernel_of(tiled, ragged) launches its instantiation
        using T = std::true_type;
        using F = std::false_type;
        if (tiled) {
            if (ragged) kernel_of(T{}, T{});
            else kernel_of(T{}, F{});
        } else {
            if (ragged) kernel_of(F{}, T{});
            else kernel_of(F{}, F{});
        }
    };
    // LOUD_PEAK_IN_B (default 1): the true peak rides pass B (see loud_chunk_filter_kernel); 0 = in pass A (the form until round 5, A/B builds)
#ifndef LOUD_PEAK_IN_B
#define LOUD_PEAK_IN_B 1
#endif
    constexpr bool kPeakInB = LOUD_PEAK_IN_B != 0;
    auto pass_a = [&](auto dl) {
        constexpr int DL = decltype(dl)::value;
        with_shape([&](auto tiled_c, auto ragged_c) {
            constexpr bool TI = decltype(tiled_c)::value, RG = decltype(ragged_c)::value;
            if constexpr (kPeakInB) hipLaunchKernelGGL((loud_chunk_peak_kernel<0, TI, RG, false>), grid, dim3(64), 0, stream, a);
            else hipLaunchKernelGGL((loud_chunk_peak_kernel<DL, TI, RG, true>), grid, dim3(64), 0, stream, a);
        });
    };
    auto pass_b = [&](auto dl) {
        constexpr int PK = kPeakInB ? decltype(dl)::value : -1;
        with_shape([&](auto tiled_c, auto ragged_c) {
            constexpr bool TI = decltype(tiled_c)::value, RG = decltype(ragged_c)::value;
            if (a.tails) hipLaunchKernelGGL((loud_chunk_filter_kernel<1, TI, true, RG, PK>), grid, dim3(64), 0, stream, a);
            else hipLaunchKernelGGL((loud_chunk_filter_kernel<1, TI, false, RG, PK>), grid, dim3(64), 0, stream, a);
        });
    };
    auto by_delay = [&](auto&& f) {
        if (a.delay_len == 12) f(std::integral_constant<int, 12>{});
        else if (a.delay_len == 24) f(std::integral_constant<int, 24>{});
        else f(std::integral_constant<int, 0>{});
    };
    by_delay(pass_a);
    if (a.scan_dd) hipLaunchKernelGGL(loud_scan_filter_kernel<true>, dim3((slots + 3u) / 4u), dim3(256), 0, stream, a, d_T);
    else hipLaunchKernelGGL(loud_scan_filter_kernel<false>, dim3((slots + 3u) / 4u), dim3(256), 0, stream, a, d_T);
    by_delay(pass_b);
    hipLaunchKernelGGL(loud_scan_q_kernel, dim3((slots + 3u) / 4u), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(loud_chunk_snapshot_kernel, dim3(a.n_streams, (a.n_blocks + 63u) / 64u), dim3(64u << a.slot_shift), 0, stream, a);
    if (a.blocks_v) hipLaunchKernelGGL(loud_chunk_advance_kernel, dim3((a.n_streams + 255u) / 256u), dim3(256), 0, stream, a);
}

}  // namespace omx
