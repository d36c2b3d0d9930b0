// K5 threeband_correlator: per stereo frame 16 f32 biquads (LR4 three-band split of L and R) and four
// f64 EMA correlators.  reference src/dsp.rs:422-432, :489-495 and
// src/visuals/stereometer/processor.rs:40-61, :115-140.
// Four lanes per stream (one per wavefront of the workgroup) run the same code with per-band coefficient sets:
//   band 0  full band      (no filter)
//   band 1  low   = LP_low(x)
//   band 2  mid   = LP_high(HP_low(x))
//   band 3  high  = HP_high(HP_low(x))      (CASCADE_HIGH = true)
// HP_low is evaluated by bands 2 and 3 on identical inputs, so both see bit-identical `above_low`.
#include <type_traits>

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

typedef float v2f __attribute__((ext_vector_type(2)));  // (left, right): both channels of a cascade element in one packed op

// Biquad::process (dsp.rs:422-432) for the L and R filters of one cascade element at once; each component follows the
// reference's statement order (no fused multiply-add), the non-finite reset is per channel.
__device__ __forceinline__ v2f biquad_step2(const BiquadCoef& c, v2f& z0, v2f& z1, v2f x) {
    const v2f out = c.b[0] * x + z0;
    z0 = c.b[1] * x - c.a[0] * out + z1;
    z1 = c.b[2] * x - c.a[1] * out;
    const bool okl = isfinite(out.x), okr = isfinite(out.y);
    if (__builtin_expect(__ballot(!(okl && okr)) != 0ull, 0)) {  // rare: keep the selects off the straight path
        z0 = v2f{okl ? z0.x : 0.0f, okr ? z0.y : 0.0f};
        z1 = v2f{okl ? z1.x : 0.0f, okr ? z1.y : 0.0f};
        return v2f{okl ? out.x : 0.0f, okr ? out.y : 0.0f};
    }
    return out;
}

struct StereoRegs {  // StereoLaneState with the two channels of every delay element paired
    v2f z0[2][2], z1[2][2];  // [stage A/B][cascade element]
    double m[3];
};
__device__ __forceinline__ StereoRegs load_regs(const StereoLaneState& st) {
    StereoRegs r;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            r.z0[g][e] = v2f{st.z[g][e][0][0], st.z[g][e][1][0]};
            r.z1[g][e] = v2f{st.z[g][e][0][1], st.z[g][e][1][1]};
        }
    r.m[0] = st.moments[0];
    r.m[1] = st.moments[1];
    r.m[2] = st.moments[2];
    return r;
}
__device__ __forceinline__ void store_regs(const StereoRegs& r, StereoLaneState& st) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            st.z[g][e][0][0] = r.z0[g][e].x;
            st.z[g][e][1][0] = r.z0[g][e].y;
            st.z[g][e][0][1] = r.z1[g][e].x;
            st.z[g][e][1][1] = r.z1[g][e].y;
        }
    st.moments[0] = r.m[0];
    st.moments[1] = r.m[1];
    st.moments[2] = r.m[2];
}

// One frame of one (stream, band): stereo fold already done.
__device__ __forceinline__ v2f stereo_frame(const BiquadCoef& ca, const BiquadCoef& cb, bool use_a, bool use_b, StereoRegs& st,
                                            double alpha, v2f x) {
    if (use_a) {  // Cascade<Biquad,2> per channel (dsp.rs:447-451)
        x = biquad_step2(ca, st.z0[0][0], st.z1[0][0], x);
        x = biquad_step2(ca, st.z0[0][1], st.z1[0][1], x);
    }
    if (use_b) {
        x = biquad_step2(cb, st.z0[1][0], st.z1[1][0], x);
        x = biquad_step2(cb, st.z0[1][1], st.z1[1][1], x);
    }
    const double ld = (double)x.x, rd = (double)x.y;  // Correlator::update (:40-46)
    st.m[0] += alpha * (ld * rd - st.m[0]);
    st.m[1] += alpha * (ld * ld - st.m[1]);
    st.m[2] += alpha * (rd * rd - st.m[2]);
    return x;
}

// ---- branch-free batch path ---------------------------------------------------------------------------------------------
// The per-frame non-finite test of Biquad::process (dsp.rs:428-431) costs a wave-level branch per cascade element, and the
// branches keep the scheduler from overlapping the elements, the correlator and the history store.  A non-finite output is
// sticky (it poisons z0 / z1 of that element for good), so the fast path runs BATCH frames without the test while accumulating
// `poison += out * 0` per element (NaN as soon as any output was inf / NaN); a clean poison means the batch was exact, otherwise
// the lane state is rolled back and the batch replayed through the per-frame reference path.  (A stage-skewed software
// pipeline of the same batch — element k on frame i - k — was slower: 3.99 vs 2.84 ms; 16-byte loads / stores changed nothing.)
// A dependent v_pk_*_f32 issues every 8 cycles on gfx950, an independent one every 4 (tools/microbench/issue_rate.hip), and the
// wavefront issues in order: the file is built with -amdgpu-sched-strategy=max-ilp so that the four cascade elements, the
// correlator and the neighbouring frames of a batch are interleaved instead of laid out chain after chain.
__device__ __forceinline__ v2f biquad_core2(const BiquadCoef& c, v2f& z0, v2f& z1, v2f x, v2f& poison) {
    const v2f out = c.b[0] * x + z0;
    z0 = c.b[1] * x - c.a[0] * out + z1;
    z1 = c.b[2] * x - c.a[1] * out;
    poison = __builtin_elementwise_fma(out, v2f{0.0f, 0.0f}, poison);
    return out;
}

// CH = 2: compile-time fold (dsp.rs:234-239 has the same specialisation); CH = 0: any channel count.
// Workgroup = 4 wavefronts, wavefront w = band w of 64 consecutive streams (lane = stream): which filters run, whether
// the history is written and whether the band is active at all are wave-uniform, so the per-frame code has no divergent
// branches (the first layout — 4 lanes per stream — spent half of its instructions on exec-mask bookkeeping).
template <int CH>
__global__ __launch_bounds__(256) void stereometer_kernel(StereometerArgs a) {
    const uint32_t band = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, in an SGPR
    const uint32_t s = blockIdx.x * 64 + (threadIdx.x & 63);
    if (s >= a.n_streams) return;
    if (a.run_if && *a.run_if == 0u) return;  // fallback launch of the chunk-parallel path: nothing non-finite was seen
    const uint32_t gid = s * 4 + band;  // state / history / output slot of this (stream, band)
    const bool active = band == 0 || a.analyze_bands != 0;
    // ragged banks: the stream's own block count, ring start and reset flag (lane = stream: per-lane values, the block loop below
    // then has a per-lane trip count)
    const bool ragged = a.blocks_v != nullptr;
    const uint32_t n_blocks_s = ragged ? a.blocks_v[s] : a.n_blocks;
    const uint32_t block_frames_s = a.frames_v != nullptr ? a.frames_v[s] : a.block_frames;  // chunk calls: the stream's own block length
    StereoLaneState st_mem = (a.state_in ? a.state_in : a.state)[gid];
    if (ragged && a.reset_v != nullptr && a.reset_v[s] != 0) memset(&st_mem, 0, sizeof(st_mem));  // reset_audio (:92-97) of this stream
    StereoRegs st = load_regs(st_mem);
    const BiquadCoef ca = a.stage_a[band], cb = a.stage_b[band];
    const bool use_a = a.use_a[band] != 0, use_b = a.use_b[band] != 0;
    const bool push_history = band == 0 || a.emit_band_points != 0;
    const uint32_t channels = CH ? (uint32_t)CH : a.fmt.channels;
    const float* pcm = a.pcm + (uint64_t)s * a.frames_total * channels;
    float* hist = a.history + ((uint64_t)s * 4 + band) * a.hist_frames * 2;
    uint32_t slot = (uint32_t)((ragged ? a.start_v[gid] : a.hist_pos[band]) % a.hist_frames);  // ring slot advanced with a 32-bit compare, not a 64-bit modulo
    const double alpha = a.alpha;
    constexpr int BATCH = 8;  // frames whose loads are issued together

    float2 xnext[BATCH];      // CH == 2 fast path: the prefetched batch
    bool have_next = false;   // xnext holds the batch the loop is about to process
    // History pairs of a batch are stored at the START of the next batch, ahead of its prefetch loads: loads and stores share
    // vmcnt on gfx950, so the wait for the prefetched frames at the top of a batch also waits for every store still in flight —
    // stores issued at the end of the previous batch exposed a full write round trip per 8 frames (the whole kernel time).
    v2f pend[BATCH];
    uint32_t pend_slot = 0;
    bool has_pend = false;
    auto flush_pending = [&]() {
        if (has_pend) {
            uint32_t sl = pend_slot;
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                *reinterpret_cast<v2f*>(hist + 2u * sl) = pend[i];
                sl = sl + 1u == a.hist_frames ? 0u : sl + 1u;
            }
            has_pend = false;
        }
    };
    const bool contiguous_batches = block_frames_s % BATCH == 0;
    for (uint32_t blk = 0; blk < n_blocks_s; ++blk) {
        if (active) {
            const float* base = pcm + (uint64_t)blk * block_frames_s * channels;
            uint32_t f = 0;
            if constexpr (CH == 2) {
                auto batches = [&](auto use_a_c, auto use_b_c, auto push_c) {
                    constexpr bool UA = decltype(use_a_c)::value, UB = decltype(use_b_c)::value, PUSH = decltype(push_c)::value;
                    for (; f + BATCH <= block_frames_s; f += BATCH) {
                        float2 x[BATCH];
                        if (!have_next) {
#pragma unroll
                            for (int i = 0; i < BATCH; ++i) xnext[i] = *reinterpret_cast<const float2*>(base + 2u * (f + i));
                        }
#pragma unroll
                        for (int i = 0; i < BATCH; ++i) x[i] = xnext[i];
                        flush_pending();
                        // the next batch's frames are requested before this batch is computed (one wavefront per SIMD: nothing
                        // else hides the round trip).  Blocks are contiguous, so the prefetch runs across block boundaries; past
                        // the end of the call it re-reads the current batch (unconditional load, selected address).
                        const uint64_t g = (uint64_t)blk * block_frames_s + f + BATCH;
                        have_next = contiguous_batches && g + BATCH <= a.frames_total;
                        const float* nxt = pcm + 2u * (have_next ? g : g - BATCH);
#pragma unroll
                        for (int i = 0; i < BATCH; ++i) xnext[i] = *reinterpret_cast<const float2*>(nxt + 2 * i);
                        const StereoRegs saved = st;
                        const uint32_t slot0 = slot;
                        v2f poison[4] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};
#pragma unroll
                        for (int i = 0; i < BATCH; ++i) {
                            // two channels: the fold is the identity on bits (dsp.rs:232-236) with the default matrix, else weights
                            const float left = 0.0f + x[i].x * a.fmt.m[0][0] + x[i].y * a.fmt.m[1][0];
                            const float right = 0.0f + x[i].x * a.fmt.m[0][1] + x[i].y * a.fmt.m[1][1];
                            v2f y{left, right};
                            if constexpr (UA) {  // Cascade<Biquad,2> per channel (dsp.rs:447-451)
                                y = biquad_core2(ca, st.z0[0][0], st.z1[0][0], y, poison[0]);
                                y = biquad_core2(ca, st.z0[0][1], st.z1[0][1], y, poison[1]);
                            }
                            if constexpr (UB) {
                                y = biquad_core2(cb, st.z0[1][0], st.z1[1][0], y, poison[2]);
                                y = biquad_core2(cb, st.z0[1][1], st.z1[1][1], y, poison[3]);
                            }
                            const double ld = (double)y.x, rd = (double)y.y;  // Correlator::update (:40-46)
                            st.m[0] += alpha * (ld * rd - st.m[0]);
                            st.m[1] += alpha * (ld * ld - st.m[1]);
                            st.m[2] += alpha * (rd * rd - st.m[2]);
                            if constexpr (PUSH) {
                                pend[i] = y;
                                slot = slot + 1u == a.hist_frames ? 0u : slot + 1u;
                            }
                        }
                        pend_slot = slot0;
                        has_pend = PUSH;
                        const v2f taint2 = (poison[0] + poison[1]) + (poison[2] + poison[3]);
                        const float taint = taint2.x + taint2.y;
                        if (__builtin_expect(__ballot(!(taint == 0.0f)) != 0ull, 0)) {  // some output was inf / NaN: exact replay
                            st = saved;
                            slot = slot0;
                            has_pend = false;  // the replay stores its pairs itself
#pragma unroll 1
                            for (int i = 0; i < BATCH; ++i) {  // frames re-read from memory: a dynamically indexed x[] would live in scratch
                                const float2 xi = *reinterpret_cast<const float2*>(base + 2u * (f + (uint32_t)i));
                                const float left = 0.0f + xi.x * a.fmt.m[0][0] + xi.y * a.fmt.m[1][0];
                                const float right = 0.0f + xi.x * a.fmt.m[0][1] + xi.y * a.fmt.m[1][1];
                                const v2f y = stereo_frame(ca, cb, UA, UB, st, alpha, v2f{left, right});
                                if constexpr (PUSH) {
                                    *reinterpret_cast<v2f*>(hist + 2u * slot) = y;
                                    slot = slot + 1u == a.hist_frames ? 0u : slot + 1u;
                                }
                            }
                        }
                    }
                };
                using T = std::true_type;
                using F = std::false_type;
                // wave-uniform role: full band (no filter), low (stage A), mid / high (A then B)
                if (use_a && use_b) push_history ? batches(T{}, T{}, T{}) : batches(T{}, T{}, F{});
                else if (use_a) push_history ? batches(T{}, F{}, T{}) : batches(T{}, F{}, F{});
                else if (!use_b) push_history ? batches(F{}, F{}, T{}) : batches(F{}, F{}, F{});
            }
            if (f < block_frames_s) flush_pending();
            for (; f < block_frames_s; ++f) {
                const float* frame = base + (uint64_t)f * channels;
                float left = 0.0f, right = 0.0f;  // dsp.rs:223-249 stereo fold
                for (uint32_t c = 0; c < channels; ++c) {
                    const float v = frame[c];
                    left = left + v * a.fmt.m[c][0];
                    right = right + v * a.fmt.m[c][1];
                }
                const v2f y = stereo_frame(ca, cb, use_a, use_b, st, alpha, v2f{left, right});
                if (push_history) {
                    *reinterpret_cast<v2f*>(hist + 2u * slot) = y;
                    slot = slot + 1u == a.hist_frames ? 0u : slot + 1u;
                }
            }
            // flush_denormals once per block (:134-140)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (fabs(st.m[i]) < 1.0e-30) st.m[i] = 0.0;
            if (band != 0) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        v2f& p = st.z0[g][e];
                        v2f& q = st.z1[g][e];
                        p = v2f{fabsf(p.x) < 1.0e-20f ? 0.0f : p.x, fabsf(p.y) < 1.0e-20f ? 0.0f : p.y};
                        q = v2f{fabsf(q.x) < 1.0e-20f ? 0.0f : q.x, fabsf(q.y) < 1.0e-20f ? 0.0f : q.y};
                    }
            }
        }
        float value = 0.0f;  // Correlator::value (:48-56)
        if (active) {
            const double denom = sqrt(st.m[1] * st.m[2]);
            if (denom > 1e-12) {
                const double v = st.m[0] / denom;
                if (isfinite(v)) value = (float)fmin(fmax(v, -1.0), 1.0);
            }
        }
        a.correlations[((uint64_t)s * a.n_blocks + blk) * 4 + band] = value;
    }
    flush_pending();
    store_regs(st, st_mem);
    a.state[gid] = st_mem;
}

// ---- role-pipelined form (2 channels, band analysis on, whole rounds per block) ---------------------------------------------
// The four-wavefront form above gives the mid and the high band FOUR cascade elements each — both evaluate HP_low on identical
// inputs — so two SIMDs issue ~62 VALU per frame while the full-band wavefront issues 22.  Here a fifth wavefront (it shares
// SIMD 0 with the light full-band role: wavefronts w and w + 4 of a workgroup sit on one SIMD) runs HP_low once and hands
// `above_low` to the mid / high wavefronts through LDS, one round of ROUND frames ahead of them.  Every filter and correlator
// still sees the same samples in the same order: results are bit-identical to the four-wavefront form.
constexpr int kStereoRound = 32;

template <int ROLE, bool PUSH>  // ROLE 0 full band, 1 low, 2 mid, 3 high, 4 HP_low producer
__device__ __forceinline__ void stereo_role(const StereometerArgs& a, v2f (*abuf)[kStereoRound][64], v2f (*xbuf)[kStereoRound][64],
                                            uint32_t s, bool live_lane, uint32_t lane) {
    constexpr int BATCH = 8, R = kStereoRound;
    constexpr uint32_t band = ROLE == 4 ? 2u : (uint32_t)ROLE;
    // only the producer reads the PCM: it publishes the folded frame (for the full-band and low roles) next to `above_low`
    constexpr bool FILTERS = ROLE != 0, FROM_LDS = ROLE != 4;
    const uint64_t total = (uint64_t)a.n_blocks * a.block_frames, rounds = total / R;
    const float* pcm = a.pcm + (uint64_t)s * a.frames_total * 2u;
    // the two cascade elements of this role: stage A of the band's slot (low: LP_low; producer: HP_low), stage B (mid / high)
    constexpr int G = (ROLE == 2 || ROLE == 3) ? 1 : 0;
    const BiquadCoef c = G ? a.stage_b[band] : a.stage_a[band];
    StereoLaneState* slot = a.state + (uint64_t)s * 4 + band;
    v2f z0[2] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}}, z1[2] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};
    double m[3] = {0.0, 0.0, 0.0};
    if (FILTERS) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            z0[e] = v2f{slot->z[G][e][0][0], slot->z[G][e][1][0]};
            z1[e] = v2f{slot->z[G][e][0][1], slot->z[G][e][1][1]};
        }
    }
    if (ROLE != 4) {
        m[0] = slot->moments[0];
        m[1] = slot->moments[1];
        m[2] = slot->moments[2];
    }
    float* hist = a.history + ((uint64_t)s * 4 + band) * a.hist_frames * 2;
    uint32_t hslot = (uint32_t)(a.hist_pos[band] % a.hist_frames);
    const double alpha = a.alpha;
    float2 xnext[BATCH];
    bool have_next = false;
    v2f pend[BATCH];
    uint32_t pend_slot = 0, in_block = 0, blk = 0;
    bool has_pend = false;
    auto flush_pending = [&]() {  // see stereometer_kernel: stores go out ahead of the next prefetch
        if (has_pend) {
            uint32_t sl = pend_slot;
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                *reinterpret_cast<v2f*>(hist + 2u * sl) = pend[i];
                sl = sl + 1u == a.hist_frames ? 0u : sl + 1u;
            }
            has_pend = false;
        }
    };
    auto fold = [&](float2 x) {
        const float left = 0.0f + x.x * a.fmt.m[0][0] + x.y * a.fmt.m[1][0];
        const float right = 0.0f + x.x * a.fmt.m[0][1] + x.y * a.fmt.m[1][1];
        return v2f{left, right};
    };
    for (uint64_t i = 0; i <= rounds; ++i) {
        const bool active = ROLE == 4 ? i < rounds : i >= 1;
        if (active) {
            const uint64_t round = ROLE == 4 ? i : i - 1;
            v2f (*ab)[64] = (ROLE == 0 || ROLE == 1) ? xbuf[round & 1] : abuf[round & 1];  // this role's input (producer: its output)
            v2f (*xb)[64] = xbuf[round & 1];
#pragma unroll 1
            for (int sb = 0; sb < R / BATCH; ++sb) {
                const uint64_t g0 = round * R + (uint64_t)sb * BATCH;  // first frame of the batch, counted from the start of the call
                v2f in[BATCH];
                if constexpr (FROM_LDS) {
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) in[k] = ab[sb * BATCH + k][lane];
                    if constexpr (PUSH) flush_pending();
                } else {
                    if (!have_next) {
#pragma unroll
                        for (int k = 0; k < BATCH; ++k) xnext[k] = *reinterpret_cast<const float2*>(pcm + 2u * (g0 + k));
                    }
                    float2 x[BATCH];
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) x[k] = xnext[k];
                    if constexpr (PUSH) flush_pending();
                    have_next = g0 + 2 * BATCH <= total;
                    const float* nxt = pcm + 2u * (have_next ? g0 + BATCH : g0);
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) xnext[k] = *reinterpret_cast<const float2*>(nxt + 2 * k);
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) in[k] = fold(x[k]);
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) xb[sb * BATCH + k][lane] = in[k];
                }
                const v2f sz0[2] = {z0[0], z0[1]}, sz1[2] = {z1[0], z1[1]};
                const double sm[3] = {m[0], m[1], m[2]};
                const uint32_t hslot0 = hslot;
                v2f poison[2] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};
                v2f y[BATCH];
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    v2f v = in[k];
                    if constexpr (FILTERS) {
                        v = biquad_core2(c, z0[0], z1[0], v, poison[0]);
                        v = biquad_core2(c, z0[1], z1[1], v, poison[1]);
                    }
                    if constexpr (ROLE != 4) {
                        const double ld = (double)v.x, rd = (double)v.y;  // Correlator::update (:40-46)
                        m[0] += alpha * (ld * rd - m[0]);
                        m[1] += alpha * (ld * ld - m[1]);
                        m[2] += alpha * (rd * rd - m[2]);
                    }
                    y[k] = v;
                    if constexpr (PUSH) hslot = hslot + 1u == a.hist_frames ? 0u : hslot + 1u;
                }
                if constexpr (FILTERS) {
                    const v2f t2 = poison[0] + poison[1];
                    const float taint = t2.x + t2.y;
                    if (__builtin_expect(__ballot(!(taint == 0.0f)) != 0ull, 0)) {  // some output was inf / NaN: exact replay
                        z0[0] = sz0[0], z0[1] = sz0[1], z1[0] = sz1[0], z1[1] = sz1[1];
                        m[0] = sm[0], m[1] = sm[1], m[2] = sm[2];
#pragma unroll 1
                        for (int k = 0; k < BATCH; ++k) {
                            v2f v;
                            if constexpr (FROM_LDS) v = ab[sb * BATCH + k][lane];
                            else v = xb[sb * BATCH + k][lane];  // the folded frame this wavefront published above
                            v = biquad_step2(c, z0[0], z1[0], v);
                            v = biquad_step2(c, z0[1], z1[1], v);
                            if constexpr (ROLE != 4) {
                                const double ld = (double)v.x, rd = (double)v.y;
                                m[0] += alpha * (ld * rd - m[0]);
                                m[1] += alpha * (ld * ld - m[1]);
                                m[2] += alpha * (rd * rd - m[2]);
                            }
                            // (a dynamically indexed y[] would live in scratch: the replayed outputs go out directly)
                            if constexpr (ROLE == 4) ab[sb * BATCH + k][lane] = v;
                            if constexpr (PUSH) {
                                uint32_t sl = hslot0 + (uint32_t)k;
                                sl = sl >= a.hist_frames ? sl - a.hist_frames : sl;
                                if (live_lane) *reinterpret_cast<v2f*>(hist + 2u * sl) = v;
                            }
                        }
                        // lanes that were clean recomputed the same values; skip the batch's normal outputs
                        if constexpr (PUSH) has_pend = false;
                        goto batch_done;
                    }
                }
                if constexpr (ROLE == 4) {
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) ab[sb * BATCH + k][lane] = y[k];
                }
                if constexpr (PUSH) {
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) pend[k] = y[k];
                    pend_slot = hslot0;
                    has_pend = live_lane;
                }
            batch_done:
                in_block += BATCH;  // (a 64-bit `% block_frames` per batch would cost more than the batch's filters)
                if (in_block == a.block_frames) {  // end of a block: flush_denormals (:134-140) and Correlator::value (:48-56)
                    if constexpr (FILTERS) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            z0[e] = v2f{fabsf(z0[e].x) < 1.0e-20f ? 0.0f : z0[e].x, fabsf(z0[e].y) < 1.0e-20f ? 0.0f : z0[e].y};
                            z1[e] = v2f{fabsf(z1[e].x) < 1.0e-20f ? 0.0f : z1[e].x, fabsf(z1[e].y) < 1.0e-20f ? 0.0f : z1[e].y};
                        }
                    }
                    if constexpr (ROLE != 4) {
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            if (fabs(m[q]) < 1.0e-30) m[q] = 0.0;
                        float value = 0.0f;
                        const double denom = sqrt(m[1] * m[2]);
                        if (denom > 1e-12) {
                            const double v = m[0] / denom;
                            if (isfinite(v)) value = (float)fmin(fmax(v, -1.0), 1.0);
                        }
                        if (live_lane) a.correlations[((uint64_t)s * a.n_blocks + blk) * 4 + band] = value;
                    }
                    in_block = 0;
                    ++blk;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if constexpr (PUSH) flush_pending();
    if (!live_lane) return;
    if constexpr (FILTERS) {
        const int n_slots = ROLE == 4 ? 2 : 1;  // the producer's HP_low state is stage A of BOTH the mid and the high slot
        for (int q = 0; q < n_slots; ++q) {
            StereoLaneState* dst = a.state + (uint64_t)s * 4 + band + q;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                dst->z[G][e][0][0] = z0[e].x;
                dst->z[G][e][1][0] = z0[e].y;
                dst->z[G][e][0][1] = z1[e].x;
                dst->z[G][e][1][1] = z1[e].y;
            }
        }
    }
    if constexpr (ROLE != 4) {
        slot->moments[0] = m[0];
        slot->moments[1] = m[1];
        slot->moments[2] = m[2];
    }
}

__global__ __launch_bounds__(320) void stereometer_roles_kernel(StereometerArgs a) {
    __builtin_amdgcn_s_setprio(3);  // see loudness_roles_kernel: latency-bound wavefronts win the issue arbitration on a shared SIMD
    extern __shared__ __attribute__((aligned(16))) unsigned char stereo_smem[];
    v2f (*abuf)[kStereoRound][64] = reinterpret_cast<v2f (*)[kStereoRound][64]>(stereo_smem);  // [2]: above_low
    v2f (*xbuf)[kStereoRound][64] = abuf + 2;                                                   // [2]: folded frames
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t s_raw = blockIdx.x * 64 + lane;
    const bool live_lane = s_raw < a.n_streams;
    const uint32_t s = live_lane ? s_raw : a.n_streams - 1;  // idle lanes shadow the last stream (loads stay in bounds; no stores)
    const bool push = a.emit_band_points != 0;
    switch (wave) {
        case 0: stereo_role<0, true>(a, abuf, xbuf, s, live_lane, lane); break;
        case 1: push ? stereo_role<1, true>(a, abuf, xbuf, s, live_lane, lane) : stereo_role<1, false>(a, abuf, xbuf, s, live_lane, lane); break;
        case 2: push ? stereo_role<2, true>(a, abuf, xbuf, s, live_lane, lane) : stereo_role<2, false>(a, abuf, xbuf, s, live_lane, lane); break;
        case 3: push ? stereo_role<3, true>(a, abuf, xbuf, s, live_lane, lane) : stereo_role<3, false>(a, abuf, xbuf, s, live_lane, lane); break;
        default: stereo_role<4, false>(a, abuf, xbuf, s, live_lane, lane); break;
    }
}

void launch_stereometer(const StereometerArgs& a, hipStream_t stream) {
    if (a.n_streams == 0 || a.n_blocks == 0) return;
    const uint32_t groups = (a.n_streams + 63) / 64;
    static const bool no_roles = [] { const char* e = tuning_env("OMX_STEREO_ROLES"); return e && atoi(e) == 0; }();
    if (a.fmt.channels == 2 && a.analyze_bands && a.block_frames % kStereoRound == 0 && !no_roles && !a.run_if && !a.blocks_v)
    {
        const size_t lds = (size_t)4 * kStereoRound * 64 * sizeof(v2f);  // 64 KiB
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stereometer_roles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
            attr_set = true;
        }
        hipLaunchKernelGGL(stereometer_roles_kernel, dim3(groups), dim3(320), lds, stream, a);
    }
    else if (a.fmt.channels == 2)
        hipLaunchKernelGGL(stereometer_kernel<2>, dim3(groups), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(stereometer_kernel<0>, dim3(groups), dim3(256), 0, stream, a);
}

struct PointsArgs {
    const float* history;
    uint32_t n_streams, hist_frames, target;
    uint64_t hist_pos[4];
    uint32_t band_valid[4];
    float* points;
    const uint64_t* pos_v;         // ragged banks: [n_streams][4] positions and validity per stream (else nullptr)
    const uint32_t* band_valid_v;
};
__global__ __launch_bounds__(256) void stereometer_points_kernel(PointsArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t band = blockIdx.y, s = blockIdx.z;
    const bool valid = a.band_valid_v ? a.band_valid_v[s * 4 + band] != 0 : a.band_valid[band] != 0;
    if (i >= a.target || !valid) return;
    // data = the newest hist_frames pairs, oldest first; pick data[i * frames / target] (:163-168)
    const uint64_t frames = a.hist_frames;
    const uint64_t idx = (uint64_t)i * frames / a.target;
    const uint64_t oldest = (a.pos_v ? a.pos_v[s * 4 + band] : a.hist_pos[band]) - frames;
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

void launch_stereometer_points_ragged(const float* history, uint32_t n_streams, uint32_t hist_frames, const uint64_t* pos_v,
                                      const uint32_t* band_valid_v, uint32_t target, float* points, hipStream_t stream) {
    if (n_streams == 0 || target == 0) return;
    PointsArgs a{};
    a.history = history;
    a.n_streams = n_streams;
    a.hist_frames = hist_frames;
    a.target = target;
    a.points = points;
    a.pos_v = pos_v;
    a.band_valid_v = band_valid_v;
    hipLaunchKernelGGL(stereometer_points_kernel, dim3((target + 255) / 256, 4, n_streams), dim3(256), 0, stream, a);
}

__global__ __launch_bounds__(64) void stereometer_ragged_plan_kernel(StereoPlanArgs a) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= a.n_streams) return;
    const bool reset = a.reset != nullptr && a.reset[s] != 0;
    const uint32_t nb = a.blocks[s];
    const uint64_t block_frames_s = a.frames != nullptr ? a.frames[s] : a.block_frames;
    const uint64_t frames = a.hist_frames, pushed = (uint64_t)nb * block_frames_s;
    uint64_t len0 = (reset || (a.zero_len_mask & 1u)) ? 0ull : a.len[s * 4];
    for (uint32_t blk = 0; blk < a.max_blocks; ++blk) {  // :116, :129, :146-150: produced iff the full-band deque is full
        if (blk < nb) len0 = min(len0 + block_frames_s, frames);
        a.produced[(uint64_t)s * a.max_blocks + blk] = (blk < nb && len0 >= frames) ? 1u : 0u;
    }
    const bool produced_last = nb != 0 && len0 >= frames;
    for (uint32_t b = 0; b < 4; ++b) {
        uint64_t len = (reset || ((a.zero_len_mask >> b) & 1u)) ? 0ull : a.len[s * 4 + b];
        const uint64_t pos = a.pos[s * 4 + b];
        a.start[s * 4 + b] = pos;
        const bool pushes = b == 0 || (a.analyze_bands != 0 && a.emit_band_points != 0);
        if (pushes) {
            a.pos[s * 4 + b] = pos + pushed;
            len = min(len + pushed, frames);
        }
        a.len[s * 4 + b] = len;
        const bool in_play = b == 0 || a.emit_band_points != 0;
        a.band_valid[s * 4 + b] = (produced_last && in_play && len >= frames) ? 1u : 0u;
    }
}
void launch_stereometer_ragged_plan(const StereoPlanArgs& a, hipStream_t stream) {
    if (a.n_streams == 0) return;
    hipLaunchKernelGGL(stereometer_ragged_plan_kernel, dim3((a.n_streams + 63) / 64), dim3(64), 0, stream, a);
}

struct RehomeArgs {
    const float* from;
    float* to;
    uint32_t n_streams, from_frames, to_frames;
    uint64_t hist_pos[4];
    uint64_t keep[4];  // newest pairs carried over per band (<= both ring lengths)
};
// segment length changed (update_config, :183-207 keeps the deques; :142-150 trims them to the new length on the next block):
// the newest keep[band] pairs move to the slots their absolute positions have in the new ring
__global__ __launch_bounds__(256) void stereometer_rehome_kernel(RehomeArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t band = blockIdx.y, s = blockIdx.z;
    if (i >= a.keep[band]) return;
    const uint64_t at = a.hist_pos[band] - a.keep[band] + i;
    const float* src = a.from + (((uint64_t)s * 4 + band) * a.from_frames + at % a.from_frames) * 2;
    float* dst = a.to + (((uint64_t)s * 4 + band) * a.to_frames + at % a.to_frames) * 2;
    dst[0] = src[0];
    dst[1] = src[1];
}

void launch_stereometer_rehome(const float* from, float* to, uint32_t n_streams, uint32_t from_frames, uint32_t to_frames,
                               const uint64_t hist_pos[4], const uint64_t keep[4], hipStream_t stream) {
    uint64_t most = 0;
    RehomeArgs a{};
    a.from = from;
    a.to = to;
    a.n_streams = n_streams;
    a.from_frames = from_frames;
    a.to_frames = to_frames;
    for (int b = 0; b < 4; ++b) {
        a.hist_pos[b] = hist_pos[b];
        a.keep[b] = keep[b];
        most = most > keep[b] ? most : keep[b];
    }
    if (n_streams == 0 || most == 0) return;
    hipLaunchKernelGGL(stereometer_rehome_kernel, dim3((uint32_t)((most + 255) / 256), 4, n_streams), dim3(256), 0, stream, a);
}

// produced[s][blk] = the full-band deque holds `frames` pairs after block blk (:116, :129, :146-150); identical for every stream
__global__ void stereometer_produced_kernel(uint32_t* out, uint32_t n_streams, uint32_t n_blocks, uint64_t len_before,
                                            uint32_t block_frames, uint32_t frames) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (uint64_t)n_streams * n_blocks) return;
    const uint64_t blk = i % n_blocks;
    const uint64_t len = len_before + (blk + 1) * (uint64_t)block_frames;  // the deque is trimmed to `frames`, never below it
    out[i] = len >= frames ? 1u : 0u;
}
void launch_stereometer_produced(uint32_t* out, uint32_t n_streams, uint32_t n_blocks, uint64_t len_before, uint32_t block_frames,
                                 uint32_t frames, hipStream_t stream) {
    const uint64_t n = (uint64_t)n_streams * n_blocks;
    if (n == 0) return;
    hipLaunchKernelGGL(stereometer_produced_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, out, n_streams, n_blocks,
                       len_before, block_frames, frames);
}

}  // namespace omx
