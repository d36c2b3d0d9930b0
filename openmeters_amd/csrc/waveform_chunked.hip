// K10c: the waveform processor evaluated chunk-parallel in time (reference src/visuals/waveform/processor.rs:92-121, :213-298,
// src/dsp.rs:298-371, :422-432, :473-495).
//
// The sequential kernels (waveform_roles_kernels.hip, waveform_kernels.hip) walk a stream's frames in order — bit-identical to the
// reference, and latency-bound: one workgroup of 5 wavefronts per four streams, ~300 cycles per frame (2.4 ms for 1024 streams x
// 16 384 frames, 0.8 % of HBM).  Everything on this path except the non-finite resets is linear in its state or a plain sum:
//
//   pass A   per (chunk, stream): the band filters from a ZERO state over the chunk -> zero-state end state (the low and mid bands in f64:
//            stereometer_chunked.hip explains why); non-finite or absurdly large input raises `bad`                    (parallel)
//   scan     per (stream, band, side): true start state of every chunk, s_{c+1} = T s_c + e_c, T = the C-frame zero-input
//            transition of the cascade (host, f64) — wave-parallel over the chunks
//   old      the tracker rings hold the last color_len / slow_len pushed values: their sums between the CUTS of the call (below)
//   pass B   per (chunk, stream): the filters again from the TRUE start state (low / mid: f64) -> band values -> the trackers' inputs
//            (|v| gain, v^2: BandTracker::process :108-121), summed in f64 between cuts; min / max / last sample of the four derived
//            channels between cuts; the newest ring-length values go to the rings; the last chunk leaves the filter states
//   prefix   per (stream, value): a double-double running total over the segment sums
//   columns  per (column, stream): every window mean is the difference of two running totals (WindowedMeans::mean :367-370 is
//            an exact sum: the reference keeps it compensated), min / max is a reduction over the column's segments with the
//            reference's `last_sample` extension (:213-250); the pseudo-column at the end of the call gives the preview and the
//            state the sequential kernels continue from (compensated pairs: sum, 0)
//
// CUTS: a window mean is needed at the column ends only, and the column ends of a lock-step call are known before it starts (the
// fractional phase is host arithmetic, :253-254, :287-291).  The host lists every frame boundary that some sum starts or ends at —
// column ends, column end minus window length for each window, the CompensatedPair refresh points, chunk ends, and a grid over the
// part of the rings that precedes the call — sorted: consecutive cuts delimit SEGMENTS, every sum that is ever needed is a run
// of whole segments, and segment sums are plain f64 additions of non-negative values (relative error n eps, no cancellation).
// The only subtraction — running total at a window's end minus running total at its start — is done in double-double, so a
// quiet window after a loud passage is as exact as the reference's compensated sum.
//
// Not bit-identical to the sequential order: the low and mid bands follow the f64 recurrence (below), the high band restarts every
// chunk from a scanned f32 state; band values differ from the reference's f32 evaluation by what THAT is from exact (~1e-6 of
// the band's level on steady signal, more through a free decay).  Bars: tests/test_gpu_parity_meters.py.
// min / max fields are bit-identical (a reduction of the same samples).  What is NOT linear — Biquad::process' non-finite reset,
// the non-finite rules of the trackers and of the min / max state machine — never runs here: `bad` sends the whole call through
// the sequential kernel (nothing but scratch has been written by then).
#include <mutex>
#include <type_traits>

#include "waveform_device.hpp"

namespace omx {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));  // (left, right)

constexpr int STEP = 16;        // frames per staged tile
constexpr int ROW_FLOATS = 34;  // 16 frames x 2 + 2 pad floats: lanes read their own row with conflict-free ds_read_b64
// frames per ring exchange of pass B: the three band wavefronts of a workgroup hand their values to one another through LDS so that a lane
// can store a 16-byte row piece.  8 (round 4): 24.5 KiB of exchange (49 with RMS history) beside 17 KiB of staged tiles = 3 (2)
// workgroups per CU, and pass B was bound by that occupancy; 4 halves it: 5 (3) workgroups per CU for one more barrier pair per step.
#ifndef WAVE_XF
#define WAVE_XF 4
#endif
constexpr int XF = WAVE_XF;
#ifndef WAVE_DENSE_UNROLL
#define WAVE_DENSE_UNROLL 2
#endif
#ifndef WAVE_DENSE_NT
#define WAVE_DENSE_NT 1
#endif
#ifndef WAVE_ROLE_COPIES
#define WAVE_ROLE_COPIES 1
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ring_store(float* dst, float4 v) {
#if WAVE_DENSE_NT
    __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(dst));
#else
    *reinterpret_cast<float4*>(dst) = v;
#endif
}
constexpr float kAbsurd = 1.0e18f;  // |sample| beyond this could overflow a squared band value: the sequential kernel's business

// Biquad::process (dsp.rs:422-432), L and R at once, the reference's statement order (the build never contracts: -ffp-contract=off)
__device__ __forceinline__ v2f biquad_lr(const BiquadCoef& c, v2f& z0, v2f& z1, v2f x) {
    const v2f out = c.b[0] * x + z0;
    z0 = c.b[1] * x - c.a[0] * out + z1;
    z1 = c.b[2] * x - c.a[1] * out;
    return out;
}
__device__ __forceinline__ void biquad_lr_f64(const WaveChunkArgs::Coef64& c, double (&z0)[2], double (&z1)[2], double (&x)[2]) {
    const double b0 = c.b[0], b1 = c.b[1], b2 = c.b[2], a0 = c.a[0], a1 = c.a[1];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const double out = __builtin_fma(b0, x[ch], z0[ch]);
        z0[ch] = __builtin_fma(-a0, out, __builtin_fma(b1, x[ch], z1[ch]));
        z1[ch] = __builtin_fma(-a1, out, b2 * x[ch]);
        x[ch] = out;
    }
}
__device__ __forceinline__ float flush20(float v) { return fabsf(v) < 1.0e-20f ? 0.0f : v; }  // flush_denormal_f32

}  // namespace

// role 0: low band (LP_low); role 1: mid (HP_low -> LP_high); role 2: high (HP_high) — ThreeBand<Biquad, false> (dsp.rs:473-495)
// workgroup = (chunk, 64 consecutive streams): every lane of a wavefront sees the same cuts
template <bool PASS_B, bool DENSE>
#ifndef WAVE_WPE
#define WAVE_WPE 0
#endif
#ifndef WAVE_XGROUP
#define WAVE_XGROUP 0
#endif
#if WAVE_WPE
#define WAVE_ATTR __attribute__((amdgpu_waves_per_eu(WAVE_WPE, WAVE_WPE)))
#else
#define WAVE_ATTR
#endif
__global__ __launch_bounds__(192) WAVE_ATTR void wave_chunk_kernel(WaveChunkArgs a) {
    static_assert(PASS_B || !DENSE, "pass A writes no rings");
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [2][64][ROW_FLOATS], then (pass B) the ring exchange [2 series][XF][64][12]
    if (PASS_B && *a.bad != 0u) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t groups = (a.n_local + 63u) / 64u;
    const uint32_t c = blockIdx.x / groups, s0 = (blockIdx.x % groups) * 64u;  // s0: first LOCAL stream (a.stream_map: local -> bank index)
    const uint32_t f0 = c * a.chunk_frames;
    const uint32_t n = min(a.chunk_frames, (uint32_t)(a.frames - f0));  // frames of this chunk (even: the host checks)
    const uint32_t steps = (n + STEP - 1u) / STEP;

    // ---- loader: (row, part) pairs of a tile, 16 bytes (two frames) each; 512 pairs over 192 threads
    const float* src[3];
    uint32_t dst[3], part2[3];
    bool live[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t q = tid + (uint32_t)k * 192u;
        const uint32_t row = q >> 3, part = q & 7u;
        const uint32_t sl = s0 + row;
        live[k] = q < 512u && sl < a.n_local;
        const uint32_t s = live[k] ? (a.stream_map ? a.stream_map[sl] : sl) : 0u;
        src[k] = a.pcm + ((uint64_t)s * a.pcm_stride + f0) * 2u + part * 4u;
        dst[k] = row * ROW_FLOATS + part * 4u;
        part2[k] = part * 2u;
    }
    float4 pre[3];
    auto issue = [&](uint32_t step) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            pre[k] = (live[k] && step * STEP + part2[k] < n) ? *reinterpret_cast<const float4*>(src[k] + (uint64_t)step * (STEP * 2))
                                                              : float4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    uint32_t bad = 0;
    auto stage = [&](uint32_t buf) {
        float* t = tile + buf * (64 * ROW_FLOATS);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if ((uint32_t)k * 192u + tid >= 512u) continue;
            const float4 p = pre[k];
            if constexpr (!PASS_B) {  // (|x| <= kAbsurd is false for NaN)
                bad |= (fabsf(p.x) <= kAbsurd && fabsf(p.y) <= kAbsurd && fabsf(p.z) <= kAbsurd && fabsf(p.w) <= kAbsurd) ? 0u : 1u;
            }
            // dsp.rs:232-239: left = (0.0 + s0 * w00) + s1 * w10 (two-channel fold, statement order kept)
            const v2f g0{0.0f + p.x * a.m00 + p.y * a.m10, 0.0f + p.x * a.m01 + p.y * a.m11};
            const v2f g1{0.0f + p.z * a.m00 + p.w * a.m10, 0.0f + p.z * a.m01 + p.w * a.m11};
            *reinterpret_cast<v2f*>(t + dst[k]) = g0;
            *reinterpret_cast<v2f*>(t + dst[k] + 2) = g1;
        }
    };

    // ---- per-lane recurrence state
    const uint32_t sl = s0 + lane;
    const bool mine = sl < a.n_local;
    const uint32_t s = mine ? (a.stream_map ? a.stream_map[sl] : sl) : 0u;
    const BiquadCoef cb = a.hp_hi;
    const WaveChunkArgs::Coef64 ca64 = a.hp_lo64, cb64 = role == 0 ? a.lp_lo64 : a.lp_hi64;
    v2f zb0{0.0f, 0.0f}, zb1{0.0f, 0.0f};  // the high band's section (f32, the reference's precision: well conditioned)
    // Roles 0 and 1 (the bands with a 200 Hz section) run in f64 in BOTH passes: a rounding of a DF2T state of such a section
    // excites its near-double pole (gain fs / (2 pi e 0.707 fc) = 20 at 48 kHz), the reference's own f32 evaluation pays that on
    // every frame and a chunked f32 evaluation pays it differently — after level steps the two sat up to 5e-5 of the band's level
    // apart (soak seeds 21051365, 21056365: above the 2e-5 + 3 |oracle - exact| bar).  In f64 this form follows the exact
    // recurrence (band values rounded to f32 once, where the trackers take them), so its distance to the reference is the
    // reference's own distance to exact.  Cost: none (v_fma_f64 is full rate on CDNA4; two dependent FMAs per frame against
    // four dependent f32 operations of the un-contracted statement order).  The filters' states are rounded to f32 once per
    // call, where the sequential kernels' layout takes them.
    double da0[2] = {0.0, 0.0}, da1[2] = {0.0, 0.0}, d0[2] = {0.0, 0.0}, d1[2] = {0.0, 0.0};
    float* cs = a.chunk_state + (((uint64_t)c * a.n_local + (mine ? sl : 0u)) * 3u + role) * 16u;
    double* cs64 = reinterpret_cast<double*>(cs);
    if (PASS_B && mine) {  // the chunk's true start state ([state k][side]: role 0 LP z0, z1; role 1 A z0, A z1, B z0, B z1 — f64; role 2 f32)
        if (role == 0) {
            d0[0] = cs64[0]; d0[1] = cs64[1]; d1[0] = cs64[2]; d1[1] = cs64[3];
        } else if (role == 1) {
            da0[0] = cs64[0]; da0[1] = cs64[1]; da1[0] = cs64[2]; da1[1] = cs64[3];
            d0[0] = cs64[4]; d0[1] = cs64[5]; d1[0] = cs64[6]; d1[1] = cs64[7];
        } else {
            zb0 = v2f{cs[0], cs[1]};
            zb1 = v2f{cs[2], cs[3]};
        }
    }
    // pass B: sums between cuts, the rings' newest values
    const float gain = role == 0 ? 1.0f : (role == 1 ? 0.7f : 2.0f);  // BAND_COLOR_GAINS (:22)
    const bool history = DENSE || a.history != 0u;  // (DENSE is launched for banks with RMS history only: a constant there)
    double acc_c[4] = {0.0, 0.0, 0.0, 0.0}, acc_p[4] = {0.0, 0.0, 0.0, 0.0};
    float mn[4] = {0.0f, 0.0f, 0.0f, 0.0f}, mx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool fresh = true;  // no frame since the last cut (uniform)
    uint32_t seg = 0;
    int32_t next_cut = 0;
    uint32_t slot_c = 0, slot_h = 0;
    const uint64_t row = (uint64_t)a.n_streams * 16u;
    if constexpr (PASS_B) {
        seg = a.chunk_seg[c];
        next_cut = a.cuts[seg + 1u];
        slot_c = (uint32_t)((a.pushes0 + f0) % a.color_len);
        slot_h = (uint32_t)((a.pushes0 + f0) % a.slow_len);
    }
    const uint64_t ring_c_from = a.frames > a.color_len ? a.frames - a.color_len : 0u;  // first frame whose colour value stays in the ring
    const uint64_t ring_h_from = a.frames > a.slow_len ? a.frames - a.slow_len : 0u;
    const bool chunk_writes_c = PASS_B && (uint64_t)f0 + n > ring_c_from;
    const bool chunk_writes_h = PASS_B && history && (uint64_t)f0 + n > ring_h_from;
    // The rings' rows are [slot][stream x 16 + channel x 3 + band]: a stream's 12 values of one frame are 48 contiguous bytes, but they
    // are computed by three wavefronts (one per band).  They meet in LDS, 8 frames at a time, and wavefront r stores floats 4 r ... 4 r + 3
    // of every row as one 16-byte store (as 12 scattered 4-byte stores per stream and frame the pass was bound by L2 write
    // requests: 1.08 ms with RMS history on, 1024 streams x 16 384 frames).
    float* xbuf = tile + 2 * 64 * ROW_FLOATS;   // [series][frame of the half][stream][12]
    float* cring = a.color_ring + (uint64_t)s * 16u + role * 4u;
    float* hring = a.hist_ring + (uint64_t)s * 16u + role * 4u;
    uint32_t half_slot_c = 0, half_slot_h = 0;

    // The step loop once per band, the band a compile-time constant in each copy (WAVE_ROLE_COPIES): with the wave-uniform `role` tested at
    // run time inside it, every frame paid three scalar branches and the register copies at their merges — a third of the pass's 132
    // VALU instructions per frame and wavefront were moves (static census, round 6).  The copies execute the same barriers in the same order.
    auto run = [&](auto role_c) {
#if WAVE_ROLE_COPIES
        constexpr uint32_t role = decltype(role_c)::value;
#endif
        issue(0);
        for (uint32_t step = 0; step < steps; ++step) {
            stage(step & 1u);
            if (step + 1u < steps) issue(step + 1u);
            __syncthreads();
            const float* rowp = tile + (step & 1u) * (64 * ROW_FLOATS) + lane * ROW_FLOATS;
            v2f x[STEP];
    #pragma unroll
            for (int f = 0; f < STEP; ++f) x[f] = *reinterpret_cast<const v2f*>(rowp + 2 * f);
    #if WAVE_XGROUP
            // keep the reads (and the f32 -> f64 conversions behind them) of a later XF-frame group from being hoisted over an earlier group's
            // recurrence: 16 converted frames live at once cost 64 VGPRs
    #define WAVE_GROUP_FENCE(f) if (((f) & (XF - 1)) == 0) __builtin_amdgcn_sched_barrier(0)
    #else
    #define WAVE_GROUP_FENCE(f)
    #endif
            const uint32_t nf = min((uint32_t)STEP, n - step * STEP);  // uniform
    #pragma unroll
            for (int f = 0; f < STEP; ++f) {
                WAVE_GROUP_FENCE(f);
                if ((uint32_t)f < nf) {
                v2f v = x[f];
                if (role != 2) {  // (wave-uniform) the low and mid bands in f64, both passes (pass A uses the end state only)
                    double xd[2] = {(double)v.x, (double)v.y};
                    if (role == 1) biquad_lr_f64(ca64, da0, da1, xd);
                    biquad_lr_f64(cb64, d0, d1, xd);
                    v = v2f{(float)xd[0], (float)xd[1]};
                } else {
                    v = biquad_lr(cb, zb0, zb1, v);
                }
                {
                if constexpr (PASS_B) {
                    const uint32_t g = f0 + step * STEP + (uint32_t)f;  // frame of the call
                    // bands of Left, Right, Mid, Side (:262-268)
                    const float bv[4] = {v.x, v.y, (v.x + v.y) * 0.5f, (v.x - v.y) * 0.5f};
                    float cv[4], pw[4];
    #pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {  // BandTracker::process (:108-121): a non-finite colour value / power counts as 0.  This call's PCM
                        // is bounded (pass A), but filter state carried over from an earlier call through the sequential kernels may be
                        // finite and huge (|x| up to 3e38 is legal input): its square overflows here
                        cv[ch] = fabsf(bv[ch]) * gain;
                        pw[ch] = bv[ch] * bv[ch];
                        cv[ch] = cv[ch] <= 3.4028235e38f ? cv[ch] : 0.0f;
                        pw[ch] = pw[ch] <= 3.4028235e38f ? pw[ch] : 0.0f;
                        acc_c[ch] += (double)cv[ch];
                        if (history) acc_p[ch] += (double)pw[ch];
                    }
                    if (role == 0) {  // derived_frame (:123-125) and the min / max of ingest_derived (:275-286)
                        const float dv[4] = {x[f].x, x[f].y, (x[f].x + x[f].y) * 0.5f, (x[f].x - x[f].y) * 0.5f};
    #pragma unroll
                        for (int ch = 0; ch < 4; ++ch) {
                            mn[ch] = fresh ? dv[ch] : wf::min_finite(mn[ch], dv[ch]);
                            mx[ch] = fresh ? dv[ch] : wf::max_finite(mx[ch], dv[ch]);
                        }
                    }
                    fresh = false;
                    if ((f & (XF - 1)) == 0) {
                        half_slot_c = slot_c;
                        half_slot_h = slot_h;
                    }
                    if (chunk_writes_c) {
    #pragma unroll
                        for (int ch = 0; ch < 4; ++ch) xbuf[((f & (XF - 1)) * 64 + (int)lane) * 12 + ch * 3 + (int)role] = cv[ch];
                    }
                    if (chunk_writes_h) {
    #pragma unroll
                        for (int ch = 0; ch < 4; ++ch) xbuf[((XF + (f & (XF - 1))) * 64 + (int)lane) * 12 + ch * 3 + (int)role] = pw[ch];
                    }
                    slot_c = slot_c + 1u == a.color_len ? 0u : slot_c + 1u;
                    slot_h = slot_h + 1u == a.slow_len ? 0u : slot_h + 1u;
                    if ((int32_t)g == next_cut) {  // uniform: the segment ends with this frame
                        if (mine) {
                            double* out = a.seg_sum + ((uint64_t)seg * a.n_local + sl) * 24u + role;
    #pragma unroll
                            for (int ch = 0; ch < 4; ++ch) {
                                out[ch * 3] = acc_c[ch];
                                if (history) out[12 + ch * 3] = acc_p[ch];
                            }
                            if (role == 0) {
                                float* mm = a.seg_mm + ((uint64_t)(seg - a.n_old_segs) * a.n_local + sl) * 12u;
                                const float dv[4] = {x[f].x, x[f].y, (x[f].x + x[f].y) * 0.5f, (x[f].x - x[f].y) * 0.5f};
    #pragma unroll
                                for (int ch = 0; ch < 4; ++ch) {
                                    mm[ch * 3] = mn[ch];
                                    mm[ch * 3 + 1] = mx[ch];
                                    mm[ch * 3 + 2] = dv[ch];
                                }
                            }
                        }
    #pragma unroll
                        for (int ch = 0; ch < 4; ++ch) acc_c[ch] = acc_p[ch] = 0.0;
                        fresh = true;
                        ++seg;
                        next_cut = a.cuts[seg + 1u];
                    }
                }
                }
                }
                if constexpr (PASS_B) {
                    // the end of an XF-frame piece (or of the chunk): the three bands' values of its frames go to the rings, one row piece per lane
                    if ((chunk_writes_c || chunk_writes_h) && ((f & (XF - 1)) == XF - 1) && (uint32_t)(f & ~(XF - 1)) < nf) {
                        __syncthreads();
                        const uint32_t g_half = f0 + step * STEP + (uint32_t)(f & ~(XF - 1));
                        if constexpr (DENSE) {
                            // identity stream map: the 64 streams' rows of one frame are 4 KiB in a row.  One store instruction = 16 whole rows
                            // (lane = stream x quarter row, the pad quarter written as zeros): every 128-byte line leaves complete, in one
                            // piece — as quarter rows 64 bytes apart (below) a line was assembled from six partial writes and its pad never
                            // written, which HBM with ECC answers with read-modify-write
    #pragma unroll WAVE_DENSE_UNROLL
                            for (int it = (int)role; it < 4 * XF; it += 3) {  // (frame of the piece, block of 16 streams), dealt round robin to the three wavefronts
                                const int k = it >> 2;
                                const uint32_t st = (uint32_t)(it & 3) * 16u + (lane >> 2), piece = lane & 3u;
                                const uint32_t g = g_half + (uint32_t)k;
                                if ((uint32_t)((f & ~(XF - 1)) + k) >= nf) break;
                                if (s0 + st >= a.n_local) continue;
                                const uint64_t col = (uint64_t)(s0 + st) * 16u + piece * 4u;
                                const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                                if (chunk_writes_c && g >= ring_c_from) {
                                    uint32_t slot = half_slot_c + (uint32_t)k;
                                    slot = slot >= a.color_len ? slot - a.color_len : slot;
                                    const float4 l4 = *reinterpret_cast<const float4*>(xbuf + (k * 64 + (int)st) * 12 + (int)(piece < 3u ? piece : 2u) * 4);  // (read whatever the lane is: a load chosen by a condition becomes a chosen POINTER)
                                    const float4 v4 = piece < 3u ? l4 : zero4;
                                    ring_store(a.color_ring + (uint64_t)slot * row + col, v4);
                                }
                                if (chunk_writes_h && g >= ring_h_from) {
                                    uint32_t slot = half_slot_h + (uint32_t)k;
                                    slot = slot >= a.slow_len ? slot - a.slow_len : slot;
                                    const float4 l4 = *reinterpret_cast<const float4*>(xbuf + ((XF + k) * 64 + (int)st) * 12 + (int)(piece < 3u ? piece : 2u) * 4);
                                    const float4 v4 = piece < 3u ? l4 : zero4;
                                    ring_store(a.hist_ring + (uint64_t)slot * row + col, v4);
                                }
                            }
                        } else
    #pragma unroll
                        for (int k = 0; k < XF; ++k) {
                            const uint32_t g = g_half + (uint32_t)k;
                            if ((uint32_t)((f & ~(XF - 1)) + k) >= nf || !mine) continue;
                            if (chunk_writes_c && g >= ring_c_from) {
                                uint32_t slot = half_slot_c + (uint32_t)k;
                                slot = slot >= a.color_len ? slot - a.color_len : slot;
                                *reinterpret_cast<float4*>(cring + (uint64_t)slot * row) = *reinterpret_cast<const float4*>(xbuf + (k * 64 + (int)lane) * 12 + (int)role * 4);
                            }
                            if (chunk_writes_h && g >= ring_h_from) {
                                uint32_t slot = half_slot_h + (uint32_t)k;
                                slot = slot >= a.slow_len ? slot - a.slow_len : slot;
                                *reinterpret_cast<float4*>(hring + (uint64_t)slot * row) = *reinterpret_cast<const float4*>(xbuf + ((XF + k) * 64 + (int)lane) * 12 + (int)role * 4);
                            }
                        }
                        __syncthreads();
                    }
                }
            }
        }
    };
#if WAVE_ROLE_COPIES
    if (role == 0u) run(std::integral_constant<uint32_t, 0>{});
    else if (role == 1u) run(std::integral_constant<uint32_t, 1>{});
    else run(std::integral_constant<uint32_t, 2>{});
#else
    run(0);
#endif
    if constexpr (!PASS_B) {
        if (__ballot(bad != 0u) != 0ull && lane == 0) atomicOr(a.bad, 1u);
        if (!mine) return;
        if (role == 0) {
            cs64[0] = d0[0]; cs64[1] = d0[1]; cs64[2] = d1[0]; cs64[3] = d1[1];
        } else if (role == 1) {
            cs64[0] = da0[0]; cs64[1] = da0[1]; cs64[2] = da1[0]; cs64[3] = da1[1];
            cs64[4] = d0[0]; cs64[5] = d0[1]; cs64[6] = d1[0]; cs64[7] = d1[1];
        } else {
            cs[0] = zb0.x; cs[1] = zb0.y; cs[2] = zb1.x; cs[3] = zb1.y;
        }
    } else {
        if (!mine || c + 1u != a.n_chunks) return;
        // the filters' state after the call, BandFilter::flush_denormals once per block (:321-323); every channel lane of the
        // band carries its own copy in the sequential kernels' layout
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            WaveLaneState& st = a.state[(uint64_t)s * 16u + (uint32_t)ch * 3u + role];
            if (role == 1) {
                st.za[0][0] = flush20((float)da0[0]); st.za[0][1] = flush20((float)da1[0]);
                st.za[1][0] = flush20((float)da0[1]); st.za[1][1] = flush20((float)da1[1]);
            }
            if (role != 2) {
                st.zb[0][0] = flush20((float)d0[0]); st.zb[0][1] = flush20((float)d1[0]);
                st.zb[1][0] = flush20((float)d0[1]); st.zb[1][1] = flush20((float)d1[1]);
            } else {
                st.zb[0][0] = flush20(zb0.x); st.zb[0][1] = flush20(zb1.x);
                st.zb[1][0] = flush20(zb0.y); st.zb[1][1] = flush20(zb1.y);
            }
        }
    }
}

// ---- scan: wavefront = (stream, role, side); lane = chunk, 64 chunks per sweep.  x_c <- x_c + T^d x_{c-d} for d = 1, 2, 4 ... 32
// (Hillis-Steele over the affine maps s -> T s + e with a common T); the state carried into the sweep enters as T carry added to
// the first lane.  T^(2^k) from the host (f64).
namespace {
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    const int lo = __shfl_up(__double2loint(v), d), hi = __shfl_up(__double2hiint(v), d);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    const int lo = __shfl(__double2loint(v), src), hi = __shfl(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
template <int N, bool F64IO>
__device__ __forceinline__ void scan_wave(const WaveChunkArgs& a, const double* __restrict__ Tp /* [6][4][4] */, uint32_t sl, uint32_t role,
                                          uint32_t side, uint32_t lane) {
    const uint32_t s = a.stream_map ? a.stream_map[sl] : sl;
    const WaveLaneState& st = a.state[(uint64_t)s * 16u + role];  // the band's lane of channel Left
    double carry[N];
    if (role == 1) {
        carry[0] = (double)st.za[side][0];
        carry[1] = (double)st.za[side][1];
        carry[N - 2] = (double)st.zb[side][0];
        carry[N - 1] = (double)st.zb[side][1];
    } else {
        carry[0] = (double)st.zb[side][0];
        carry[1] = (double)st.zb[side][1];
    }
    const uint32_t nb = a.n_chunks;
    for (uint32_t c0 = 0; c0 < nb; c0 += 64u) {
        const uint32_t c = c0 + lane;
        const bool live = c < nb;
        float* cs = a.chunk_state + (((uint64_t)(live ? c : c0) * a.n_local + sl) * 3u + role) * 16u;
        double* cs64 = reinterpret_cast<double*>(cs);
        double x[N];
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] = !live ? 0.0 : F64IO ? cs64[2 * k + side] : (double)cs[2 * k + side];
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                double acc = x[k];
#pragma unroll
                for (int m = 0; m < N; ++m) acc += Tp[k * 4 + m] * carry[m];
                x[k] = acc;
            }
        }
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int d = 1 << step;
            const double* Td = Tp + step * 16;
            double up[N];
#pragma unroll
            for (int k = 0; k < N; ++k) up[k] = shfl_up_f64(x[k], d);
            if ((int)lane >= d) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    double acc = x[k];
#pragma unroll
                    for (int m = 0; m < N; ++m) acc += Td[k * 4 + m] * up[m];
                    x[k] = acc;
                }
            }
        }
        // x = state AFTER chunk c; the start state of chunk c is lane c - 1's (the carry for the first lane)
        const uint32_t last = min(nb - c0, 64u) - 1u;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double start = shfl_up_f64(x[k], 1);
            if (lane == 0) start = carry[k];
            if (live) {
                if constexpr (F64IO) cs64[2 * k + side] = start;
                else cs[2 * k + side] = (float)start;
            }
            const double e = shfl_f64(x[k], (int)last);
            carry[k] = F64IO ? e : (double)(float)e;  // (the filters' states are f32; the low band keeps its f64 trajectory inside a call)
        }
    }
}
}  // namespace

__global__ __launch_bounds__(256) void wave_scan_states_kernel(WaveChunkArgs a, const double* __restrict__ T /* [3][6][4][4] */) {
    if (*a.bad != 0u) return;
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (w >= a.n_local * 6u) return;
    const uint32_t s = w / 6u, role = (w % 6u) >> 1, side = w & 1u;  // s: local stream
    if (role == 0) scan_wave<2, true>(a, T, s, role, side, lane);
    else if (role == 1) scan_wave<4, true>(a, T + 96, s, role, side, lane);
    else scan_wave<2, false>(a, T + 192, s, role, side, lane);
}

// ---- kept totals: a lock-step bank leaves the double-double running total at every cut a LATER call will start a window at
// (the column phase is host arithmetic, so those cuts are known while the frames are still being summed), and a later call takes
// an old segment's sum as the difference of two kept totals instead of reading the segment back from the ring.
struct DD {
    double hi, lo;
};
__device__ __forceinline__ DD dd_add(DD x, DD y) {
    const double s = x.hi + y.hi;
    const double bb = s - x.hi;
    const double e = ((x.hi - (s - bb)) + (y.hi - bb)) + (x.lo + y.lo);
    const double h = s + e;
    return DD{h, e - (h - s)};
}
// an old segment is served from the kept totals when both of its cuts have one and no voided call lies behind cut 0
__device__ __forceinline__ bool old_seg_kept(const WaveChunkArgs& a, uint32_t j) {
    return a.old_slot && a.old_slot[j] != kWaveNoSlot && a.old_slot[j + 1u] != kWaveNoSlot && *a.void_end <= a.first_count;
}

// every old cut has a kept total (the host's flag) and none of them is void: nothing old is summed or scanned at all — the columns
// take a window's start from the kept totals directly (wave_columns_kernel), the scan starts at cut -1
__device__ __forceinline__ bool old_all_kept(const WaveChunkArgs& a) { return a.all_kept != 0u && *a.void_end <= a.first_count; }

// some old cuts have kept totals, some not: the segments between two that have
__global__ __launch_bounds__(256) void wave_old_kept_kernel(WaveChunkArgs a) {
    if (*a.bad != 0u) return;
    const uint64_t total = (uint64_t)a.n_local * 24u;
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t j = blockIdx.y;
    if (t >= total || (!a.history && t % 24u >= 12u) || !old_seg_kept(a, j)) return;
    const double* t0 = a.totals + (uint64_t)a.old_slot[j] * 2u * total;
    const double* t1 = a.totals + (uint64_t)a.old_slot[j + 1u] * 2u * total;
    a.seg_sum[(uint64_t)j * total + t] = (t1[t] - t0[t]) + (t1[total + t] - t0[total + t]);
}

// thread = (kept cut, stream, value): total at the cut = total at the call's start + (running total at the cut - at cut -1).
// A call the sequential kernel had to do (`bad`) leaves the start's total at its cuts — finite, and never used: void_end keeps later
// calls off every difference that would reach behind this call's end.
__global__ __launch_bounds__(256) void wave_keep_totals_kernel(WaveChunkArgs a) {
    const uint64_t total = (uint64_t)a.n_local * 24u;
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t k = blockIdx.y;
    if (t >= total || (!a.history && t % 24u >= 12u)) return;
    DD at{0.0, 0.0};
    if (a.base_slot != kWaveNoSlot) {
        const double* b = a.totals + (uint64_t)a.base_slot * 2u * total;
        at = DD{b[t], b[total + t]};
    }
    if (*a.bad != 0u) {
        if (k == 0u && t == 0u) atomicMax(reinterpret_cast<unsigned long long*>(a.void_end), (unsigned long long)a.end_count);
    } else {
        const uint64_t i = a.keep[2u * k], i0 = a.n_old_segs;
        const DD run{a.prefix_hi[i * total + t], a.prefix_lo[i * total + t]};
        const DD start{-a.prefix_hi[i0 * total + t], -a.prefix_lo[i0 * total + t]};
        at = dd_add(at, dd_add(run, start));
    }
    double* out = a.totals + (uint64_t)a.keep[2u * k + 1u] * 2u * total;
    out[t] = at.hi;
    out[total + t] = at.lo;
}

// ---- old: sums of the rings' contents between the cuts that precede the call.  wavefront = (old segment, four streams x 16 ring lanes)
__global__ __launch_bounds__(256) void wave_old_sums_kernel(WaveChunkArgs a) {
    if (*a.bad != 0u) return;
    const uint32_t j = blockIdx.x;
    if (old_all_kept(a) || old_seg_kept(a, j)) return;
    const uint32_t gid = blockIdx.y * 256u + threadIdx.x;
    const uint32_t sl = gid >> 4, ln = gid & 15u;
    const bool live = sl < a.n_local && ln < 12u;
    const uint32_t s = live ? (a.stream_map ? a.stream_map[sl] : sl) : 0u;
    const uint64_t row = (uint64_t)a.n_streams * 16u;
    const int64_t lo = (int64_t)a.cuts[j] + 1, hi = (int64_t)a.cuts[j + 1u];  // frames lo ..= hi, all negative
    auto sum_ring = [&](const float* ring, uint32_t len) -> double {
        // frame g < 0 sits in slot (pushes0 + g) mod len while g >= -len and the stream has pushed that far back
        const int64_t from = max(lo, max(-(int64_t)len, -(int64_t)min(a.pushes0, (uint64_t)0x7FFFFFFFu)));
        if (from > hi) return 0.0;
        uint32_t slot = (uint32_t)((a.pushes0 + (uint64_t)(from + (int64_t)len)) % len);
        const float* p = ring + (uint64_t)s * 16u + (live ? ln : 0u);
        double acc = 0.0;
        int64_t g = from;
        for (; g + 7 <= hi; g += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = p[(uint64_t)slot * row];
                slot = slot + 1u == len ? 0u : slot + 1u;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += (double)v[k];
        }
        for (; g <= hi; ++g) {
            acc += (double)p[(uint64_t)slot * row];
            slot = slot + 1u == len ? 0u : slot + 1u;
        }
        return acc;
    };
    const double c = sum_ring(a.color_ring, a.color_len);
    const double p = a.history ? sum_ring(a.hist_ring, a.slow_len) : 0.0;
    if (live) {
        double* out = a.seg_sum + ((uint64_t)j * a.n_local + sl) * 24u;
        out[ln] = c;
        if (a.history) out[12u + ln] = p;
    }
}

// ---- prefix: thread = (stream, value); running totals as double-double pairs, prefix[i] = sum of the segments before cut i
__global__ __launch_bounds__(256) void wave_prefix_kernel(WaveChunkArgs a) {
    if (*a.bad != 0u) return;
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x, total = (uint64_t)a.n_local * 24u;
    if (t >= total) return;
    if (!a.history && t % 24u >= 12u) return;
    const uint32_t first = old_all_kept(a) ? a.n_old_segs : 0u;  // (all kept: the running totals start at cut -1)
    double hi = 0.0, lo = 0.0;
    a.prefix_hi[(uint64_t)first * total + t] = 0.0;
    a.prefix_lo[(uint64_t)first * total + t] = 0.0;
    const double* src = a.seg_sum + t;
    // One thread per value, 384 wavefronts in all: less than one per SIMD, so nothing hides a load's latency but the thread itself —
    // two batches of 16 segment sums are kept in flight ahead of the batch being added (as one batch per round trip the kernel spent
    // 67 us on ~38 round trips per thread; the dependent double-double additions themselves are ~27 us).
    auto fetch = [&](double (&x)[16], uint32_t j0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = j0 + (uint32_t)k < a.n_segs ? src[(uint64_t)(j0 + (uint32_t)k) * total] : 0.0;
    };
    auto add = [&](const double (&x)[16], uint32_t j0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (j0 + (uint32_t)k >= a.n_segs) break;
            // (hi, lo) += x: two-sum, then renormalise
            const double sum = hi + x[k];
            const double bb = sum - hi;
            const double err = (hi - (sum - bb)) + (x[k] - bb);
            const double l2 = lo + err;
            const double h2 = sum + l2;
            lo = l2 - (h2 - sum);
            hi = h2;
            a.prefix_hi[(uint64_t)(j0 + (uint32_t)k + 1u) * total + t] = hi;
            a.prefix_lo[(uint64_t)(j0 + (uint32_t)k + 1u) * total + t] = lo;
        }
    };
    double x0[16], x1[16], x2[16];
    fetch(x0, first);
    fetch(x1, first + 16u);
    for (uint32_t j0 = first; j0 < a.n_segs; j0 += 48u) {
        fetch(x2, j0 + 32u);
        add(x0, j0);
        if (j0 + 16u >= a.n_segs) break;
        fetch(x0, j0 + 48u);
        add(x1, j0 + 16u);
        if (j0 + 32u >= a.n_segs) break;
        fetch(x1, j0 + 64u);
        add(x2, j0 + 32u);
    }
}

// ---- columns: thread = (eval, stream, channel); eval = a kept column or the pseudo-column at the end of the call
__global__ __launch_bounds__(256) void wave_columns_kernel(WaveChunkArgs a) {
    if (*a.bad != 0u) return;
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t ch = (uint32_t)(t & 3u);
    const uint64_t rest = t >> 2;
    const uint32_t sl = (uint32_t)(rest % a.n_local);
    const uint64_t e = rest / a.n_local;
    if (e >= a.n_evals) return;
    const uint32_t s = a.stream_map ? a.stream_map[sl] : sl;
    const WaveEval ev = a.evals[e];
    const bool final_eval = ev.out == 0xFFFFFFFFu;
    const uint64_t total = (uint64_t)a.n_local * 24u;
    const bool all_kept = old_all_kept(a);
    auto window_sum = [&](uint32_t series, uint32_t band, uint32_t from_cut) -> double {  // running total at idx_end minus at from_cut
        const uint64_t v = (uint64_t)sl * 24u + series * 12u + ch * 3u + band;
        const double h1 = a.prefix_hi[(uint64_t)ev.idx_end * total + v], l1 = a.prefix_lo[(uint64_t)ev.idx_end * total + v];
        double h0, l0;
        if (all_kept && from_cut < a.n_old_segs) {  // a start before the call: (kept total there) - (kept total at the call's start), <= 0
            const double* t0 = a.totals + (uint64_t)a.old_slot[from_cut] * 2u * total;
            const double* tb = a.totals + (uint64_t)a.old_slot[a.n_old_segs] * 2u * total;
            const DD d = dd_add(DD{t0[v], t0[total + v]}, DD{-tb[v], -tb[total + v]});
            h0 = d.hi;
            l0 = d.lo;
        } else {
            h0 = a.prefix_hi[(uint64_t)from_cut * total + v];
            l0 = a.prefix_lo[(uint64_t)from_cut * total + v];
        }
        return (h1 - h0) + (l1 - l0);
    };
    // ---- min / max (:213-250): the column's samples, extended by the last sample of the column before it
    const uint32_t k_from = ev.mm_from, k_to = ev.mm_to;  // new-segment indices [from, to)
    WaveLaneState& st0 = a.state[(uint64_t)s * 16u + ch * 3u];  // the channel's band-0 lane carries the column state machine
    bool some = false;
    float mn = 0.0f, mx = 0.0f;
    if (ev.carry && st0.cur_some) {
        some = true;
        mn = st0.cur_min;
        mx = st0.cur_max;
    }
    for (uint32_t k = k_from; k < k_to; ++k) {
        const float* mm = a.seg_mm + ((uint64_t)k * a.n_local + sl) * 12u + ch * 3u;
        mn = some ? wf::min_finite(mn, mm[0]) : mm[0];
        mx = some ? wf::max_finite(mx, mm[1]) : mm[1];
        some = true;
    }
    bool ext_valid;
    float ext = 0.0f;
    if (ev.carry) {
        ext_valid = st0.last_valid != 0u;
        ext = st0.last_sample;
    } else {
        ext_valid = true;
        ext = a.seg_mm[((uint64_t)(k_from - 1u) * a.n_local + sl) * 12u + ch * 3u + 2u];
    }
    float cmin = 0.0f, cmax = 0.0f;
    if (some) {
        cmin = mn;
        cmax = mx;
        if (ext_valid) {
            cmin = fminf(cmin, ext);
            cmax = fmaxf(cmax, ext);
        }
    }
    omx_wave_column col;
    col.min = cmin;
    col.max = cmax;
#pragma unroll
    for (uint32_t band = 0; band < 3; ++band) {
        const double m = fmax(window_sum(0u, band, ev.idx_start[0]) / (double)ev.count[0], 0.0);
        col.color_bands[band] = (float)m;
        if (a.history) {
            const double m0 = fmax(window_sum(1u, band, ev.idx_start[1]) / (double)ev.count[1], 0.0);
            const double m1 = fmax(window_sum(1u, band, ev.idx_start[2]) / (double)ev.count[2], 0.0);
            col.rms_db[0][band] = wf::power_to_db_f((float)m0, -140.0f);
            col.rms_db[1][band] = wf::power_to_db_f((float)m1, -140.0f);
        } else {
            col.rms_db[0][band] = -140.0f;
            col.rms_db[1][band] = -140.0f;
        }
    }
    if (!final_eval) {
        a.columns[((uint64_t)s * a.col_stride + ev.out) * 4u + ch] = col;
        return;
    }
    if (a.write_preview) a.preview[(uint64_t)s * 4u + ch] = col;
    // ---- the state the sequential kernels continue from
    // (every thread of the call read the column state of its own (stream, channel) above; only this one writes it)
    const bool partial = k_to > k_from;  // frames since the last column end of this call
    float last = 0.0f;
    if (partial) last = a.seg_mm[((uint64_t)(k_to - 1u) * a.n_local + sl) * 12u + ch * 3u + 2u];
    if (ev.carry) {  // no column ended in this call: the open column grew, last_sample / last_valid stay
        st0.cur_some = some ? 1u : 0u;
        st0.cur_min = mn;
        st0.cur_max = mx;
        st0.cur_last = last;       // (carry and no frames cannot happen: an empty call never gets here)
        st0.cur_has_last = 1u;
    } else {
        st0.cur_some = partial ? 1u : 0u;
        st0.cur_min = partial ? mn : 0.0f;
        st0.cur_max = partial ? mx : 0.0f;
        st0.cur_last = partial ? last : 0.0f;
        st0.cur_has_last = partial ? 1u : 0u;
        st0.last_valid = 1u;
        st0.last_sample = ext;  // the last sample of the newest finished column
    }
    // CompensatedPairs (dsp.rs:264-296): (window sum, 0) and (sum since the last refresh, 0)
#pragma unroll
    for (uint32_t band = 0; band < 3; ++band) {
        WaveLaneState& st = a.state[(uint64_t)s * 16u + ch * 3u + band];
        st.color[0] = window_sum(0u, band, ev.idx_start[0]);
        st.color[1] = window_sum(0u, band, ev.idx_refresh[0]);
        st.color[2] = st.color[3] = 0.0;
        for (uint32_t w = 0; w < 2u; ++w) {
            st.hist[w][0] = a.history ? window_sum(1u, band, ev.idx_start[1u + w]) : 0.0;
            st.hist[w][1] = a.history ? window_sum(1u, band, ev.idx_refresh[1u + w]) : 0.0;
            st.hist[w][2] = st.hist[w][3] = 0.0;
        }
    }
}

__global__ __launch_bounds__(256) void wave_mirror_copy_kernel(const uint8_t* __restrict__ src, uint32_t n, uint64_t* pushes_v, double* phase_v,
                                                               uint32_t* cols_v, float* progress_v, const uint32_t* bad) {
    if (*bad != 0u) return;
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n) return;
    pushes_v[s] = reinterpret_cast<const uint64_t*>(src)[s];
    phase_v[s] = reinterpret_cast<const double*>(src + (size_t)n * 8u)[s];
    cols_v[s] = reinterpret_cast<const uint32_t*>(src + (size_t)n * 16u)[s];
    progress_v[s] = reinterpret_cast<const float*>(src + (size_t)n * 20u)[s];
}
__global__ __launch_bounds__(256) void wave_reset_streams_kernel(WaveLaneState* state, const uint32_t* __restrict__ streams, uint32_t n) {
    constexpr uint32_t kWords = sizeof(WaveLaneState) * 16u / 4u;  // 32-bit words of one stream's 16 lane states
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= (uint64_t)n * kWords) return;
    const uint32_t s = streams[t / kWords];
    reinterpret_cast<uint32_t*>(state + (uint64_t)s * 16u)[t % kWords] = 0u;
}
void launch_waveform_reset_streams(WaveLaneState* state, const uint32_t* streams, uint32_t n, hipStream_t stream) {
    if (n == 0) return;
    const uint64_t threads = (uint64_t)n * (sizeof(WaveLaneState) * 16u / 4u);
    hipLaunchKernelGGL(wave_reset_streams_kernel, dim3((uint32_t)((threads + 255u) / 256u)), dim3(256), 0, stream, state, streams, n);
}
void launch_waveform_mirror_copy(const uint8_t* src, uint32_t n, uint64_t* pushes_v, double* phase_v, uint32_t* cols_v, float* progress_v,
                                 const uint32_t* bad, hipStream_t stream) {
    hipLaunchKernelGGL(wave_mirror_copy_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, src, n, pushes_v, phase_v, cols_v, progress_v, bad);
}

// phase 1: everything that only writes scratch and may raise `bad`; phase 2: the rest.  A ragged call runs phase 1 of EVERY group of
// streams before phase 2 of any (the sequential fallback must find the state untouched).
void launch_waveform_chunked_phase1(const WaveChunkArgs& a, const double* d_T, hipStream_t stream) {
    const uint32_t groups = (a.n_local + 63u) / 64u;
    const size_t lds = (size_t)2 * 64 * ROW_FLOATS * sizeof(float);
    hipLaunchKernelGGL((wave_chunk_kernel<false, false>), dim3(groups * a.n_chunks), dim3(192), lds, stream, a);
    hipLaunchKernelGGL(wave_scan_states_kernel, dim3((a.n_local * 6u + 3u) / 4u), dim3(256), 0, stream, a, d_T);
}
void launch_waveform_chunked_phase2(const WaveChunkArgs& a, hipStream_t stream) {
    const uint32_t groups = (a.n_local + 63u) / 64u;
    const size_t lds = (size_t)2 * 64 * ROW_FLOATS * sizeof(float);
    const uint32_t value_groups = (uint32_t)(((uint64_t)a.n_local * 24u + 255u) / 256u);
    if (a.n_old_segs) hipLaunchKernelGGL(wave_old_sums_kernel, dim3(a.n_old_segs, (a.n_local * 16u + 255u) / 256u), dim3(256), 0, stream, a);
    if (a.n_old_segs && a.old_slot && !a.all_kept) hipLaunchKernelGGL(wave_old_kept_kernel, dim3(value_groups, a.n_old_segs), dim3(256), 0, stream, a);
    const size_t lds_b = lds + (size_t)(a.history ? 2 : 1) * XF * 64 * 12 * sizeof(float);  // + the ring exchange
    static std::once_flag attr_once;  // (two host threads may race on the first launch; one device per process, omx.h)
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wave_chunk_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wave_chunk_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    });
    // identity stream map and the RMS-history ring (every frame of the call leaves a 64-byte row per stream): whole ring rows per store
    // instruction.  Without the history ring only the last color_len frames leave rows and the plain form's register allocation is the
    // better one (0.48 against 0.54 ms per 1024 x 16 384 call)
    if (a.stream_map || !a.history) hipLaunchKernelGGL((wave_chunk_kernel<true, false>), dim3(groups * a.n_chunks), dim3(192), lds_b, stream, a);
    else hipLaunchKernelGGL((wave_chunk_kernel<true, true>), dim3(groups * a.n_chunks), dim3(192), lds_b, stream, a);
    hipLaunchKernelGGL(wave_prefix_kernel, dim3(value_groups), dim3(256), 0, stream, a);
    if (a.n_keep) hipLaunchKernelGGL(wave_keep_totals_kernel, dim3(value_groups, a.n_keep), dim3(256), 0, stream, a);
    // the columns first, then — behind them in the stream — the pseudo-column: it overwrites the column state the first column reads
    WaveChunkArgs cols = a, tail = a;
    cols.n_evals = a.n_evals - 1u;
    tail.evals = a.evals + (a.n_evals - 1u);
    tail.n_evals = 1u;
    if (cols.n_evals) {
        const uint64_t threads = (uint64_t)cols.n_evals * a.n_local * 4u;
        hipLaunchKernelGGL(wave_columns_kernel, dim3((uint32_t)((threads + 255u) / 256u)), dim3(256), 0, stream, cols);
    }
    hipLaunchKernelGGL(wave_columns_kernel, dim3((uint32_t)(((uint64_t)a.n_local * 4u + 255u) / 256u)), dim3(256), 0, stream, tail);
}

}  // namespace omx
