// K6 / K7, wide form — the oscilloscope of every configuration whose autocorrelation is an 8192-point transform (sample rates
// 27.3 ... 54.6 kHz: the reference's 44.1 / 48 kHz).  Three launches per call, whatever its shape (one block of a single-stream
// handle, a lock-step bank call, a ragged bank call with per-stream block counts):
//
//   scope_push_kernel       every frame of the call projected into the trace rings (rings hold history + call)
//   scope_estimate2_kernel  PeriodEstimator::estimate_period (:93-181) for every (stream, block, view) at once — a pure
//                           function of the trace; 256 threads, 47 KiB of LDS, three workgroups per CU
//   scope_trigger_kernel    the stateful part, one workgroup of 512 threads per stream, blocks in timeline order:
//                           StableTrigger::{capture, stabilize, locate, prepare, prepare_template, find_best, write_candidate,
//                           update_reference} (:306-528), zero-crossing capture (:769-786), write_snapshot (:725-750)
//
// What makes the trigger pass fast (round 2: 77 k cycles per block, now see DESIGN §4 K7): the search span, the mean-removed
// copy, the template, the learnt reference and the scores stay in LDS for the whole call — no phase goes to global memory
// except the one load of the block's span and the header store; every reduction is one DPP scan per wavefront + one barrier
// (results ping-pong between two LDS slots, so no trailing barrier); the correlation sweeps of a search round run four offsets
// per wavefront on packed f32 (v_pk_fma_f32: two elements per lane-instruction, the template read once per four offsets);
// the argmax of a round is taken by every wavefront for itself (no broadcast barrier); passes that only prepared the next
// reduction are fused (mean + peak of a span from its sum / max / min: |x - m| is monotone in x under rounding).
//
// Summation order (parity): sums over a span are taken per lane over elements lane, lane + 512, ... (sweeps: groups of four
// consecutive elements, lane + 64 k), then by a DPP prefix scan in lane order, then over the eight wavefronts as a three-level tree.  That is
// neither the reference's sequential order (correlation_stats, :206-208) nor its four stride-4 chains (:210-227): a bit-equal
// order would put a 1920-long dependent f32 chain in front of every search (7 us per block against the 4 us the whole block
// takes here), see DESIGN §2 for what is checked instead (exact-arithmetic third leg, near-tie test).
#include <mutex>
#include "scope_device.hpp"

#include "fft_device.hpp"
#include "fft_pow2_device.hpp"

#include <type_traits>

namespace omx {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
using ScopeTwiddles = TwiddleSource<true, true>;  // pass-2 table in LDS, pass-3 twiddles in VGPRs

// LDS-only workgroup barrier: waits for this wavefront's LDS traffic, not for its global loads / stores / LDS-DMA in flight
// (__syncthreads() fences those too and would drain the span DMA of the trigger pass at its first barrier)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// values every lane holds identically (LDS broadcasts, reduction results) moved to SGPRs: scalar control flow, fewer VGPRs
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float uni(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ uint64_t uni(uint64_t x) { return ((uint64_t)uni((uint32_t)(x >> 32)) << 32) | uni((uint32_t)x); }
__device__ __forceinline__ uint32_t wave_index() { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }  // an SGPR

// ---------------------------------------------------------------- wave / workgroup reductions
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, ROW_MASK, 0xf, false);
}
enum { OP_SUM = 0, OP_MAX = 1, OP_MIN = 2 };
template <int OP>
__device__ __forceinline__ float op_f(float a, float b) {
    return OP == OP_SUM ? a + b : (OP == OP_MAX ? fmaxf(a, b) : fminf(a, b));
}
// inclusive scan in lane order (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31); lane 63 holds the reduction of the wavefront
template <int OP>
__device__ __forceinline__ float wave_scan(float x) {
    x = op_f<OP>(x, dpp_f<0x111, 0xf>(OP == OP_SUM ? 0.0f : x, x));
    x = op_f<OP>(x, dpp_f<0x112, 0xf>(OP == OP_SUM ? 0.0f : x, x));
    x = op_f<OP>(x, dpp_f<0x114, 0xf>(OP == OP_SUM ? 0.0f : x, x));
    x = op_f<OP>(x, dpp_f<0x118, 0xf>(OP == OP_SUM ? 0.0f : x, x));
    x = op_f<OP>(x, dpp_f<0x142, 0xa>(OP == OP_SUM ? 0.0f : x, x));
    x = op_f<OP>(x, dpp_f<0x143, 0xc>(OP == OP_SUM ? 0.0f : x, x));
    return x;
}
template <int OP>
__device__ __forceinline__ float wave_all(float x) {  // the reduction in every lane
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_scan<OP>(x)), 63));
}
template <bool MAX>
__device__ __forceinline__ uint32_t wave_all_u32(uint32_t x) {
    auto op = [](uint32_t a, uint32_t b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
    x = op(x, dpp_u<0x111, 0xf>(x, x));
    x = op(x, dpp_u<0x112, 0xf>(x, x));
    x = op(x, dpp_u<0x114, 0xf>(x, x));
    x = op(x, dpp_u<0x118, 0xf>(x, x));
    x = op(x, dpp_u<0x142, 0xa>(x, x));
    x = op(x, dpp_u<0x143, 0xc>(x, x));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// Cross-lane sums of several per-lane partials at once (gfx950 v_permlane32_swap / v_permlane16_swap; lane mapping checked by
// tools/microbench/permlane_swap.hip): halves, then 16-lane rows, are exchanged between TWO values, so that after two levels one
// register carries the 16-lane partial sums of four values (row r: value {0, 2, 1, 3}[r]) and four row_shr steps finish all four —
// 10 instructions for four values against 24 for four 6-step scans.  Totals land in lane 16 r + 15.
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
// (inline asm: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane32_swap into one register when they are added —
// tools/microbench/permlane_reduce.hip; the s_nop covers the VALU-write -> permlane-read wait states the compiler would insert)
__device__ __forceinline__ float swap_add32(float a, float b) {  // lanes 0-31: a[l] + a[l + 32]; lanes 32-63: b[l - 32] + b[l]
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float swap_add16(float a, float b) {  // rows (a0 + a1, b0 + b1, a2 + a3, b2 + b3)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float row_sum(float x) {  // lane 15 of every 16-lane row: the row's sum
    x += dpp_f<0x111, 0xf>(0.0f, x);
    x += dpp_f<0x112, 0xf>(0.0f, x);
    x += dpp_f<0x114, 0xf>(0.0f, x);
    x += dpp_f<0x118, 0xf>(0.0f, x);
    return x;
}
// totals of v[0..M) in lanes 15 / 31 / 47 / 63 for M = 4 (value order 0, 2, 1, 3), lanes 31 / 63 for M = 2, lane 63 for M = 1
template <int M>
__device__ __forceinline__ float reduce_m(const float (&v)[M]) {
#ifdef OMX_SCOPE_NO_BUTTERFLY
    float out = 0.0f;
    const int lane = threadIdx.x & 63;
    for (int m = 0; m < M; ++m) {
        const float t = wave_all<OP_SUM>(v[m]);
        if (lane == (M == 1 ? 63 : (M == 2 ? (m == 0 ? 31 : 63) : (m == 0 ? 15 : (m == 1 ? 47 : (m == 2 ? 31 : 63)))))) out = t;
    }
    return out;
#endif
    if constexpr (M == 1) {
        return wave_scan<OP_SUM>(v[0]);
    } else if constexpr (M == 2) {
        float z = row_sum(swap_add32(v[0], v[1]));
        z += dpp_f<0x142, 0xa>(0.0f, z);  // row_bcast15: rows 1 / 3 add lane 15 of rows 0 / 2
        return z;
    } else {
        const float z1 = swap_add32(v[0], v[1]);
        const float z2 = swap_add32(v[2], M == 4 ? v[M - 1] : 0.0f);
        return row_sum(swap_add16(z1, z2));
    }
}
// which lane holds value m's total after reduce_m<M>
template <int M>
__device__ __forceinline__ int reduce_lane(int m) {
    if constexpr (M == 1) return 63;
    else if constexpr (M == 2) return m == 0 ? 31 : 63;
    else return m == 0 ? 15 : (m == 1 ? 47 : (m == 2 ? 31 : 63));
}

// K reductions over the workgroup behind ONE barrier: lane 63 of every wavefront publishes its K results into one of two
// alternating slot sets; after the barrier every thread combines the W wavefronts (a tree for W = 8).  A slot set is rewritten two
// reductions later, i.e. after another barrier every thread has passed: no trailing barrier.
template <int W>
struct alignas(16) RedSlots {
    float f[2][8][W];
    uint32_t u[2][W];
};
template <int W>
struct Reducer {
    RedSlots<W>* slots;
    int phase;
    template <int K, int OPS>  // OPS: two bits per component, OP_SUM / OP_MAX / OP_MIN
    __device__ __forceinline__ void run(float (&v)[K]) {
        static_assert(K <= 8, "RedSlots holds eight components");
        const unsigned lane = threadIdx.x & 63u, wave = wave_index();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int op = (OPS >> (2 * k)) & 3;
            v[k] = op == OP_SUM ? wave_scan<OP_SUM>(v[k]) : (op == OP_MAX ? wave_scan<OP_MAX>(v[k]) : wave_scan<OP_MIN>(v[k]));
        }
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < K; ++k) slots->f[phase][k][wave] = v[k];
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int op = (OPS >> (2 * k)) & 3;
            auto comb = [&](float x, float y) { return op == OP_SUM ? x + y : (op == OP_MAX ? fmaxf(x, y) : fminf(x, y)); };
            float r;
            if constexpr (W == 8) {  // two 16-byte reads and a three-level tree instead of eight reads and a seven-long chain
                const v4f lo = *reinterpret_cast<const v4f*>(&slots->f[phase][k][0]), hi = *reinterpret_cast<const v4f*>(&slots->f[phase][k][4]);
                r = comb(comb(comb(lo.x, lo.y), comb(lo.z, lo.w)), comb(comb(hi.x, hi.y), comb(hi.z, hi.w)));
            } else {
                r = slots->f[phase][k][0];
#pragma unroll
                for (int w = 1; w < W; ++w) r = comb(r, slots->f[phase][k][w]);
            }
            v[k] = uni(r);
        }
        phase ^= 1;
    }
    template <bool MAX>
    __device__ __forceinline__ uint32_t run_u32(uint32_t x) {
        const unsigned lane = threadIdx.x & 63u, wave = wave_index();
        x = wave_all_u32<MAX>(x);
        if (lane == 63) slots->u[phase][wave] = x;
        lds_barrier();
        uint32_t r = slots->u[phase][0];
#pragma unroll
        for (int w = 1; w < W; ++w) {
            const uint32_t o = slots->u[phase][w];
            r = MAX ? (o > r ? o : r) : (o < r ? o : r);
        }
        phase ^= 1;
        return uni(r);
    }
};

// four consecutive floats at a 4-byte aligned LDS address (two ds_read2_b32)
struct __attribute__((packed, aligned(4))) U4 {
    v4f v;
};
__device__ __forceinline__ v4f load4u(const float* p) { return reinterpret_cast<const U4*>(p)->v; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// tuning aid (ScopeArgs::phase_timing, OMX_SCOPE_PHASES=1): cycles thread 0 spends between marks, summed over workgroups and blocks
__device__ unsigned long long g_fast_phase_cycles[SCOPE_PHASES];
struct PhaseClock {  // thread 0 accumulates into LDS (no global traffic inside the timed code), flushed once per workgroup
    long long t;
    unsigned long long* acc;  // LDS [SCOPE_PHASES]
    bool on;
    __device__ __forceinline__ void start(bool enabled, unsigned long long* lds_acc) {
        acc = lds_acc;
        on = enabled && threadIdx.x == 0;
        if (on) {
            for (int i = 0; i < SCOPE_PHASES; ++i) acc[i] = 0;
            t = clock64();
        }
    }
    __device__ __forceinline__ void mark(int i) {
        if (on) {
            const long long now = clock64();
            acc[i] += (unsigned long long)(now - t);
            t = now;
        }
    }
    __device__ __forceinline__ void flush() {
        if (on)
            for (int i = 0; i < SCOPE_PHASES; ++i) atomicAdd(&g_fast_phase_cycles[i], acc[i]);
    }
};

// gaussian (:199-204) with exp as v_exp_f32(x * log2 e): absolute error of a weight <= ~3e-8 (relative error |x| 2^-24 on
// weights e^-|x|; the libm expf of the single-pass kernel is ~25 instructions, this is 3), far below the f32 noise of the sums the
// weights enter.  The caller guarantees len > 1 and std > eps.
__device__ __forceinline__ float gaussian_fast(float center, float index, float std_) {
    const float r = (index - center) / std_;
    return __builtin_amdgcn_exp2f((-0.5f * (r * r)) * 1.4426950408889634f);
}

// ---------------------------------------------------------------- scope_trigger_kernel
// StableTrigger's scalar fields in registers for the length of a capture (ScopeTriggerState itself, with its padding array, is
// left in scratch memory by the compiler — and a scratch reload's vmcnt(0) would drain the span DMA)
struct TrigRegs {
    int has_period;
    float period;
    uint32_t missed_periods;
    float reference_period;
    float mean;
    uint32_t ref_len;
};
__device__ __forceinline__ TrigRegs load_trig(const ScopeTriggerState& s) {  // per lane
    return TrigRegs{s.has_period, s.period, s.missed_periods, s.reference_period, s.mean, s.ref_len};
}
__device__ __forceinline__ TrigRegs load_trig_uniform(const ScopeTriggerState& s) {  // the same state in every lane -> SGPRs
    return TrigRegs{uni(s.has_period), uni(s.period), uni(s.missed_periods), uni(s.reference_period), uni(s.mean), uni(s.ref_len)};
}
__device__ __forceinline__ void store_trig(ScopeTriggerState& d, const TrigRegs& t) {
    d.has_period = t.has_period;
    d.period = t.period;
    d.missed_periods = t.missed_periods;
    d.reference_period = t.reference_period;
    d.mean = t.mean;
    d.ref_len = t.ref_len;
}

// LDS of the trigger kernel (floats).  Fixed: the learnt reference.  Per block, sized by the block's own period:
//   raw[span] work[span]     the search span as read from the ring, and minus the tracked mean (span = len + search)
//   tmpl[0..3][len + 8]      the template; copy a is delayed by a samples (zero padded), so that a correlation at an offset
//                            o = 4 q + a reads work and template both at 16-byte aligned addresses (ds_read_b128, 256 B/clk;
//                            two ds_read2_b32 at the 4-byte alignment of an arbitrary offset move half of that).  Copy 0 is
//                            also `candidate`.  Spans too long for four copies (periods beyond ~1500 samples at 48 kHz) keep
//                            copy 0 only and read the work array unaligned.
constexpr int kRoundEntries = 64;  // entries of one search round (coarse <= 40, fine <= 15, see find_best)
struct RoundSums {
    // sum x, sum x^2, sum x y of entry k (scan order) — written by the sweeps, read by every wavefront.  Two sets, alternating by
    // round: a wavefront that is through with a round's argmax starts the next round's sweeps while others still read this one's
    float s[2][kRoundEntries][4];
};
template <int T>
struct Ctx {
    static constexpr int W = T / 64;
    float *ref, *dyn, *raw, *work, *tmpl;  // LDS: the fixed reference, the per-block region and its arrays
    uint32_t tmpl_stride;            // floats between template copies (0: one copy, unaligned sweeps)
    uint32_t dyn_floats;             // floats behind ref[]
    float ref_peak;  // max |ref[]| of the resident reference (kept current by whoever changes ref[])
    RoundSums* sums;
    uint32_t round;  // parity selects the RoundSums set
    Reducer<W> red;
    PhaseClock pc;
    __device__ __forceinline__ bool aligned() const { return tmpl_stride != 0; }
};

// The three sums of normalized_correlation (:210-227) of work[o .. o + len) with the template for M offsets of one alignment
// class a = o & 3, ALIGNED form: groups of four elements j = 4 g .. 4 g + 3 of the delayed template copy a against
// work[(o - a) + j]; the template group is read once for the M offsets.  Interior groups (every element inside the window) run
// unmasked on packed f32 with the next group's reads in flight; the two edge groups are taken element-wise by eight lanes.
template <int M>
__device__ __forceinline__ void sweep_aligned(const float* work, const float* tmpl_a, uint32_t a, uint32_t len, const uint32_t (&off)[M],
                                              float (&r)[3]) {
    const unsigned lane = threadIdx.x & 63u;
    v2f sx[M], sxx[M], sxy[M];
#pragma unroll
    for (int m = 0; m < M; ++m) sx[m] = sxx[m] = sxy[m] = v2f{0.0f, 0.0f};
    const uint32_t groups = (len + a + 3) >> 2;  // groups of the delayed template; 0 and groups - 1 are the edges
    const float* xb[M];
#pragma unroll
    for (int m = 0; m < M; ++m) xb[m] = work + (off[m] - a);
    // Interior groups 1 .. groups - 2, lane l takes 1 + l + 64 t.  The first `full` trips are complete for every lane: a uniform
    // loop, two register sets alternating so that the reads of trip t + 1 are in flight while trip t is accumulated (no copies
    // between the sets: hipcc then counts the waits instead of draining them); the last, partial trip is masked.
    auto accumulate = [&](const v4f& y, const v4f (&x)[M]) {
        const v2f y01{y.x, y.y}, y23{y.z, y.w};
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const v2f x01{x[m].x, x[m].y}, x23{x[m].z, x[m].w};
            sx[m] = (sx[m] + x01) + x23;
            sxx[m] = pk_fma(x23, x23, pk_fma(x01, x01, sxx[m]));
            sxy[m] = pk_fma(x23, y23, pk_fma(x01, y01, sxy[m]));
        }
    };
    auto fetch = [&](uint32_t g, v4f& y, v4f (&x)[M]) {
        y = *reinterpret_cast<const v4f*>(tmpl_a + 4u * g);
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = *reinterpret_cast<const v4f*>(xb[m] + 4u * g);
    };
    const uint32_t interior = groups > 2 ? groups - 2 : 0u;
    const uint32_t full = uni(interior >> 6);  // trips every lane takes
    {
        v4f ya, yb, xa[M], xb_[M];
        uint32_t t = 0;
        if (full) fetch(1 + lane, ya, xa);
        while (t + 2 <= full) {  // set a holds trip t
            fetch(1 + lane + 64u * (t + 1), yb, xb_);
            accumulate(ya, xa);
            fetch(1 + lane + 64u * min(t + 2, full - 1), ya, xa);  // (the last one may fetch a trip again: never accumulated twice)
            accumulate(yb, xb_);
            t += 2;
        }
        if (t < full) accumulate(ya, xa);
        const uint32_t g = 1 + lane + 64u * full;
        if (g + 1 < groups) {
            fetch(g, ya, xa);
            accumulate(ya, xa);
        }
    }
    if (lane < 8) {  // edge groups: lanes 0-3 group 0, lanes 4-7 the last group; an element counts when it lies inside the window
        const uint32_t j = lane < 4 ? lane : 4u * (groups - 1) + (lane - 4);
        const bool valid = j >= a && j - a < len && (lane < 4 || groups > 1);
        if (valid) {
            const float yv = tmpl_a[j];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float xv = xb[m][j];
                sx[m].x += xv;
                sxx[m].x = __builtin_fmaf(xv, xv, sxx[m].x);
                sxy[m].x = __builtin_fmaf(xv, yv, sxy[m].x);
            }
        }
    }
    float px[M], pxx[M], pxy[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        px[m] = sx[m].x + sx[m].y;
        pxx[m] = sxx[m].x + sxx[m].y;
        pxy[m] = sxy[m].x + sxy[m].y;
    }
    r[0] = reduce_m<M>(px);  // value m's totals in lane reduce_lane<M>(m) of r[0..2]
    r[1] = reduce_m<M>(pxx);
    r[2] = reduce_m<M>(pxy);
}
// the same sums with one template copy: work read at the 4-byte alignment of the offset (two ds_read2_b32 per group)
template <int M>
__device__ __forceinline__ void sweep_unaligned(const float* work, const float* tmpl, uint32_t len, const uint32_t (&off)[M], float (&r)[3]) {
    const unsigned lane = threadIdx.x & 63u;
    v2f sx[M], sxx[M], sxy[M];
#pragma unroll
    for (int m = 0; m < M; ++m) sx[m] = sxx[m] = sxy[m] = v2f{0.0f, 0.0f};
    const uint32_t groups = len >> 2;
    for (uint32_t g = lane; g < groups; g += 64) {
        const v4f y = *reinterpret_cast<const v4f*>(tmpl + 4u * g);
        const v2f y01{y.x, y.y}, y23{y.z, y.w};
        v4f x[M];
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = load4u(work + off[m] + 4u * g);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const v2f x01{x[m].x, x[m].y}, x23{x[m].z, x[m].w};
            sx[m] = (sx[m] + x01) + x23;
            sxx[m] = pk_fma(x23, x23, pk_fma(x01, x01, sxx[m]));
            sxy[m] = pk_fma(x23, y23, pk_fma(x01, y01, sxy[m]));
        }
    }
    const uint32_t tail = len & 3u;
    if (lane < tail) {
        const uint32_t i = 4u * groups + lane;
        const float yv = tmpl[i];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float xv = work[off[m] + i];
            sx[m].x += xv;
            sxx[m].x = __builtin_fmaf(xv, xv, sxx[m].x);
            sxy[m].x = __builtin_fmaf(xv, yv, sxy[m].x);
        }
    }
    float px[M], pxx[M], pxy[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        px[m] = sx[m].x + sx[m].y;
        pxx[m] = sxx[m].x + sxx[m].y;
        pxy[m] = sxy[m].x + sxy[m].y;
    }
    r[0] = reduce_m<M>(px);  // value m's totals in lane reduce_lane<M>(m) of r[0..2]
    r[1] = reduce_m<M>(pxx);
    r[2] = reduce_m<M>(pxy);
}

// One search round.  Entry k (scan order of :455-470) is offset top - k * step for k < cnt, then offset 0 when `extra_zero`.
// Wavefront w sweeps entries w, w + W, w + 2 W, w + 3 W together (their offsets differ by multiples of 4: one alignment class),
// the extra entry rides alone.  The sums of entry k land in sums->s[k]; the barrier at the end publishes them.
template <int T>
__device__ __forceinline__ void eval_round(Ctx<T>& c, uint32_t len, uint32_t top, uint32_t step, uint32_t cnt, bool extra_zero) {
    constexpr uint32_t W = Ctx<T>::W;
    const uint32_t wave = wave_index(), lane = threadIdx.x & 63u;
    auto run = [&](auto mtag, uint32_t k0, uint32_t kstep) {
        constexpr int M = decltype(mtag)::value;
        uint32_t off[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const uint32_t k = k0 + (uint32_t)m * kstep;
            off[m] = k < cnt ? top - k * step : 0u;
        }
        float r[3];
        if (c.aligned()) {
            const uint32_t a = off[0] & 3u;
            sweep_aligned<M>(c.work, c.tmpl + a * c.tmpl_stride, a, len, off, r);
        } else {
            sweep_unaligned<M>(c.work, c.tmpl, len, off, r);
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if ((int)lane == reduce_lane<M>(m))
                *reinterpret_cast<v4f*>(c.sums->s[c.round & 1u][k0 + (uint32_t)m * kstep]) = v4f{r[0], r[1], r[2], 0.0f};
        }
    };
    constexpr uint32_t MMAX = T >= 1024 ? 2 : 4;  // entries per sweep (1024 threads: 128 VGPRs per lane)
    for (uint32_t k0 = wave; k0 < cnt; k0 += MMAX * W) {
        const uint32_t nv = min(MMAX, (cnt - k0 + W - 1) / W);
        if constexpr (MMAX >= 4) {
            if (nv >= 4) {
                run(std::integral_constant<int, 4>{}, k0, W);
                continue;
            }
            if (nv == 3) {
                run(std::integral_constant<int, 3>{}, k0, W);
                continue;
            }
        }
        if (nv == 2) run(std::integral_constant<int, 2>{}, k0, W);
        else run(std::integral_constant<int, 1>{}, k0, W);
    }
    if (extra_zero && wave == W - 1) run(std::integral_constant<int, 1>{}, cnt, W);  // (the last wavefront never has more entries than another)
    lds_barrier();
}

// Scores of the round's entries, one per lane (normalized_correlation's closing arithmetic, :228-236), and the strict-> scan of
// the round (:455-470): the incumbent only loses to a strictly larger score, the earliest entry in scan order wins among equals.
// Every wavefront does this for itself from the published sums (identical inputs, identical result: no broadcast).
struct RoundScores {
    float sc;  // lane k: score of entry k
    uint32_t top, step, cnt;
    __device__ __forceinline__ float at(uint32_t k) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), (int)k));
    }
};
template <int T>
__device__ __forceinline__ RoundScores select_round(Ctx<T>& c, uint32_t len, float sum_y, float sum_yy, uint32_t top, uint32_t step, uint32_t cnt,
                                                    bool extra_zero, uint32_t& bo, float& bs) {
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t total = cnt + (extra_zero ? 1u : 0u);
    float v = NEG_INF;
    if (lane < total) {
        const v4f s = *reinterpret_cast<const v4f*>(c.sums->s[c.round & 1u][lane]);
        const float nf = (float)len;
        const float ey = fmaxf(sum_yy - sum_y * sum_y / nf, 0.0f);
        const float dot = s.z - s.x * sum_y / nf;
        const float ex = fmaxf(s.y - s.x * s.x / nf, 0.0f);
        const float denom = sqrtf(ex * ey);
        v = len != 0 && denom > F32_EPS ? rclamp(dot / denom, -1.0f, 1.0f) : 0.0f;
    }
    const float cand = lane < total ? v : NEG_INF;  // (a NaN score never wins: every comparison with it is false)
    const float mx = wave_all<OP_MAX>(cand);
    const uint32_t kmin = wave_all_u32<false>(cand == mx ? lane : 0xFFFFFFFFu);
    if (kmin < total && mx > bs) {
        bo = kmin < cnt ? top - kmin * step : 0u;
        bs = mx;
    }
    c.round += 1;
    return RoundScores{v, top, step, cnt};
}

// find_best (:441-484) with the template statistics (correlation_stats, :206-208) already reduced.  The reference fills a score
// cache lazily; here a round evaluates all of its entries — an entry evaluated again in a later round gives the same bits (its
// sums are a pure function of work, template and offset), which is what the cache would have returned.
template <int T>
__device__ __forceinline__ void find_best(Ctx<T>& c, uint32_t len, uint32_t search, float period, float sum_y, float sum_yy, uint32_t& best_off_out,
                          float& frac_out) {
    uint32_t stride = f2u(roundf(period / 16.0f));
    stride = min(max(stride, 1u), 128u);
    stride = min(stride, max(search, 1u));
    // coarse: (0..=search).rev().step_by(stride).chain([0])
    const uint32_t n_coarse = search / stride + 1;  // <= 1.5 P / (P / 16 - 1/2) + 2: below 40 for every period
    uint32_t best_off = search / 2;
    float best_score = NEG_INF;
    eval_round(c, len, search, stride, n_coarse, true);
    c.pc.mark(8);  // (sub-phase) sweeps + barrier
    RoundScores rs = select_round(c, len, sum_y, sum_yy, search, stride, n_coarse, true, best_off, best_score);
    c.pc.mark(9);  // (sub-phase) scores + argmax
    uint32_t step = stride;
    while (step > 1) {
        const uint32_t next = max(step / 4, 1u);
        const uint32_t lo = best_off > step ? best_off - step : 0;
        const uint32_t hi = min(best_off + step, search);
        const uint32_t cnt = (hi - lo) / next + 1;  // <= 2 step / next + 1 <= 15
        eval_round(c, len, hi, next, cnt, false);  // (lo..=hi).rev().step_by(next)
        c.pc.mark(8);
        rs = select_round(c, len, sum_y, sum_yy, hi, next, cnt, false, best_off, best_score);
        c.pc.mark(9);
        step = next;
    }
    float frac = 0.0f;
    if (best_off > 0 && best_off < search) {
        // the neighbours' scores: entries of the last round when it was a dense one that covers them, else evaluated now
        float prev, nxt;
        const bool dense = rs.step == 1 && best_off + 1 <= rs.top && rs.top - (best_off - 1) < rs.cnt;
        if (dense) {
            prev = rs.at(rs.top - (best_off - 1));
            nxt = rs.at(rs.top - (best_off + 1));
        } else {
            uint32_t bo = 0;
            float bs = NEG_INF;
            eval_round(c, len, best_off + 1, 2, 2, false);
            const RoundScores r2 = select_round(c, len, sum_y, sum_yy, best_off + 1, 2, 2, false, bo, bs);
            nxt = r2.at(0);
            prev = r2.at(1);
        }
        frac = rclamp(parabolic_refine(prev, best_score, nxt, best_off) - (float)best_off, -0.5f, 0.5f);
    }
    best_off_out = best_off;
    frac_out = frac;
}

// prepare_template (:422-439) + correlation_stats of it: -g(i) below the middle, +g(mirror) from it on (the middle element of an odd
// length ends up +g), plus the learnt reference; one pass.  Aligned mode writes the four delayed copies.
template <int T>
__device__ __forceinline__ void prepare_template(Ctx<T>& c, uint32_t len, float period, bool use_reference, float (&acc)[2]) {
    const uint32_t midpoint = len / 2;
    const float max_width = fmaxf((float)max(midpoint, 1u) / 3.0f, 1.0f);
    const float width = rclamp(SLOPE_WIDTH_PERIODS * period, 1.0f, max_width);
    acc[0] = acc[1] = 0.0f;  // per-thread partials of correlation_stats: the caller reduces them (with whatever else it has)
    const bool copies = c.aligned();
    const bool flat = len <= 1 || width <= F32_EPS;  // gaussian() returns 0 (:200)
    const float center = (float)(len - 1) * 0.5f;
    auto put = [&](uint32_t e, float v) {  // v: the edge weight plus the learnt reference (x + 0 = x when there is none)
        c.tmpl[e] = v;
        if (copies) {
            c.tmpl[c.tmpl_stride + e + 1] = v;
            c.tmpl[2 * c.tmpl_stride + e + 2] = v;
            c.tmpl[3 * c.tmpl_stride + e + 3] = v;
        }
        acc[0] += v;
        acc[1] = __builtin_fmaf(v, v, acc[1]);
    };
    // the weight of element e < len / 2 serves e (negated) and its mirror: one exponential per pair; an odd middle element is +g
    for (uint32_t e = threadIdx.x; e < (len + 1) / 2; e += T) {
        const uint32_t mirror = len - 1 - e;
        const float weight = flat ? 0.0f : gaussian_fast(center, (float)e, width);
        const float r_lo = use_reference ? c.ref[e] : 0.0f, r_hi = use_reference ? c.ref[mirror] : 0.0f;
        if (mirror != e) put(e, -0.5f * EDGE_STRENGTH * 2.0f * weight + r_lo);
        put(mirror, 0.5f * EDGE_STRENGTH * 2.0f * weight + r_hi);
    }
    if (copies && threadIdx.x < 32) {  // zero padding of copy a: a floats in front, up to the next multiple of four (+4) behind
        const uint32_t a = threadIdx.x >> 3, q = threadIdx.x & 7u;
        float* ta = c.tmpl + a * c.tmpl_stride;
        if (q < a) ta[q] = 0.0f;
        ta[len + a + q] = 0.0f;
    }
}

// write_candidate (:509-527): candidate = windowed, peak-normalised, mean-removed segment; returns its correlation with the
// reference.  Two reductions: (sum, max, min) of the segment give mean and peak (|x - mean| is largest at an extreme of x, and
// rounding is monotone, so the peak is exact); then the five sums of normalized_correlation + correlation_stats.
template <int T>
__device__ __forceinline__ float write_candidate(Ctx<T>& c, const float* seg, uint32_t n, float period) {
    float* cand = c.tmpl;
    float r3[3] = {0.0f, NEG_INF, -NEG_INF};
    for (uint32_t i = threadIdx.x; i < n; i += T) {
        const float v = seg[i];
        r3[0] += v;
        r3[1] = fmaxf(r3[1], v);
        r3[2] = fminf(r3[2], v);
    }
    c.red.template run<3, (OP_SUM) | (OP_MAX << 2) | (OP_MIN << 4)>(r3);
    const float mean = r3[0] / (float)max(n, 1u);
    const float pk = n ? fmaxf(fabsf(r3[1] - mean), fabsf(r3[2] - mean)) : 0.0f;
    const float scale = 1.0f / fmaxf(pk, NORMALIZE_FLOOR);  // normalize_peak (:191-197)
    const float std_ = fmaxf(period * BUFFER_FALLOFF_PERIODS, 1.0f);
    float acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // sum y, sum y^2, sum x, sum x^2, sum x y
    const bool flat = n <= 1 || std_ <= F32_EPS;
    const float center = (float)(n - 1) * 0.5f;
    auto put = [&](uint32_t i, float sv, float xv, float weight) {
        float yv = (sv - mean) * scale;
        yv *= weight;
        cand[i] = yv;
        acc[0] += yv;
        acc[1] = __builtin_fmaf(yv, yv, acc[1]);
        acc[2] += xv;
        acc[3] = __builtin_fmaf(xv, xv, acc[3]);
        acc[4] = __builtin_fmaf(xv, yv, acc[4]);
    };
    for (uint32_t i = threadIdx.x; i < (n + 1) / 2; i += T) {  // one weight per mirrored pair (:519-526)
        const uint32_t mirror = n - 1 - i;
        const float weight = flat ? 0.0f : gaussian_fast(center, (float)i, std_);
        const float s_lo = seg[i], s_hi = seg[mirror], x_lo = c.ref[i], x_hi = c.ref[mirror];
        put(i, s_lo, x_lo, weight);
        if (mirror != i) put(mirror, s_hi, x_hi, weight);
    }
    c.red.template run<5, 0>(acc);  // its barrier also publishes cand[]
    if (n == 0) return 0.0f;
    const float nf = (float)n;
    const float dot = acc[4] - acc[2] * acc[0] / nf;
    const float ex = fmaxf(acc[3] - acc[2] * acc[2] / nf, 0.0f);
    const float ey = fmaxf(acc[1] - acc[0] * acc[0] / nf, 0.0f);
    const float denom = sqrtf(ex * ey);
    return denom > F32_EPS ? rclamp(dot / denom, -1.0f, 1.0f) : 0.0f;
}

// locate (:358-411) on the LDS-resident reference of this trigger
template <int T>
__device__ __forceinline__ Capture locate(Ctx<T>& c, uint32_t& t_ref_len, float& t_reference_period, float& t_mean, const View& trace, Estimate est,
                                          uint32_t cycles, float rate) {
    const Capture none{0, 0.0f, 0, 0.0f};
    const uint32_t n = trace.n;
    const float period = fmaxf(est.period, 1.0f);
    const float span = period * (float)max(cycles, 1u);
    const uint32_t frames = f2u(ceilf(span)) + 1;
    const uint32_t len = trigger_kernel_len(period, rate);
    const uint32_t before = len / 2, after = len - before;
    const uint32_t tail = max(frames, after);
    if (n < tail) return none;
    const uint32_t right = n - tail;
    if (right < before) return none;
    uint32_t search = max(f2u(roundf(period * SEARCH_PERIODS)), 1u);
    search = min(min(search, len / 2), right - before);
    const uint32_t left = right - search;
    const View data = trace.sub(left - before, (right + after) - (left - before));  // len + search samples

    // this block's LDS layout
    {
        const uint32_t span4 = (data.n + 8 + 3) & ~3u, len4 = (len + 12 + 3) & ~3u;
        c.raw = c.dyn;
        c.work = c.dyn + span4;
        c.tmpl = c.dyn + 2 * span4;
        c.tmpl_stride = 2 * span4 + 4 * len4 <= c.dyn_floats ? len4 : 0u;
    }
    // the block's span: the one global read of the pass — LDS-DMA (global_load_lds_dword: ring -> raw[] without registers), issued
    // now, landing while the template is prepared.  Wavefront w moves elements q T + 64 w + lane, i.e. every thread's own strided
    // elements: after its own vmcnt(0) a thread may read them without a barrier.
    {
        const uint32_t wave = wave_index(), lane = threadIdx.x & 63u;
        for (uint32_t base = wave * 64u; base < data.n; base += T) {
            const uint32_t i = base + lane;
            if (i < data.n)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(data.ring + ((data.start + i) & data.mask)),
                                                 (__attribute__((address_space(3))) void*)(c.raw + base), 4, 0, 0);
        }
    }
    c.pc.mark(1);  // span load issued
    // prepare (:413-420): retune_reference (:486-498) ...
    if (t_ref_len == 0) {
        for (uint32_t i = threadIdx.x; i < len; i += T) c.ref[i] = 0.0f;
        t_ref_len = len;
        t_reference_period = period;
        c.ref_peak = 0.0f;
        lds_barrier();
    } else {
        const float semitones = log2f(period / t_reference_period) * 12.0f;
        if (t_ref_len != len || fabsf(semitones) >= BUFFER_RETUNE_SEMITONES) {  // retune_reference fn (:249-263), through tmpl[]
            const float ratio = period / t_reference_period;
            const bool bad = !isfinite(ratio) || ratio <= F32_EPS;
            const float old_center = (float)(t_ref_len ? t_ref_len - 1 : 0) * 0.5f;
            const float new_center = (float)(len ? len - 1 : 0) * 0.5f;
            float pk[1] = {0.0f};
            for (uint32_t i = threadIdx.x; i < len; i += T) {
                const float v = bad ? 0.0f : sample_linear_zero(c.ref, t_ref_len, old_center + ((float)i - new_center) / ratio);
                c.tmpl[i] = v;
                pk[0] = fmaxf(pk[0], fabsf(v));
            }
            c.red.template run<1, OP_MAX>(pk);
            c.ref_peak = pk[0];
            for (uint32_t i = threadIdx.x; i < len; i += T) c.ref[i] = c.tmpl[i];
            t_ref_len = len;
            t_reference_period = period;
            lds_barrier();
        }
    }
    // ... any(|reference| > 1e-3) (:381) as the reference's peak (carried along: taken where the reference last changed) ...
    const float ref_peak = c.ref_peak;
    const bool use_reference = ref_peak > 1.0e-3f;
    const bool confident = est.confidence >= MIN_PERIODICITY;
    uint32_t offset = 0;
    float frac_offset = 0.0f;
    bool reset = false;
    // (:383-399) as one loop: pass 0 searches with the learnt reference in the template; a confident estimate then writes the
    // candidate, and a candidate that no longer matches the reference (< 0.3) clears it and searches once more (pass 1)
    for (int pass = 0;; ++pass) {
        float acc[3];
        {
            float ty[2];
            prepare_template(c, len, period, use_reference && pass == 0, ty);  // does not depend on the span
            acc[0] = ty[0];
            acc[1] = ty[1];
            acc[2] = 0.0f;
        }
        if (pass == 0) {
            c.pc.mark(2);  // retune, template
            // ... and the EMA-tracked mean of the span: its sum rides in the same reduction as the template's statistics
            wait_vm0();  // this wavefront's part of the span has landed
            for (uint32_t i = threadIdx.x; i < data.n; i += T) acc[2] += c.raw[i];
        }
        c.red.template run<3, 0>(acc);  // its barrier also publishes tmpl[]
        const float sum_y = acc[0], sum_yy = acc[1];
        if (pass == 0) {
            const float mean = acc[2] / (float)max(data.n, 1u);
            t_mean += MEAN_RESPONSIVENESS * (mean - t_mean);
            const float tm = t_mean;
            for (uint32_t i = threadIdx.x; i < data.n; i += T) c.work[i] = c.raw[i] - tm;
            if (threadIdx.x < 8) c.work[data.n + threadIdx.x] = 0.0f;  // read (and masked out) by the last aligned group
            lds_barrier();  // work[] (and every wavefront's part of raw[])
            c.pc.mark(3);  // template statistics + span mean, work[]
        }
        find_best(c, len, search, period, sum_y, sum_yy, offset, frac_offset);
        c.pc.mark(4);  // coarse-to-fine search
        if (!confident) break;
        // segment(offset) = trace[left + offset - before ..][..len] = raw[offset ..]
        const float match = write_candidate(c, c.raw + offset, len, period);
        c.pc.mark(5);  // candidate vs reference
        if (pass == 0 && use_reference && match < RESET_BELOW_MATCH) {
            reset = true;
            for (uint32_t i = threadIdx.x; i < len; i += T) c.ref[i] = 0.0f;
            continue;
        }
        break;
    }
    if (confident) {
        // update_reference (:500-507): normalize_peak (:191-197) by the carried peak (0 after a reset), then the EMA, one pass;
        // the new reference's peak is taken on the way (its reduction is also the barrier that ends this capture's use of ref[])
        const float rscale = 1.0f / fmaxf(reset ? 0.0f : ref_peak, NORMALIZE_FLOOR);
        float pk[1] = {0.0f};
        for (uint32_t i = threadIdx.x; i < len; i += T) {
            float rv = c.ref[i] * rscale;
            rv += BUFFER_RESPONSIVENESS * (c.tmpl[i] - rv);
            c.ref[i] = rv;
            pk[0] = fmaxf(pk[0], fabsf(rv));
        }
        c.red.template run<1, OP_MAX>(pk);
        c.ref_peak = pk[0];
        t_reference_period += BUFFER_RESPONSIVENESS * (period - t_reference_period);
    }
    c.pc.mark(6);  // reference update
    uint32_t start = left + offset;
    if (frac_offset < 0.0f && start > 0) {
        start -= 1;
        frac_offset += 1.0f;
    }
    return Capture{1, span, start, frac_offset};
}

// find_rising_zero_crossing over indices lo..=hi (:530-551); reversed = iterate from hi down
template <int T>
__device__ __forceinline__ uint32_t find_rising_zero_crossing(Ctx<T>& c, const View& v, uint32_t lo, uint32_t hi, bool reversed) {
    if (lo > hi) return 0xFFFFFFFFu;  // callers only pass in-range spans (hi < v.n)
    // a crossing between adjacent indices (i - 1, i): v[i] > 0 && v[i - 1] <= 0, reported as i
    if (!reversed) {
        uint32_t best = 0xFFFFFFFFu;
        for (uint32_t i = lo + 1 + threadIdx.x; i <= hi; i += T)
            if (v.at(i) > 0.0f && v.at(i - 1) <= 0.0f) {
                best = i;
                break;
            }
        return c.red.template run_u32<false>(best);
    }
    uint32_t best = 0u;
    for (uint32_t i = lo + 1 + threadIdx.x; i <= hi; i += T)
        if (v.at(i) > 0.0f && v.at(i - 1) <= 0.0f) best = i + 1u;  // keep the largest
    best = c.red.template run_u32<true>(best);
    return best == 0u ? 0xFFFFFFFFu : best - 1u;
}

// zero_crossing_capture (:769-786)
template <int T>
__device__ __forceinline__ Capture zero_crossing_capture(Ctx<T>& c, const View& v, uint32_t frames_in, uint32_t search_range) {
    const uint32_t frames = min(frames_in, v.n);
    if (frames == 0) return Capture{0, 0.0f, 0, 0.0f};
    const uint32_t end = v.n > 0 ? v.n - 1 : 0;
    const uint32_t right_lo = end > search_range ? end - search_range : 0;
    uint32_t right = find_rising_zero_crossing(c, v, right_lo, end, true);
    if (right == 0xFFFFFFFFu) right = end;
    const uint32_t left_lo = right > frames ? right - frames : 0;
    const uint32_t left_hi = min(left_lo + search_range, right > 2 ? right - 2 : 0u);
    uint32_t left = find_rising_zero_crossing(c, v, left_lo, left_hi, false);
    if (left == 0xFFFFFFFFu) left = left_lo;
    return Capture{1, (float)max(right > left ? right - left : 0u, 1u), left, 0.0f};
}

// LDS floats of the trigger kernel: the reference (fixed) and the per-block arrays at their largest — raw span, mean-removed
// span, one template copy; whatever the launch adds on top is room for the aligned template copies of shorter periods
struct TriggerLayout {
    uint32_t ref, dyn_min;
};
__host__ __device__ inline TriggerLayout trigger_layout(uint32_t max_kernel, uint32_t max_period) {
    const uint32_t ms = (uint32_t)(((uint64_t)max_period * 3 + 1) / 2) + 2;  // ceil(max_period * SEARCH_PERIODS) + 2
    TriggerLayout l;
    l.ref = (max_kernel + 8 + 3) & ~3u;
    const uint32_t span4 = (max_kernel + ms + 8 + 3) & ~3u, len4 = (max_kernel + 12 + 3) & ~3u;
    l.dyn_min = 2 * span4 + len4;
    return l;
}

}  // namespace

// StableTrigger::capture's unlock (:298-304) + stabilize (:336-356) as a pure function of the trigger's registers and the block's
// estimate: the registers after the step, and the estimate `locate` is called with (some = 0: no capture from locate)
__device__ __forceinline__ Estimate stabilise(TrigRegs& r, const ScopeEstimate& pre, uint32_t probe_len) {
    if (probe_len > 0 && pre.last_peak < MIN_SIGNAL_PEAK) {  // unlock
        r.has_period = 0;
        r.missed_periods = 0;
        r.ref_len = 0;
        r.reference_period = 0.0f;
        r.mean = 0.0f;
    }
    Estimate est{pre.some, pre.period, pre.confidence};
    if (!est.some) {
        if (r.has_period) {
            r.missed_periods = r.missed_periods >= 255 ? 255 : r.missed_periods + 1;
            if (r.missed_periods > MAX_MISSED_PERIODS) {
                r.has_period = 0;
                r.missed_periods = 0;
                r.ref_len = 0;
                r.reference_period = 0.0f;
                r.mean = 0.0f;
            } else {
                est = Estimate{1, r.period, 0.0f};
            }
        }
    } else {
        r.missed_periods = 0;
        if (r.has_period) {
            const float q = est.period / r.period;
            if (q >= 0.9f && q <= 1.1f) est.period = r.period + 0.35f * (est.period - r.period);
        }
        r.has_period = 1;
        r.period = est.period;
    }
    return est;
}

// whether `locate` for this estimate fits its arrays (raw span, mean-removed span, one template copy) into `dyn_floats` and its kernel
// into the reference region: the head of locate (:358-372), without side effects
__device__ __forceinline__ bool locate_fits(uint32_t n, float est_period, uint32_t cycles, float rate, uint32_t dyn_floats, uint32_t ref_cap) {
    const float period = fmaxf(est_period, 1.0f);
    const float span = period * (float)max(cycles, 1u);
    const uint32_t frames = f2u(ceilf(span)) + 1;
    const uint32_t len = trigger_kernel_len(period, rate);
    const uint32_t before = len / 2, after = len - before;
    const uint32_t tail = max(frames, after);
    if (n < tail) return true;  // (locate returns before it touches LDS)
    const uint32_t right = n - tail;
    if (right < before) return true;
    uint32_t search = max(f2u(roundf(period * SEARCH_PERIODS)), 1u);
    search = min(min(search, len / 2), right - before);
    const uint32_t span4 = (len + search + 8 + 3) & ~3u, len4 = (len + 12 + 3) & ~3u;
    return 2u * span4 + len4 <= dyn_floats && len <= ref_cap;
}

template <int T>
__global__ __launch_bounds__(T) void scope_trigger_kernel(ScopeArgs a, uint32_t lds_floats) {
    constexpr int W = T / 64;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    __shared__ RedSlots<W> slots;
    __shared__ __attribute__((aligned(16))) RoundSums sums;
    __shared__ ScopeTriggerState trig[kScopeTraces];
    __shared__ uint64_t s_head[2][kScopeTraces], s_len[2][kScopeTraces];  // by block parity: one barrier per block publishes them
    constexpr uint32_t kEstChunk = 64;
    __shared__ ScopeEstimate est_lds[kEstChunk][kScopeTraces];  // the estimates of 64 blocks at a time
    const unsigned tid = threadIdx.x;
    const uint32_t s = blockIdx.x;
    Ctx<T> c;
    // (ref_cap != 0: launched with less LDS than the worst case — blocks whose arrays do not fit are handed over, see below)
    const uint32_t ref_cap = a.ref_cap ? a.ref_cap : a.max_kernel;
    {
        const uint32_t ref_floats = a.ref_cap ? ((a.ref_cap + 8 + 3) & ~3u) : trigger_layout(a.max_kernel, a.max_period).ref;
        c.ref = lds_f;
        c.dyn = lds_f + ref_floats;
        c.dyn_floats = lds_floats - ref_floats;
        c.raw = c.work = c.tmpl = c.dyn;
        c.tmpl_stride = 0;
    }
    c.sums = &sums;
    c.round = 0;
    c.red.slots = &slots;
    c.red.phase = 0;
    // ragged banks: the stream's own block count, ring positions and reset flag (workgroup-uniform: one workgroup per stream)
    const bool ragged = a.blocks_v != nullptr;
    const bool reset_stream = ragged && a.reset_v != nullptr && a.reset_v[s] != 0;  // clear_history (:714-723) of this stream
    const uint32_t n_blocks_s = ragged ? a.blocks_v[s] : a.n_blocks;
    const uint32_t block_frames_s = a.frames_v != nullptr ? a.frames_v[s] : a.block_frames;  // chunk calls: the stream's own block length
    if (tid < kScopeTraces) {
        const ScopeTriggerState* src = a.trig + (uint64_t)s * kScopeTraces + tid;
        store_trig(trig[tid], reset_stream ? TrigRegs{0, 0.0f, 0, 0.0f, 0.0f, 0} : load_trig(*src));
        trig[tid]._pad[0] = trig[tid]._pad[1] = 0;
        s_head[1][tid] = ragged ? a.pos_v[((uint64_t)s * kScopeTraces + tid) * 2] : a.head[tid];
        s_len[1][tid] = ragged ? (reset_stream ? 0ull : a.pos_v[((uint64_t)s * kScopeTraces + tid) * 2 + 1]) : a.len[tid];
    }
    const uint64_t mask = a.cap - 1;
    const float* rings = a.rings + (uint64_t)s * kScopeTraces * a.cap;
    const bool active0 = a.trace_channel[0] != OMX_CHANNEL_NONE, active1 = a.trace_channel[1] != OMX_CHANNEL_NONE;
    const bool stable = a.trigger_mode != OMX_TRIGGER_ZERO_CROSSING;
    const int linked_view = a.matching_trace >= 0 ? a.matching_trace : (a.separate_source ? 2 : -1);
    int resident = -1;  // the trigger whose reference sits in c.ref
    __syncthreads();
    __shared__ unsigned long long phase_acc[SCOPE_PHASES];
    c.pc.start(a.phase_timing != 0, phase_acc);

    uint32_t blocks_run = n_blocks_s;
    for (uint32_t blk = 0; blk < n_blocks_s; ++blk) {
        const uint32_t par = blk & 1u;
        if (tid < kScopeTraces) {  // the block's frames are in the rings already (scope_push_kernel)
            const bool on = tid == 0 ? active0 : (tid == 1 ? active1 : a.separate_source != 0);
            s_head[par][tid] = s_head[par ^ 1u][tid] + (on ? block_frames_s : 0u);
            s_len[par][tid] = on ? min(s_len[par ^ 1u][tid] + (uint64_t)block_frames_s, (uint64_t)a.history_frames) : 0ull;
        }
        if (stable && blk % kEstChunk == 0) {  // the next 64 blocks' estimates (the barrier below publishes them)
            const uint32_t cnt = min(kEstChunk, n_blocks_s - blk) * kScopeTraces;
            const ScopeEstimate* src = a.estimates + ((uint64_t)s * a.n_blocks + blk) * kScopeTraces;
            if (blk != 0) lds_barrier();  // every thread has taken its estimate of the chunk's last block
            if (tid < cnt) (&est_lds[0][0])[tid] = src[tid];
        }
        lds_barrier();  // positions, estimates; and the previous block is through with every LDS array
        if (a.resume_blk != nullptr && stable) {
            // Capped LDS: will every capture of this block (the job loop below, dry) fit its arrays?  Decided before anything of the block
            // is touched — the trigger records, the resident reference — from values every thread holds alike; a block that does not fit
            // ends this kernel's part of the stream, the one-workgroup-per-stream kernel continues from here (oscilloscope_kernels.hip).
            bool fits = true;
            const bool linked_runs = linked_view >= 0 && uni((uint32_t)s_len[par][linked_view]) >= a.base_frames;
            for (int job = 0; job < 3; ++job) {
                int view_index, trig_index;
                if (job == 0) {
                    if (linked_view < 0) continue;
                    view_index = linked_view;
                    trig_index = 2;
                } else {
                    if (!(job == 1 ? active0 : active1) || linked_runs) continue;
                    view_index = trig_index = job - 1;
                }
                const uint32_t n = uni((uint32_t)s_len[par][view_index]);
                if (n < a.base_frames) continue;
                TrigRegs r = load_trig_uniform(trig[trig_index]);
                fits = fits && r.ref_len <= ref_cap;  // the learnt reference itself
                const ScopeEstimate& pl = est_lds[blk % kEstChunk][view_index];
                const ScopeEstimate pre{uni(pl.some), uni(pl.period), uni(pl.confidence), uni(pl.last_peak)};
                const Estimate est = stabilise(r, pre, min(a.probe_frames, n));
                if (est.some) fits = fits && locate_fits(n, est.period, a.num_cycles, a.sample_rate, c.dyn_floats, ref_cap);
            }
            if (!fits) {
                blocks_run = blk;
                break;
            }
        }
        auto view_of = [&](int t) {
            return View{rings + (uint64_t)t * a.cap, uni((uint32_t)((s_head[par][t] - s_len[par][t]) & mask)), (uint32_t)mask, uni((uint32_t)s_len[par][t])};
        };

        // ---- captures (:683-700): job 0 the linked capture, jobs 1 / 2 the slots' own when there is no linked one
        Capture linked{0, 0.0f, 0, 0.0f}, cap0{0, 0.0f, 0, 0.0f}, cap1{0, 0.0f, 0, 0.0f};
        bool ran = false;
        for (int job = 0; job < 3; ++job) {
            int view_index, trig_index;
            if (job == 0) {
                if (linked_view < 0) continue;
                view_index = linked_view;
                trig_index = 2;
            } else {
                if (!(job == 1 ? active0 : active1)) continue;
                if (linked.some) {
                    if (job == 1) cap0 = linked;
                    else cap1 = linked;
                    continue;
                }
                view_index = trig_index = job - 1;
            }
            const View trace = view_of(view_index);
            Capture cap{0, 0.0f, 0, 0.0f};
            if (ran) lds_barrier();  // a second capture in one block: the first one's LDS arrays and trigger record are settled
            if (!stable) {
                cap = zero_crossing_capture(c, trace, a.base_frames, a.max_period);
            } else if (trace.n >= a.base_frames) {
                // StableTrigger::capture (:306-334); the trigger state lives in LDS, thread-uniform updates are done redundantly
                const ScopeEstimate& pl = est_lds[blk % kEstChunk][view_index];
                const ScopeEstimate pre{uni(pl.some), uni(pl.period), uni(pl.confidence), uni(pl.last_peak)};
                // (plain scalars, not a struct handed around by reference: hipcc otherwise keeps two of the fields in scratch memory,
                // and a scratch reload costs a global-memory round trip per block)
                int has_period = uni(trig[trig_index].has_period);
                float period = uni(trig[trig_index].period);
                uint32_t missed = uni(trig[trig_index].missed_periods);
                float reference_period = uni(trig[trig_index].reference_period);
                float mean = uni(trig[trig_index].mean);
                uint32_t ref_len = uni(trig[trig_index].ref_len);
                if (resident != trig_index) {  // bring this trigger's learnt reference into LDS (kept there for the rest of the call)
                    if (resident >= 0) {
                        float* g = a.reference + ((uint64_t)s * kScopeTraces + resident) * a.max_kernel;
                        for (uint32_t i = tid; i < trig[resident].ref_len; i += T) g[i] = c.ref[i];
                    }
                    __syncthreads();
                    const float* gn = a.reference + ((uint64_t)s * kScopeTraces + trig_index) * a.max_kernel;
                    float pk[1] = {0.0f};
                    for (uint32_t i = tid; i < ref_len; i += T) {
                        const float v = gn[i];
                        c.ref[i] = v;
                        pk[0] = fmaxf(pk[0], fabsf(v));
                    }
                    c.red.template run<1, OP_MAX>(pk);  // (its barrier publishes ref[])
                    c.ref_peak = pk[0];
                    resident = trig_index;
                }
                c.pc.mark(0);  // bookkeeping
                const uint32_t probe_len = min(a.probe_frames, trace.n);
                TrigRegs regs{has_period, period, missed, reference_period, mean, ref_len};
                const Estimate est = stabilise(regs, pre, probe_len);
                has_period = regs.has_period;
                period = regs.period;
                missed = regs.missed_periods;
                reference_period = regs.reference_period;
                mean = regs.mean;
                ref_len = regs.ref_len;
                if (est.some) cap = locate(c, ref_len, reference_period, mean, trace, est, a.num_cycles, a.sample_rate);
                if (!cap.some) {
                    cap.some = 1;
                    cap.span = (float)max(a.base_frames > 0 ? a.base_frames - 1 : 0u, 1u);
                    cap.start = trace.n > a.base_frames ? trace.n - a.base_frames : 0;
                    cap.frac_offset = 0.0f;
                }
                const TrigRegs local{has_period, period, missed, reference_period, mean, ref_len};
                if (tid == 0) store_trig(trig[trig_index], local);
                ran = true;
            }
            if (job == 0) linked = cap;
            else if (job == 1) cap0 = cap;
            else cap1 = cap;
        }

        // ---- write_snapshot (:725-750) + downsample_trace (:788-803)
        uint32_t produced = 0, channels = 0, slot_a = 0, slot_b = 0, spc = 0, cstart = 0;
        float cfrac = 0.0f;
        if (cap0.some || cap1.some) {
            const uint32_t t0 = f2u(fmaxf(roundf(cap0.span), 1.0f)) + 1, t1 = f2u(fmaxf(roundf(cap1.span), 1.0f)) + 1;
            uint32_t target = cap0.some && cap1.some ? max(t0, t1) : (cap0.some ? t0 : t1);
            target = min(max(target, 2u), (uint32_t)kScopeTarget);
            produced = 1;
            cstart = cap0.some ? cap0.start : cap1.start;
            cfrac = cap0.some ? cap0.frac_offset : cap1.frac_offset;
            const bool newest = blk + 1 == n_blocks_s;
            for (int slot = 0; slot < 2; ++slot) {
                const Capture cs = slot == 0 ? cap0 : cap1;
                if (!cs.some) continue;
                const View tr = view_of(slot);
                const uint32_t start = min(cs.start, tr.n);
                const View data = tr.sub(start, tr.n - start);
                if (data.n < 2) continue;
                const float last = (float)(data.n - 1);
                const float start_offset = rclamp(cs.frac_offset, 0.0f, last);
                const float span = fminf(cs.span, last - start_offset);
                if (!(isfinite(span) && span > 0.0f)) continue;
                const float step = span / (float)(target - 1);
                if (newest) {
                    float* out = a.samples + ((uint64_t)s * 2 + channels) * kScopeTarget;
                    for (uint32_t i = tid; i < target; i += T) out[i] = sample_linear_zero_view(data, start_offset + (float)i * step);
                }
                if (channels == 0) slot_a = (uint32_t)slot;
                else slot_b = (uint32_t)slot;
                channels += 1;
            }
            spc = channels == 0 ? 0 : target;
        }
        if (tid == 0) {
            ScopeBlockHeader hdr;
            hdr.produced = produced;
            hdr.channels = channels;
            hdr.slots[0] = slot_a;
            hdr.slots[1] = slot_b;
            hdr.samples_per_channel = spc;
            // last_cycle_rate (:602-609): source trigger first, then the traces
            const int which = trig[2].has_period ? 2 : (trig[0].has_period ? 0 : (trig[1].has_period ? 1 : -1));
            hdr.locked = which >= 0 ? 1 : 0;
            hdr.period = which >= 0 ? trig[which].period : 0.0f;
            hdr.capture_start = cstart;
            hdr.capture_frac = cfrac;
            hdr._pad = 0;
            a.headers[(uint64_t)s * a.n_blocks + blk] = hdr;
        }
        c.pc.mark(7);  // snapshot + header
    }
    __syncthreads();
    if (resident >= 0) {
        float* g = a.reference + ((uint64_t)s * kScopeTraces + resident) * a.max_kernel;
        for (uint32_t i = tid; i < trig[resident].ref_len; i += T) g[i] = c.ref[i];
    }
    if (tid < kScopeTraces) {
        a.trig[(uint64_t)s * kScopeTraces + tid] = trig[tid];
        const uint32_t last = (blocks_run & 1u) ^ 1u;  // parity of the last block run (1 = the initial slot when none ran)
        if (ragged) {
            a.pos_v[((uint64_t)s * kScopeTraces + tid) * 2] = s_head[last][tid];
            a.pos_v[((uint64_t)s * kScopeTraces + tid) * 2 + 1] = s_len[last][tid];
        }
        if (a.resume_blk != nullptr) {  // where the one-workgroup-per-stream kernel takes over (nowhere: blocks_run == the stream's count)
            a.resume_pos[((uint64_t)s * kScopeTraces + tid) * 2] = s_head[last][tid];
            a.resume_pos[((uint64_t)s * kScopeTraces + tid) * 2 + 1] = s_len[last][tid];
            if (tid == 0) a.resume_blk[s] = blocks_run;
        }
    }
    if (ragged && tid == 0 && reset_stream) a.epoch_v[s] += 1;
    c.pc.flush();
}

// ---------------------------------------------------------------- debug: find_best on caller-supplied arrays
// (tests: scores against the oracle's normalized_correlation at every offset, near-tie behaviour of the coarse-to-fine walk)
__global__ __launch_bounds__(512) void scope_find_best_debug_kernel(const float* work, const float* tmpl, uint32_t len, uint32_t search,
                                                                     float period, uint32_t lds_floats, uint32_t* best_off, float* frac_offset,
                                                                     float* best_score, float* scores) {
    constexpr int T = 512;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    __shared__ RedSlots<T / 64> slots;
    __shared__ __attribute__((aligned(16))) RoundSums sums;
    Ctx<T> c;
    const uint32_t span = len + search, span4 = (span + 8 + 3) & ~3u, len4 = (len + 12 + 3) & ~3u;
    c.ref = c.dyn = lds_f;
    c.raw = c.work = lds_f;
    c.tmpl = lds_f + span4;
    c.dyn_floats = lds_floats;
    c.tmpl_stride = span4 + 4 * len4 <= lds_floats ? len4 : 0u;
    c.sums = &sums;
    c.round = 0;
    c.red.slots = &slots;
    c.red.phase = 0;
    c.ref_peak = 0.0f;
    c.pc.start(false, nullptr);
    for (uint32_t i = threadIdx.x; i < span; i += T) c.work[i] = work[i];
    if (threadIdx.x < 8) c.work[span + threadIdx.x] = 0.0f;
    float acc[2] = {0.0f, 0.0f};
    for (uint32_t e = threadIdx.x; e < len; e += T) {
        const float v = tmpl[e];
        c.tmpl[e] = v;
        if (c.aligned()) {
            c.tmpl[c.tmpl_stride + e + 1] = v;
            c.tmpl[2 * c.tmpl_stride + e + 2] = v;
            c.tmpl[3 * c.tmpl_stride + e + 3] = v;
        }
        acc[0] += v;
        acc[1] = __builtin_fmaf(v, v, acc[1]);
    }
    if (c.aligned() && threadIdx.x < 32) {
        const uint32_t a = threadIdx.x >> 3, q = threadIdx.x & 7u;
        float* ta = c.tmpl + a * c.tmpl_stride;
        if (q < a) ta[q] = 0.0f;
        ta[len + a + q] = 0.0f;
    }
    c.red.template run<2, 0>(acc);
    uint32_t off = 0;
    float frac = 0.0f;
    find_best(c, len, search, period, acc[0], acc[1], off, frac);
    // every offset's score, 64 entries per round
    for (uint32_t top = search;; top -= 64) {
        const uint32_t cnt = min(64u, top + 1);
        uint32_t bo = 0;
        float bs = NEG_INF;
        eval_round(c, len, top, 1, cnt, false);
        const RoundScores rs = select_round(c, len, acc[0], acc[1], top, 1, cnt, false, bo, bs);
        if (threadIdx.x < cnt) scores[top - threadIdx.x] = rs.sc;
        if (threadIdx.x == 0 && off <= top && top - off < cnt) *best_score = rs.at(top - off);
        if (top < 64) break;
    }
    if (threadIdx.x == 0) {
        *best_off = off;
        *frac_offset = frac;
    }
}

void launch_scope_find_best_debug(const float* d_work, const float* d_tmpl, uint32_t len, uint32_t search, float period, uint32_t* d_best_off,
                                  float* d_frac, float* d_best_score, float* d_scores, hipStream_t stream) {
    const uint32_t span4 = (len + search + 8 + 3) & ~3u, len4 = (len + 12 + 3) & ~3u;
    const uint32_t lds_floats = std::min<uint32_t>(span4 + 4 * len4, 152 * 1024 / sizeof(float));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scope_find_best_debug_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    hipLaunchKernelGGL(scope_find_best_debug_kernel, dim3(1), dim3(512), (size_t)lds_floats * sizeof(float), stream, d_work, d_tmpl, len, search,
                       period, lds_floats, d_best_off, d_frac, d_best_score, d_scores);
}

// ---------------------------------------------------------------- scope_push_kernel
// every frame of the call projected into the trace rings at once (dsp.rs:223-249 stereo fold, channel.rs:13-21)
__global__ __launch_bounds__(256) void scope_push2_kernel(ScopeArgs a) {
    const uint32_t s = blockIdx.y;
    const uint64_t f = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t frames_s = a.blocks_v ? (uint64_t)a.blocks_v[s] * (a.frames_v != nullptr ? a.frames_v[s] : a.block_frames) : a.frames_total;
    if (f >= frames_s) return;
    const float* frame = a.pcm + ((uint64_t)s * a.frames_total + f) * a.fmt.channels;
    float left = 0.0f, right = 0.0f;
    for (uint32_t ch = 0; ch < a.fmt.channels; ++ch) {
        const float v = frame[ch];
        left = left + v * a.fmt.m[ch][0];
        right = right + v * a.fmt.m[ch][1];
    }
    float* rings = a.rings + (uint64_t)s * kScopeTraces * a.cap;
    const uint64_t mask = a.cap - 1;
    for (int t = 0; t < kScopeTraces; ++t) {
        const uint32_t ch = t < 2 ? a.trace_channel[t] : a.trigger_source;
        const bool on = t < 2 ? a.trace_channel[t] != OMX_CHANNEL_NONE : a.separate_source != 0;
        if (!on) continue;
        const uint64_t head = a.pos_v ? a.pos_v[((uint64_t)s * kScopeTraces + t) * 2] : a.head[t];
        rings[(uint64_t)t * a.cap + ((head + f) & mask)] = scope_project(ch, left, right);
    }
}

// ---------------------------------------------------------------- scope_estimate2_kernel
// One workgroup per (stream, block, candidate view): the period estimate the trigger pass will ask for after block `blk`.
//   probe (the newest probe_frames samples) -> registers in the packed-real layout z[m] = x[2m] + i x[2m+1], m = j + 256 t
//   (sum, max, min) -> mean, last_peak                                                    [one reduction]
//   centred squares -> LDS, chunked prefix scan -> energy prefix E (in the transform buffer), NSDF denominators
//   D[tau] = E[n - tau] + (E[n] - E[tau]) -> their own 9.6 KiB                             [scan + two barriers]
//   autocorrelation = IFFT_4096(packed |FFT_8192|^2) through one 4096-point transform each way (as in round 1)
//   nsdf[tau] written over D[tau]; zero crossing, best candidate, first candidate within the cutoff [three reductions]
#ifdef OMX_EST_PHASES
#define EST_T0 long long est_t = clock64(); long long est_ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define EST_MARK(i) { const long long now = clock64(); est_ph[i] += now - est_t; est_t = now; }
#define EST_REPORT if (threadIdx.x == 0 && blockIdx.x == 3 && blockIdx.y == 40) printf("estimate2 phases: load+reduce %lld | squares+scan %lld | D %lld | fwd %lld | power %lld | inv %lld | nsdf %lld | zero crossing %lld | candidates %lld\n", est_ph[0], est_ph[1], est_ph[2], est_ph[3], est_ph[4], est_ph[5], est_ph[6], est_ph[7], est_ph[8]);
#else
#define EST_T0
#define EST_MARK(i)
#define EST_REPORT
#endif
__global__ __launch_bounds__(256, 3) void scope_estimate2_kernel(ScopeArgs a) {
    EST_T0
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ RedSlots<4> slots;
    __shared__ v2f tw2_lds[256];
    const unsigned tid = threadIdx.x;
    const uint32_t s = blockIdx.x, blk = blockIdx.y, view = a.est_views[blockIdx.z];
    const bool ragged = a.blocks_v != nullptr;
    ScopeEstimate* out = a.estimates + ((uint64_t)s * a.n_blocks + blk) * kScopeTraces + view;
    if (ragged && blk >= a.blocks_v[s]) return;
    const uint32_t block_frames_s = a.frames_v != nullptr ? a.frames_v[s] : a.block_frames;  // chunk calls: the stream's own block length
    // which captures the trigger pass will attempt after this block (:683-700), from the deque lengths alone
    const bool reset_stream = ragged && a.reset_v != nullptr && a.reset_v[s] != 0;
    auto on = [&](int t) { return t < 2 ? a.trace_channel[t] != OMX_CHANNEL_NONE : a.separate_source != 0; };
    auto len0 = [&](int t) -> uint64_t { return ragged ? (reset_stream ? 0ull : a.pos_v[((uint64_t)s * kScopeTraces + t) * 2 + 1]) : a.len[t]; };
    auto len_after = [&](int t) -> uint64_t {
        return on(t) ? min(len0(t) + (uint64_t)(blk + 1) * block_frames_s, (uint64_t)a.history_frames) : 0ull;
    };
    const int linked_view = a.matching_trace >= 0 ? a.matching_trace : (a.separate_source ? 2 : -1);
    bool needed = false;
    if (a.trigger_mode != OMX_TRIGGER_ZERO_CROSSING && on((int)view)) {
        const bool linked_runs = linked_view >= 0 && len_after(linked_view) >= a.base_frames;
        if ((int)view == linked_view) needed = linked_runs;
        else if (view < 2) needed = !linked_runs && len_after((int)view) >= a.base_frames;
    }
    const ScopeEstimate none{0, 0.0f, 0.0f, 0.0f};
    if (!needed) {
        if (tid == 0) *out = none;
        return;
    }
    Reducer<4> red{&slots, 0};
    v2f* fft = reinterpret_cast<v2f*>(smem_raw);
    float* E = reinterpret_cast<float*>(smem_raw);           // energy prefix, dead before the transform starts
    float* D = reinterpret_cast<float*>(smem_raw) + 2 * FFT4096_LDS;  // denominators, then the NSDF
    ScopeTwiddles tw;
    tw.j = tid;
    tw.tw3_global = a.tw4096;
    tw.tw2 = tw2_lds;
#pragma unroll
    for (int t = 1; t < 16; ++t) tw.tw3[t - 1] = a.tw4096[tid * (unsigned)t];
    tw2_lds[tid] = a.tw256[tid];

    const uint64_t n_trace = len_after((int)view);
    const uint64_t head0 = ragged ? a.pos_v[((uint64_t)s * kScopeTraces + view) * 2] : a.head[view];
    const uint64_t head = head0 + (uint64_t)(blk + 1) * block_frames_s;
    const uint32_t n = (uint32_t)min((uint64_t)a.probe_frames, n_trace);
    const float* ring = a.rings + ((uint64_t)s * kScopeTraces + view) * a.cap;
    const uint32_t mask = (uint32_t)(a.cap - 1), start = (uint32_t)((head - n) & (a.cap - 1));  // 32-bit index arithmetic per load
    float last_peak = 0.0f;
    if (n < 3) {  // (:308-313: the estimator is not run, last_peak = 0)
        if (tid == 0) *out = none;
        return;
    }
    const int j = (int)tid;
    v2f v[16];
    float r3[3] = {0.0f, NEG_INF, -NEG_INF};
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t i0 = 2u * (uint32_t)(j + 256 * t);
        // (unconditional loads — the index is masked into the ring whatever i0 is — and a select: a load under a per-lane condition
        // is an exec-mask region of its own, 32 of them here)
        const float r0 = ring[(start + i0) & mask], r1 = ring[(start + i0 + 1u) & mask];
        v[t] = v2f{i0 < n ? r0 : 0.0f, i0 + 1u < n ? r1 : 0.0f};
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t i0 = 2u * (uint32_t)(j + 256 * t);
        if (i0 < n) {
            r3[0] += v[t].x;
            r3[1] = fmaxf(r3[1], v[t].x);
            r3[2] = fminf(r3[2], v[t].x);
        }
        if (i0 + 1u < n) {
            r3[0] += v[t].y;
            r3[1] = fmaxf(r3[1], v[t].y);
            r3[2] = fminf(r3[2], v[t].y);
        }
    }
    red.run<3, (OP_SUM) | (OP_MAX << 2) | (OP_MIN << 4)>(r3);
    EST_MARK(0)
    const float mean = r3[0] / (float)n;
    last_peak = fmaxf(fabsf(r3[1] - mean), fabsf(r3[2] - mean));  // max |x - mean| (:98-101): attained at an extreme of x
    const float rate = a.sample_rate;
    const uint32_t min_period = f2u(fmaxf(roundf(rate / MAX_HZ), 2.0f));
    const uint32_t max_period = min(f2u(roundf(rate / MIN_HZ)), n / 2);
    if (last_peak < MIN_SIGNAL_PEAK || max_period <= min_period + 1) {
        if (tid == 0) *out = ScopeEstimate{0, 0.0f, 0.0f, last_peak};
        return;
    }
    const uint32_t max_lag = max_period;
    // compute_periodicity (:133-181); the 8192-point transform size is what routes a configuration to this kernel
    // centred samples (the transform's input) and their squares in index order
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t i0 = 2u * (uint32_t)(j + 256 * t);
        v[t].x = i0 < n ? v[t].x - mean : 0.0f;
        v[t].y = i0 + 1u < n ? v[t].y - mean : 0.0f;
        *reinterpret_cast<v2f*>(E + i0) = v2f{v[t].x * v[t].x, v[t].y * v[t].y};  // E[i] <- c_i^2 for now (i0 + 1 <= 8191 < 2 FFT4096_LDS)
    }
    __syncthreads();
    {   // E[i + 1] = c_i^2 + E[i] (:141-146): per-thread chunks, a wave scan of the chunk sums, the wavefronts in order; in place
        // (the squares sit one slot below their prefix: every chunk is in registers before anything is overwritten)
        const uint32_t chunk = (n + 255) / 256;  // <= 32 (n <= 8192); 19 at the 4800-sample probe of 48 kHz
        const uint32_t lo = min(tid * chunk, n), hi = min(lo + chunk, n);
        auto scan = [&](auto ctag) {
            constexpr int C = decltype(ctag)::value;
            float sq[C];
            float local = 0.0f;
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const uint32_t i = lo + (uint32_t)q;
                const float e = E[i];  // (i <= n + 32: inside the buffer; read unconditionally, selected below)
                sq[q] = (uint32_t)q < chunk && i < hi ? e : 0.0f;
                local = sq[q] + local;
            }
            const float incl = wave_scan<OP_SUM>(local);  // inclusive over the wavefront
            const unsigned lane = tid & 63u, wave = wave_index();
            if (lane == 63) slots.f[red.phase][0][wave] = incl;
            __syncthreads();  // (also: every chunk is in registers)
            float base = incl - local;
            for (unsigned w = 0; w < wave; ++w) base += slots.f[red.phase][0][w];
            red.phase ^= 1;
            if (tid == 0) E[0] = 0.0f;
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const uint32_t i = lo + (uint32_t)q;
                if ((uint32_t)q < chunk && i < hi) {
                    base = sq[q] + base;
                    E[1 + i] = base;
                }
            }
        };
        if (chunk <= 20) scan(std::integral_constant<int, 20>{});
        else scan(std::integral_constant<int, 32>{});
    }
    __syncthreads();
    EST_MARK(1)
    const float total_energy = E[n];
    for (uint32_t tau = tid; tau <= max_lag; tau += 256) D[tau] = E[n - tau] + (total_energy - E[tau]);
    if (total_energy <= F32_EPS) {  // (:168)
        if (tid == 0) *out = ScopeEstimate{0, 0.0f, 0.0f, last_peak};
        return;
    }
    __syncthreads();  // E is dead: the transform buffer takes its place
    EST_MARK(2)
    // Autocorrelation of the zero-padded real probe through two 4096-point transforms (:147-160 computes FFT_8192(x + 0i),
    // |.|^2, IFFT_8192, real part):  z[m] = x[2m] + i x[2m+1];  Z = FFT_4096(z);  E, O = even / odd sample spectra;
    //   X[k] = E + w^k O, X[k+N] = E - w^k O;  P = |X|^2 (real, P[2N-k] = P[k]);
    //   acf[2m] + i acf[2m+1] = IFFT_4096( (P[k] + P[k+N]) + i (P[k] - P[k+N]) conj(w^k) )
    constexpr uint32_t N = 4096;
    v2f w8[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) w8[t] = a.tw_fft[(uint32_t)(j + 256 * t)];  // exp(-2 pi i k / 8192), wanted after the forward transform
    fft4096t<false, false>(v, fft, fft, j, tw);
    __syncthreads();
    EST_MARK(3)
#pragma unroll
    for (int t = 0; t < 16; ++t) fft[pad16(j + 256 * t)] = v[t];
    __syncthreads();
    v2f y[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t k = (uint32_t)(j + 256 * t);
        const v2f z = v[t];
        const v2f zr = fft[pad16((int)((N - k) & (N - 1)))];
        // (two-element vector arithmetic: v_pk_add / v_pk_mul / v_pk_fma halve the VALU count of this step)
        const v2f zc{zr.x, -zr.y};                       // conj Zr
        const v2f e = (z + zc) * 0.5f;                    // (Z + conj Zr) / 2
        const v2f d = (z - zc) * 0.5f;                    // (Z - conj Zr) / 2
        const v2f o{d.y, -d.x};                           // ... / i
        const v2f w = w8[t];
        const v2f wo = cmul(o, w);                        // w^k O
        const v2f xp = e + wo, xm = e - wo;               // X[k], X[k + N]
        const v2f sqp = xp * xp, sqm = xm * xm;
        const float p0 = sqp.x + sqp.y, p1 = sqm.x + sqm.y;  // P[k], P[k + N]
        const float sum = p0 + p1, dif = p0 - p1;
        y[t] = v2f{sum + dif * w.y, dif * w.x};
    }
    __syncthreads();
    EST_MARK(4)
    fft4096t<true, false>(y, fft, fft, j, tw);  // y[t] = (acf[2m], acf[2m + 1]), m = j + 256 t
    EST_MARK(5)
    const float norm = 1.0f / 8192.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t m = (uint32_t)(j + 256 * t);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t tau = 2u * m + (uint32_t)h;
            if (tau > max_lag) continue;
            const float acf = h ? y[t].y : y[t].x;
            const float denom = D[tau];
            D[tau] = denom > F32_EPS ? 2.0f * acf * norm / denom : 0.0f;
        }
    }
    __syncthreads();
    EST_MARK(6)
    const float* nsdf = D;
    // first tau >= 1 with nsdf <= 0 (:110)
    uint32_t zc = 0xFFFFFFFFu;
    for (uint32_t tau = 1 + tid; tau <= max_period; tau += 256)
        if (nsdf[tau] <= 0.0f) {
            zc = tau;
            break;
        }
    zc = red.run_u32<false>(zc);
    EST_MARK(7)
    ScopeEstimate res{0, 0.0f, 0.0f, last_peak};
    const uint32_t first_tau = max(min_period, zc);
    if (zc != 0xFFFFFFFFu && first_tau < max_period) {
        auto is_candidate = [&](uint32_t tau) {
            return nsdf[tau] >= MIN_PERIODICITY && nsdf[tau] >= nsdf[tau - 1] && nsdf[tau] >= nsdf[tau + 1];
        };
        // max_by(total_cmp) keeps the LAST maximum (:119-121): candidates are >= 0.5, so their bit patterns order like their
        // values; reduce the key first, then the largest tau that carries it
        uint32_t bestk = 0u, besttau = 0u;
        for (uint32_t tau = first_tau + tid; tau < max_period; tau += 256)
            if (is_candidate(tau)) {
                const uint32_t k = total_order_key(nsdf[tau]);
                if (k >= bestk) {
                    bestk = k;
                    besttau = tau;
                }
            }
        const uint32_t kmax = red.run_u32<true>(bestk);
        const uint32_t best = red.run_u32<true>(bestk == kmax && kmax != 0u ? besttau : 0u);
        if (kmax != 0u) {
            const float cutoff = nsdf[best] * PEAK_CUTOFF;
            uint32_t peak = 0xFFFFFFFFu;
            for (uint32_t tau = first_tau + tid; tau <= best; tau += 256)
                if (is_candidate(tau) && nsdf[tau] >= cutoff) {
                    peak = tau;
                    break;
                }
            peak = red.run_u32<false>(peak);
            if (peak == 0xFFFFFFFFu) peak = best;
            res.some = 1;
            res.period = parabolic_refine(nsdf[peak - 1], nsdf[peak], nsdf[peak + 1], peak);
            res.confidence = rclamp(nsdf[peak], 0.0f, 1.0f);
        }
    }
    EST_MARK(8)
    EST_REPORT
    if (tid == 0) *out = res;
}


// ---------------------------------------------------------------- scope_estimate_big_kernel (round 4)
// The same estimate for the rates whose autocorrelation is a 16 384- or 32 768-point transform (54.6 ... 109 kHz: 88.2 / 96 kHz;
// 109 ... 218 kHz: 176.4 / 192 kHz) — until round 3 these configurations ran a radix-2 transform in GLOBAL memory inside the
// one-workgroup-per-stream kernel (96 kHz: 27.9 ms per 256 x 64-block call against 1.3 ms at 48 kHz).  M = fft_size / 2 complex
// points per (stream, block, view) as one size-templated LDS transform each way (fft_pow2_device.hpp: T = M / 16 threads,
// four radix-16 passes in place), real-input packing as in scope_estimate2_kernel.  The NSDF denominators of a thread's own lags
// wait in registers (the energy prefix shares the transform buffer and is dead before the transform starts; the NSDF takes the
// buffer over after the inverse), so the kernel needs the padded M-point buffer and nothing else: 70 KiB / 139 KiB.
template <int LOGM>
__global__ __launch_bounds__(FftGeom<LOGM>::T) void scope_estimate_big_kernel(ScopeArgs a) {
    using G = FftGeom<LOGM>;
    constexpr int M = G::N, T = G::T, W = T / 64;
    static_assert(G::FRAMES == 1 && (LOGM == 13 || LOGM == 14), "8192 / 16384 complex points, one transform per workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ RedSlots<W> slots;
    __shared__ v2f tw2_lds[256];
    const unsigned tid = threadIdx.x;
    const uint32_t s = blockIdx.x, blk = blockIdx.y, view = a.est_views[blockIdx.z];
    const bool ragged = a.blocks_v != nullptr;
    ScopeEstimate* out = a.estimates + ((uint64_t)s * a.n_blocks + blk) * kScopeTraces + view;
    if (ragged && blk >= a.blocks_v[s]) return;
    const uint32_t block_frames_s = a.frames_v != nullptr ? a.frames_v[s] : a.block_frames;  // chunk calls: the stream's own block length
    // which captures the trigger pass will attempt after this block (:683-700), from the deque lengths alone
    const bool reset_stream = ragged && a.reset_v != nullptr && a.reset_v[s] != 0;
    auto on = [&](int t) { return t < 2 ? a.trace_channel[t] != OMX_CHANNEL_NONE : a.separate_source != 0; };
    auto len0 = [&](int t) -> uint64_t { return ragged ? (reset_stream ? 0ull : a.pos_v[((uint64_t)s * kScopeTraces + t) * 2 + 1]) : a.len[t]; };
    auto len_after = [&](int t) -> uint64_t {
        return on(t) ? min(len0(t) + (uint64_t)(blk + 1) * block_frames_s, (uint64_t)a.history_frames) : 0ull;
    };
    const int linked_view = a.matching_trace >= 0 ? a.matching_trace : (a.separate_source ? 2 : -1);
    bool needed = false;
    if (a.trigger_mode != OMX_TRIGGER_ZERO_CROSSING && on((int)view)) {
        const bool linked_runs = linked_view >= 0 && len_after(linked_view) >= a.base_frames;
        if ((int)view == linked_view) needed = linked_runs;
        else if (view < 2) needed = !linked_runs && len_after((int)view) >= a.base_frames;
    }
    const ScopeEstimate none{0, 0.0f, 0.0f, 0.0f};
    if (!needed) {
        if (tid == 0) *out = none;
        return;
    }
    Reducer<W> red{&slots, 0};
    v2f* fft = reinterpret_cast<v2f*>(smem_raw);
    float* E = reinterpret_cast<float*>(smem_raw);  // energy prefix (n + 1 <= 2 M + 1 floats), dead before the transform starts; then the NSDF
    TwiddlesPow2<LOGM> tw;
    tw.tw2 = tw2_lds;
    tw.load(a.tw4096, tid);  // `tw4096` carries exp(-2 pi i k / M) for this M
    if (tid < 256) tw2_lds[tid] = a.tw256[tid];

    const uint64_t n_trace = len_after((int)view);
    const uint64_t head0 = ragged ? a.pos_v[((uint64_t)s * kScopeTraces + view) * 2] : a.head[view];
    const uint64_t head = head0 + (uint64_t)(blk + 1) * block_frames_s;
    const uint32_t n = (uint32_t)min((uint64_t)a.probe_frames, n_trace);
    const float* ring = a.rings + ((uint64_t)s * kScopeTraces + view) * a.cap;
    const uint32_t mask = (uint32_t)(a.cap - 1), start = (uint32_t)((head - n) & (a.cap - 1));
    float last_peak = 0.0f;
    if (n < 3) {  // (:308-313: the estimator is not run, last_peak = 0)
        if (tid == 0) *out = none;
        return;
    }
    const int j = (int)tid;
    v2f v[16];
    float r3[3] = {0.0f, NEG_INF, -NEG_INF};
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t i0 = 2u * (uint32_t)(j + T * t);
        const float x0 = i0 < n ? ring[(start + i0) & mask] : 0.0f;
        const float x1 = i0 + 1u < n ? ring[(start + i0 + 1u) & mask] : 0.0f;
        v[t] = v2f{x0, x1};
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t i0 = 2u * (uint32_t)(j + T * t);
        if (i0 < n) {
            r3[0] += v[t].x;
            r3[1] = fmaxf(r3[1], v[t].x);
            r3[2] = fminf(r3[2], v[t].x);
        }
        if (i0 + 1u < n) {
            r3[0] += v[t].y;
            r3[1] = fmaxf(r3[1], v[t].y);
            r3[2] = fminf(r3[2], v[t].y);
        }
    }
    red.template run<3, (OP_SUM) | (OP_MAX << 2) | (OP_MIN << 4)>(r3);
    const float mean = r3[0] / (float)n;
    last_peak = fmaxf(fabsf(r3[1] - mean), fabsf(r3[2] - mean));  // max |x - mean| (:98-101): attained at an extreme of x
    const float rate = a.sample_rate;
    const uint32_t min_period = f2u(fmaxf(roundf(rate / MAX_HZ), 2.0f));
    const uint32_t max_period = min(f2u(roundf(rate / MIN_HZ)), n / 2);
    if (last_peak < MIN_SIGNAL_PEAK || max_period <= min_period + 1) {
        if (tid == 0) *out = ScopeEstimate{0, 0.0f, 0.0f, last_peak};
        return;
    }
    const uint32_t max_lag = max_period;
    // compute_periodicity (:133-181): centred samples (the transform's input) and their squares in index order
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t i0 = 2u * (uint32_t)(j + T * t);
        v[t].x = i0 < n ? v[t].x - mean : 0.0f;
        v[t].y = i0 + 1u < n ? v[t].y - mean : 0.0f;
        *reinterpret_cast<v2f*>(E + i0) = v2f{v[t].x * v[t].x, v[t].y * v[t].y};  // E[i] <- c_i^2 for now
    }
    __syncthreads();
    {   // E[i + 1] = c_i^2 + E[i] (:141-146): per-thread chunks, a wave scan of the chunk sums, the wavefronts in order; in place
        const uint32_t chunk = (n + (uint32_t)T - 1u) / (uint32_t)T;  // <= 32 (n <= 2 M)
        const uint32_t lo = min(tid * chunk, n), hi = min(lo + chunk, n);
        auto scan = [&](auto ctag) {
            constexpr int C = decltype(ctag)::value;
            float sq[C];
            float local = 0.0f;
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const uint32_t i = lo + (uint32_t)q;
                sq[q] = (uint32_t)q < chunk && i < hi ? E[i] : 0.0f;
                local = sq[q] + local;
            }
            const float incl = wave_scan<OP_SUM>(local);  // inclusive over the wavefront
            const unsigned lane = tid & 63u, wave = wave_index();
            if (lane == 63) slots.f[red.phase][0][wave] = incl;
            __syncthreads();  // (also: every chunk is in registers)
            float base = incl - local;
            for (unsigned w = 0; w < wave; ++w) base += slots.f[red.phase][0][w];
            red.phase ^= 1;
            if (tid == 0) E[0] = 0.0f;
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const uint32_t i = lo + (uint32_t)q;
                if ((uint32_t)q < chunk && i < hi) {
                    base = sq[q] + base;
                    E[1 + i] = base;
                }
            }
        };
        if (chunk <= 20) scan(std::integral_constant<int, 20>{});
        else scan(std::integral_constant<int, 32>{});
    }
    __syncthreads();
    const float total_energy = E[n];
    // the denominators of this thread's own lags tau = 2 (j + T t) + h <= max_lag <= M: t <= M / (2 T) = 8
    constexpr int TD = 9;
    float den[TD][2];
#pragma unroll
    for (int t = 0; t < TD; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t tau = 2u * (uint32_t)(j + T * t) + (uint32_t)h;
            den[t][h] = tau <= max_lag ? E[n - tau] + (total_energy - E[tau]) : 0.0f;
        }
    if (total_energy <= F32_EPS) {  // (:168)
        if (tid == 0) *out = ScopeEstimate{0, 0.0f, 0.0f, last_peak};
        return;
    }
    __syncthreads();  // E is dead: the transform buffer takes its place
    // autocorrelation of the zero-padded real probe through two M-point transforms (derivation: scope_estimate2_kernel)
    fftp_inplace<false, LOGM>(v, fft, j, tw);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) fft[pad16(j + T * t)] = v[t];
    __syncthreads();
    v2f y[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t k = (uint32_t)(j + T * t);
        const v2f z = v[t];
        const v2f zr = fft[pad16((int)(((uint32_t)M - k) & ((uint32_t)M - 1u)))];
        const v2f zc{zr.x, -zr.y};                       // conj Zr
        const v2f e = (z + zc) * 0.5f;                    // (Z + conj Zr) / 2
        const v2f d = (z - zc) * 0.5f;                    // (Z - conj Zr) / 2
        const v2f o{d.y, -d.x};                           // ... / i
        const v2f w = a.tw_fft[k];                        // exp(-2 pi i k / 2M)
        const v2f wo = cmul(o, w);                        // w^k O
        const v2f xp = e + wo, xm = e - wo;               // X[k], X[k + M]
        const v2f sqp = xp * xp, sqm = xm * xm;
        const float p0 = sqp.x + sqp.y, p1 = sqm.x + sqm.y;  // P[k], P[k + M]
        const float sum = p0 + p1, dif = p0 - p1;
        y[t] = v2f{sum + dif * w.y, dif * w.x};
    }
    __syncthreads();
    fftp_inplace<true, LOGM>(y, fft, j, tw);  // y[t] = (acf[2m], acf[2m + 1]), m = j + T t
    __syncthreads();  // the last pass still reads the buffer: the NSDF takes it over
    float* nsdf_w = reinterpret_cast<float*>(smem_raw);
    const float norm = 1.0f / (float)(2 * M);
#pragma unroll
    for (int t = 0; t < TD; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t tau = 2u * (uint32_t)(j + T * t) + (uint32_t)h;
            if (tau > max_lag) continue;
            const float acf = h ? y[t].y : y[t].x;
            nsdf_w[tau] = den[t][h] > F32_EPS ? 2.0f * acf * norm / den[t][h] : 0.0f;
        }
    __syncthreads();
    const float* nsdf = nsdf_w;
    // first tau >= 1 with nsdf <= 0 (:110)
    uint32_t zc = 0xFFFFFFFFu;
    for (uint32_t tau = 1 + tid; tau <= max_period; tau += T)
        if (nsdf[tau] <= 0.0f) {
            zc = tau;
            break;
        }
    zc = red.template run_u32<false>(zc);
    ScopeEstimate res{0, 0.0f, 0.0f, last_peak};
    const uint32_t first_tau = max(min_period, zc);
    if (zc != 0xFFFFFFFFu && first_tau < max_period) {
        auto is_candidate = [&](uint32_t tau) {
            return nsdf[tau] >= MIN_PERIODICITY && nsdf[tau] >= nsdf[tau - 1] && nsdf[tau] >= nsdf[tau + 1];
        };
        uint32_t bestk = 0u, besttau = 0u;
        for (uint32_t tau = first_tau + tid; tau < max_period; tau += T)
            if (is_candidate(tau)) {
                const uint32_t k = total_order_key(nsdf[tau]);
                if (k >= bestk) {
                    bestk = k;
                    besttau = tau;
                }
            }
        const uint32_t kmax = red.template run_u32<true>(bestk);
        const uint32_t best = red.template run_u32<true>(bestk == kmax && kmax != 0u ? besttau : 0u);
        if (kmax != 0u) {
            const float cutoff = nsdf[best] * PEAK_CUTOFF;
            uint32_t peak = 0xFFFFFFFFu;
            for (uint32_t tau = first_tau + tid; tau <= best; tau += T)
                if (is_candidate(tau) && nsdf[tau] >= cutoff) {
                    peak = tau;
                    break;
                }
            peak = red.template run_u32<false>(peak);
            if (peak == 0xFFFFFFFFu) peak = best;
            res.some = 1;
            res.period = parabolic_refine(nsdf[peak - 1], nsdf[peak], nsdf[peak + 1], peak);
            res.confidence = rclamp(nsdf[peak], 0.0f, 1.0f);
        }
    }
    if (tid == 0) *out = res;
}

// ring re-homing on growth: the newest `history` samples of every trace keep their absolute positions, only the modulus changes
__global__ __launch_bounds__(256) void scope_rehome_kernel(const float* from, uint64_t from_cap, float* to, uint64_t to_cap,
                                                           const uint64_t* pos_v, ScopeArgs a) {
    const uint32_t s = blockIdx.y, t = blockIdx.z;
    const uint64_t head = pos_v ? pos_v[((uint64_t)s * kScopeTraces + t) * 2] : a.head[t];
    const uint64_t len = pos_v ? pos_v[((uint64_t)s * kScopeTraces + t) * 2 + 1] : a.len[t];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    const uint64_t p = head - len + i;
    to[((uint64_t)s * kScopeTraces + t) * to_cap + (p & (to_cap - 1))] = from[((uint64_t)s * kScopeTraces + t) * from_cap + (p & (from_cap - 1))];
}

void scope_fast_phase_cycles(unsigned long long out[SCOPE_PHASES], bool reset) {
    OMX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fast_phase_cycles), SCOPE_PHASES * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long zero[SCOPE_PHASES] = {};
        OMX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fast_phase_cycles), zero, sizeof(zero)));
    }
}

// smallest LDS the trigger kernel runs in (reference + the largest span twice + one template copy); the launch asks for room
// for the aligned template copies on top, up to what a CU has
uint64_t scope_trigger_lds_bytes(uint32_t max_kernel, uint32_t max_period) {
    const TriggerLayout l = trigger_layout(max_kernel, max_period);
    return (uint64_t)(l.ref + l.dyn_min) * sizeof(float);
}

void launch_scope_rehome(const float* from, uint64_t from_cap, float* to, uint64_t to_cap, const uint64_t* pos_v, const ScopeArgs& a,
                         uint64_t max_len, hipStream_t stream) {
    if (a.n_streams == 0 || max_len == 0) return;
    hipLaunchKernelGGL(scope_rehome_kernel, dim3((uint32_t)((max_len + 255) / 256), a.n_streams, kScopeTraces), dim3(256), 0, stream, from,
                       from_cap, to, to_cap, pos_v, a);
}

void launch_oscilloscope_fast(const ScopeArgs& a, hipStream_t stream) {
    if (a.n_streams == 0 || a.n_blocks == 0) return;
    hipLaunchKernelGGL(scope_push2_kernel, dim3((uint32_t)((a.frames_total + 255) / 256), a.n_streams), dim3(256), 0, stream, a);
    static std::once_flag attr_once;  // (two host threads may race on the first launch; one device per process, omx.h)
    static int threads = 512;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scope_estimate2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scope_trigger_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scope_trigger_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scope_trigger_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (const char* e = tuning_env("OMX_SCOPE_THREADS")) {  // tuning hook
            const int t = atoi(e);
            threads = t == 1024 || t == 256 ? t : 512;
        }
    });
    if (a.trigger_mode != OMX_TRIGGER_ZERO_CROSSING && a.est_view_count) {
        const size_t lds_est = (size_t)(2ull * FFT4096_LDS + a.max_period + 8ull) * sizeof(float);
        hipLaunchKernelGGL(scope_estimate2_kernel, dim3(a.n_streams, a.n_blocks, a.est_view_count), dim3(256), lds_est, stream, a);
    }
    const TriggerLayout l = trigger_layout(a.max_kernel, a.max_period);
    const uint32_t len4 = (a.max_kernel + 12 + 3) & ~3u;
    uint32_t lds_floats = std::min<uint32_t>(l.ref + l.dyn_min + 3 * len4, 152 * 1024 / sizeof(float));  // + ~5 KiB of static LDS <= 160 KiB
    if (a.resume_blk != nullptr) {
        // capped form (see launch_oscilloscope_big): 64 KiB instead of the 105 ... 150 KiB worst case — the kernel is max(40 ms, two periods)
        // long, 1920 samples at 48 kHz for everything above 50 Hz (5 len + 36 floats = 38 KiB; the cap admits periods up to 1634 samples,
        // 29 Hz) — so two trigger workgroups share a CU, or one leaves room for the other meters' kernels; blocks that do not fit are
        // handed to the one-workgroup-per-stream kernel
        ScopeArgs w = a;
        lds_floats = 64 * 1024 / sizeof(float);
        w.ref_cap = std::min<uint32_t>(a.max_kernel, (lds_floats - 36u) / 5u);
        hipLaunchKernelGGL(scope_trigger_kernel<512>, dim3(a.n_streams), dim3(512), (size_t)lds_floats * sizeof(float), stream, w, lds_floats);
        ScopeArgs r = a;
        r.resume_mode = 1;
        r.pre_pushed = 1;
        r.fft_global = nullptr;
        r.lds_scratch = scope_locate_lds_bytes(a.max_kernel, a.max_period) <= 150 * 1024 ? 2u : 0u;
        launch_oscilloscope(r, stream);
        return;
    }
    if (threads == 256)
        hipLaunchKernelGGL(scope_trigger_kernel<256>, dim3(a.n_streams), dim3(256), (size_t)lds_floats * sizeof(float), stream, a, lds_floats);
    else if (threads == 1024)
        hipLaunchKernelGGL(scope_trigger_kernel<1024>, dim3(a.n_streams), dim3(1024), (size_t)lds_floats * sizeof(float), stream, a, lds_floats);
    else
        hipLaunchKernelGGL(scope_trigger_kernel<512>, dim3(a.n_streams), dim3(512), (size_t)lds_floats * sizeof(float), stream, a, lds_floats);
}


// fft_size 16384 / 32768 (88.2 ... 192 kHz): every block pushed first, the period estimates of all (stream, block, view) by
// scope_estimate_big_kernel, then the one-workgroup-per-stream kernel of round 1 (oscilloscope_kernels.hip) with pre_pushed = 1: it
// takes its estimates from a.estimates and runs only the stateful part
void launch_oscilloscope_big(const ScopeArgs& a, hipStream_t stream) {
    if (a.n_streams == 0 || a.n_blocks == 0) return;
    hipLaunchKernelGGL(scope_push2_kernel, dim3((uint32_t)((a.frames_total + 255) / 256), a.n_streams), dim3(256), 0, stream, a);
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scope_estimate_big_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(FftGeom<13>::LDS * sizeof(v2f)));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scope_estimate_big_kernel<14>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(FftGeom<14>::LDS * sizeof(v2f)));
    });
    if (a.trigger_mode != OMX_TRIGGER_ZERO_CROSSING && a.est_view_count) {
        if (a.fft_size == 16384)
            hipLaunchKernelGGL(scope_estimate_big_kernel<13>, dim3(a.n_streams, a.n_blocks, a.est_view_count), dim3(FftGeom<13>::T),
                               FftGeom<13>::LDS * sizeof(v2f), stream, a);
        else
            hipLaunchKernelGGL(scope_estimate_big_kernel<14>, dim3(a.n_streams, a.n_blocks, a.est_view_count), dim3(FftGeom<14>::T),
                               FftGeom<14>::LDS * sizeof(v2f), stream, a);
    }
    if (a.resume_blk == nullptr) {  // (round-4 first form, kept for A/B: the one-workgroup-per-stream kernel runs every block)
        launch_oscilloscope(a, stream);
        return;
    }
    // the wide trigger pass on 152 KiB of LDS: the kernel is max(40 ms, two periods) long (:184-189), so at 192 kHz every signal above
    // 50 Hz fits (5 len + 36 floats: reference, raw span, mean-removed span, template); what does not is handed over block by block
    static std::once_flag trig_once;
    std::call_once(trig_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scope_trigger_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    });
    ScopeArgs w = a;
    const uint32_t lds_floats = 152 * 1024 / sizeof(float);
    w.ref_cap = std::min<uint32_t>(a.max_kernel, (lds_floats - 36u) / 5u);
    hipLaunchKernelGGL(scope_trigger_kernel<512>, dim3(a.n_streams), dim3(512), (size_t)lds_floats * sizeof(float), stream, w, lds_floats);
    ScopeArgs r = a;
    r.resume_mode = 1;
    launch_oscilloscope(r, stream);
}

}  // namespace omx
