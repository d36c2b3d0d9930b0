// Wavefront reductions on the DPP path (gfx950): row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, then row_bcast 15 / 31 — six VALU
// instructions with the data movement riding the operand, against six ds_bpermute round trips through the LDS pipeline for the
// __shfl_xor butterfly.  Lane 63 holds the wavefront's reduction; `all` hands it to every lane through an SGPR.
#pragma once
#include <hip/hip_runtime.h>

namespace omx {

// Workgroup barrier for kernels whose threads exchange data through LDS only: waits for this wavefront's LDS traffic and leaves its global
// loads and stores in flight.  __syncthreads() also drains vmcnt at each call — the next tile's prefetch issued just before it, every ring
// / row / point store since the last one.
__device__ __forceinline__ void lds_workgroup_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

namespace wave {

// max / min as the bare instructions: fmaxf / fminf make hipcc quiet every operand it cannot prove canonical first (v_max_f32 v, v, v
// per loaded value or packed-arithmetic result — the spectrum kernel carried 60 of those).  The instructions themselves already
// return the other operand for a NaN and quiet a signalling one (IEEE mode is on).
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
enum { SUM = 0, MAX = 1, MIN = 2 };
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp(float old, float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
template <int OP>
__device__ __forceinline__ float op(float a, float b) {
    return OP == SUM ? a + b : (OP == MAX ? vmax(a, b) : vmin(a, b));
}
template <int OP>
__device__ __forceinline__ float scan(float x) {  // inclusive, lane order
    x = op<OP>(x, dpp<0x111, 0xf>(OP == SUM ? 0.0f : x, x));
    x = op<OP>(x, dpp<0x112, 0xf>(OP == SUM ? 0.0f : x, x));
    x = op<OP>(x, dpp<0x114, 0xf>(OP == SUM ? 0.0f : x, x));
    x = op<OP>(x, dpp<0x118, 0xf>(OP == SUM ? 0.0f : x, x));
    x = op<OP>(x, dpp<0x142, 0xa>(OP == SUM ? 0.0f : x, x));
    x = op<OP>(x, dpp<0x143, 0xc>(OP == SUM ? 0.0f : x, x));
    return x;
}
template <int OP>
__device__ __forceinline__ float all(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, scan<OP>(x)), 63));
}
// The six reductions a hop pair needs (two sums, two maxima, two minima) as ONE interleaved DPP sequence: 36 instructions, one per
// value and step.  Written as assembly because the fused form (v_max_f32_dpp in place: a lane whose DPP source does not exist keeps
// its value) is only chosen by hipcc behind fmaxf's canonicalisation, and the bare-instruction max above costs a copy, a DPP move
// and hazard nops per step instead.  Each value is touched once per group of six instructions, which covers the two wait states a
// DPP read needs after a VALU write of the same register; the leading s_nop covers the producer of the inputs.  Lane 63 holds the results.
#define OMX_DPP6(step)                                          \
    "v_add_f32_dpp %0, %0, %0 " step "\n v_add_f32_dpp %1, %1, %1 " step "\n" \
    "v_max_f32_dpp %2, %2, %2 " step "\n v_max_f32_dpp %3, %3, %3 " step "\n" \
    "v_min_f32_dpp %4, %4, %4 " step "\n v_min_f32_dpp %5, %5, %5 " step "\n"
__device__ __forceinline__ void scan_sum2_max2_min2(float& s0, float& s1, float& hi0, float& hi1, float& lo0, float& lo1) {
    asm("s_nop 1\n"
        OMX_DPP6("row_shr:1 row_mask:0xf bank_mask:0xf")
        OMX_DPP6("row_shr:2 row_mask:0xf bank_mask:0xf")
        OMX_DPP6("row_shr:4 row_mask:0xf bank_mask:0xf")
        OMX_DPP6("row_shr:8 row_mask:0xf bank_mask:0xf")
        OMX_DPP6("row_bcast:15 row_mask:0xa bank_mask:0xf")
        OMX_DPP6("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(s0), "+v"(s1), "+v"(hi0), "+v"(hi1), "+v"(lo0), "+v"(lo1));
}
#undef OMX_DPP6
// the same for two maxima and two minima (the hop pair's sample ranges; the sums come from window_sum_kernels.hip since round 6).
// Each value is touched once per group of four instructions: the two wait states a DPP read needs after a VALU write are covered.
#define OMX_DPP4(step)                                          \
    "v_max_f32_dpp %0, %0, %0 " step "\n v_max_f32_dpp %1, %1, %1 " step "\n" \
    "v_min_f32_dpp %2, %2, %2 " step "\n v_min_f32_dpp %3, %3, %3 " step "\n"
__device__ __forceinline__ void scan_max2_min2(float& hi0, float& hi1, float& lo0, float& lo1) {
    asm("s_nop 1\n"
        OMX_DPP4("row_shr:1 row_mask:0xf bank_mask:0xf")
        OMX_DPP4("row_shr:2 row_mask:0xf bank_mask:0xf")
        OMX_DPP4("row_shr:4 row_mask:0xf bank_mask:0xf")
        OMX_DPP4("row_shr:8 row_mask:0xf bank_mask:0xf")
        OMX_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
        OMX_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(hi0), "+v"(hi1), "+v"(lo0), "+v"(lo1));
}
#undef OMX_DPP4
__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (uint32_t)(127 + e) << 23); }  // 2^e, -126 <= e <= 127

}  // namespace wave
}  // namespace omx
