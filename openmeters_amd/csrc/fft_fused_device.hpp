// 16-point DFT with the twiddle products folded into the butterflies (gfx950 packed-f32 VALU).
//
// dft16 of fft_device.hpp multiplies, then adds: a radix-4 butterfly on (x0, w1 x1, w2 x2, w3 x3) costs three complex products
// (6 packed ops) and eight packed adds.  Here the products ride the adds as fused multiply-adds:
//     t0 = x0 + w2 x2            two v_pk_fma_f32 (complex multiply-add)
//     t1 = 2 x0 - t0             one v_pk_fma_f32           (= x0 - w2 x2)
//     y1 = w1 x1                 two packed ops
//     t2 = y1 + w3 x3            two v_pk_fma_f32
//     d  = 2 y1 - t2             one v_pk_fma_f32           (= y1 - w3 x3)
// and the four output adds: 12 packed ops where the unfused form takes 14.  The same butterfly serves the w16 twiddles between
// the two radix-4 stages (constants in scalar registers).  Per 16-point DFT: 75 packed ops without outer twiddles (80 unfused),
// 97 with fifteen outer twiddles (110).  Every fused operation rounds once where the unfused form rounds twice, so the result
// is at least as close to exact arithmetic (tests/test_exact_f64.py holds the kernels that use this to the same bars).
#pragma once
#include "fft_device.hpp"

namespace omx {

// c + a * w  (forward)  /  c + a * conj(w)  (inverse), two packed fused multiply-adds
template <bool INV>
__device__ __forceinline__ v2f cmadd(v2f c, v2f a, v2f w) {
    v2f u, r;
    if constexpr (!INV) {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(u) : "v"(a), "v"(w), "v"(c));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(u));
    } else {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(u) : "v"(a), "v"(w), "v"(c));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(u));
    }
    return r;
}
// the same with a wave-uniform (compile-time) w in scalar registers
template <bool INV>
__device__ __forceinline__ v2f cmadd_const(v2f c, v2f a, v2f w) {
    v2f u, r;
    if constexpr (!INV) {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(u) : "v"(a), "s"(w), "v"(c));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "s"(w), "v"(u));
    } else {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(u) : "v"(a), "s"(w), "v"(c));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(w), "v"(u));
    }
    return r;
}
template <bool INV>
__device__ __forceinline__ v2f cmul_k(v2f a, v2f w) {  // a * w / a * conj(w), w in scalar registers
    v2f t, r;
    if constexpr (!INV) {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "s"(w));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    } else {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    }
    return r;
}
// 2 c - t, one packed fused multiply-add (the scalar pair holds 2.0 in its low word; op_sel_hi 0 reads it for both halves)
__device__ __forceinline__ v2f twice_minus(v2f c, v2f t) {
    v2f r;
    const float two = 2.0f;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(c), "s"(v2f{two, two}), "v"(t));
    return r;
}

// radix-4 butterfly on (x0, w1 x1, w2 x2, w3 x3), in place
template <bool INV, bool CONST>
__device__ __forceinline__ void dft4_fused(v2f& x0, v2f& x1, v2f& x2, v2f& x3, v2f w1, v2f w2, v2f w3) {
    v2f t0, y1, t2;
    if constexpr (CONST) {
        t0 = cmadd_const<INV>(x0, x2, w2);
        y1 = cmul_k<INV>(x1, w1);
        t2 = cmadd_const<INV>(y1, x3, w3);
    } else {
        t0 = cmadd<INV>(x0, x2, w2);
        y1 = twmul<INV>(x1, w1);
        t2 = cmadd<INV>(y1, x3, w3);
    }
    const v2f t1 = twice_minus(x0, t0), d = twice_minus(y1, t2);
    x0 = t0 + t2;
    x1 = add_rot<INV>(t1, d);
    x2 = t0 - t2;
    x3 = sub_rot<INV>(t1, d);
}

// second radix-4 stage of the 4 x 4 split with the w16 twiddles folded in: v[4 k1 + n2] carries w16^(n2 k1)
template <bool INV>
__device__ __forceinline__ void dft16_stage2_fused(v2f (&v)[16]) {
    constexpr float C1 = 0.92387953251128675613f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508977173f;  // sin(pi/8)
    constexpr float H = 0.70710678118654752440f;   // sqrt(1/2)
    const v2f w1{C1, -S1}, w2{H, -H}, w3{S1, -C1}, w6{-H, -H}, w9{-C1, S1};
    dft4<INV>(v[0], v[1], v[2], v[3]);
    dft4_fused<INV, true>(v[4], v[5], v[6], v[7], w1, w2, w3);
    {   // k1 = 2: twiddles 1, w^2, w^4 = -+ i, w^6
        const v2f t0 = add_rot<INV>(v[8], v[10]), t1 = sub_rot<INV>(v[8], v[10]);
        const v2f y1 = cmul_k<INV>(v[9], w2);
        const v2f t2 = cmadd_const<INV>(y1, v[11], w6);
        const v2f d = twice_minus(y1, t2);
        v[8] = t0 + t2;
        v[9] = add_rot<INV>(t1, d);
        v[10] = t0 - t2;
        v[11] = sub_rot<INV>(t1, d);
    }
    dft4_fused<INV, true>(v[12], v[13], v[14], v[15], w3, w6, w9);
}

// In-register 16-point DFT, no outer twiddles.  On return X[k] sits in v[DFT16_OUT(k)] (same convention as dft16).
template <bool INV>
__device__ __forceinline__ void dft16_fused(v2f (&v)[16]) {
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    dft16_stage2_fused<INV>(v);
}
// ... of (v[0], w[1] v[1], ..., w[15] v[15]): the fifteen outer twiddles of a Stockham pass folded into the first stage
template <bool INV>
__device__ __forceinline__ void dft16_fused_tw(v2f (&v)[16], const v2f (&w)[16]) {
    dft4_fused<INV, false>(v[0], v[4], v[8], v[12], w[4], w[8], w[12]);
#pragma unroll
    for (int n2 = 1; n2 < 4; ++n2) {
        v[n2] = twmul<INV>(v[n2], w[n2]);
        dft4_fused<INV, false>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2], w[4 + n2], w[8 + n2], w[12 + n2]);
    }
    dft16_stage2_fused<INV>(v);
}

}  // namespace omx
