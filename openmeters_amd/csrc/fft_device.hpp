// Device-side FFT building blocks for gfx950 (CDNA4), hand-written for 64-wide wavefronts.
//
// Two families:
//   * fft4096_*  — the hot path: one 4096-point complex FFT per 256-thread workgroup as three
//     radix-16 Stockham passes (16 complex values per thread in VGPRs, exchanged through LDS with a
//     +1/16 padded layout that keeps ds_read_b64 / ds_write_b64 conflict-free in every pass).
//   * fft_radix2_* — the generic path: any power-of-two size in a caller-supplied buffer (LDS or
//     global scratch), plain radix-2 with the bit-reversal order of the CPU oracle, used for the
//     config shapes the specialised kernels do not cover.
//
// Built with -ffp-contract=off: every fused multiply-add below is spelled `__builtin_fmaf`.
#pragma once
#include <hip/hip_runtime.h>

#include "bluestein_plan.hpp"

namespace omx {

// Barrier of the transforms in this header.  A translation unit whose kernels exchange data between their threads through LDS only defines
// OMX_FRAME_SYNC_LDS_ONLY before including it: the barrier then waits for LDS traffic and leaves global loads / stores in flight
// (__syncthreads() drains vmcnt at every pass).
__device__ __forceinline__ void fft_sync() {
#ifdef OMX_FRAME_SYNC_LDS_ONLY
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

typedef float v2f __attribute__((ext_vector_type(2)));  // (re, im)

// Complex products as two packed-f32 VALU ops.  hipcc lowers the plain C++ form to v_xor + v_mov +
// v_pk_mul + v_pk_fma (it materialises (-w.y, w.x) in registers); the VOP3P op_sel / neg modifiers do
// the swizzle and the sign inside the multiply, halving the instruction count of every twiddle.
// Rounding is identical to  fma(a.x, w.x, -(a.y*w.y)), fma(a.x, w.y, a.y*w.x).
__device__ __forceinline__ v2f cmul(v2f a, v2f w) {  // a * w
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ v2f cmulc(v2f a, v2f w) {  // a * conj(w) = (a.x w.x + a.y w.y, a.y w.x - a.x w.y)
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
template <bool INV>
__device__ __forceinline__ v2f twmul(v2f a, v2f w) {  // forward: a*w ; inverse: a*conj(w)
    return INV ? cmulc(a, w) : cmul(a, w);
}
// t + (-i) d = (t.x + d.y, t.y - d.x)   and   t + (+i) d = (t.x - d.y, t.y + d.x), one packed add each
__device__ __forceinline__ v2f add_mi(v2f t, v2f d) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(t), "v"(d));
    return r;
}
__device__ __forceinline__ v2f add_pi(v2f t, v2f d) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(t), "v"(d));
    return r;
}
template <bool INV>
__device__ __forceinline__ v2f add_rot(v2f t, v2f d) {  // t + rot(d), rot = *(-i) forward, *(+i) inverse
    return INV ? add_pi(t, d) : add_mi(t, d);
}
template <bool INV>
__device__ __forceinline__ v2f sub_rot(v2f t, v2f d) {  // t - rot(d)
    return INV ? add_mi(t, d) : add_pi(t, d);
}

template <bool INV>
__device__ __forceinline__ void dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
    const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
    a0 = t0 + t2;
    a1 = add_rot<INV>(t1, d);
    a2 = t0 - t2;
    a3 = sub_rot<INV>(t1, d);
}
// dft4 whose third input still has to be multiplied by rot (the w16^4 twiddle of the 4x4 split)
template <bool INV>
__device__ __forceinline__ void dft4_rot2(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
    const v2f t0 = add_rot<INV>(a0, a2), t1 = sub_rot<INV>(a0, a2), t2 = a1 + a3, d = a1 - a3;
    a0 = t0 + t2;
    a1 = add_rot<INV>(t1, d);
    a2 = t0 - t2;
    a3 = sub_rot<INV>(t1, d);
}

// In-register 16-point DFT (4x4 Cooley-Tukey) on v[OFF .. OFF+15] of an N-element register array (constant
// indices only, so the array stays in VGPRs).  On return X[k] sits in v[OFF + DFT16_OUT(k)].
#define DFT16_OUT(k) (4 * ((k)&3) + ((k) >> 2))
template <bool INV, int N, int OFF>
__device__ __forceinline__ void dft16_at(v2f (&v)[N]) {
    constexpr float C1 = 0.92387953251128675613f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508977173f;  // sin(pi/8)
    constexpr float H = 0.70710678118654752440f;   // sqrt(1/2)
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4<INV>(v[OFF + n2], v[OFF + 4 + n2], v[OFF + 8 + n2], v[OFF + 12 + n2]);
    // v[4*k1 + n2] *= w16^(n2*k1), w16 = exp(-+ 2*pi*i/16)
    const v2f w1{C1, -S1}, w3{S1, -C1}, w9{-C1, S1}, hh{H, H}, nh{-H, -H};
    v[OFF + 5] = twmul<INV>(v[OFF + 5], w1);                      // k1=1,n2=1
    v[OFF + 6] = add_rot<INV>(v[OFF + 6], v[OFF + 6]) * hh;       // k1=1,n2=2 : w^2 = H(1 -+ i)
    v[OFF + 7] = twmul<INV>(v[OFF + 7], w3);                      // k1=1,n2=3
    v[OFF + 9] = add_rot<INV>(v[OFF + 9], v[OFF + 9]) * hh;       // k1=2,n2=1 : w^2
    // k1=2,n2=2 : w^4 = -+ i, folded into dft4_rot2 below
    v[OFF + 11] = sub_rot<INV>(v[OFF + 11], v[OFF + 11]) * nh;    // k1=2,n2=3 : w^6 = -H(1 +- i)
    v[OFF + 13] = twmul<INV>(v[OFF + 13], w3);                    // k1=3,n2=1
    v[OFF + 14] = sub_rot<INV>(v[OFF + 14], v[OFF + 14]) * nh;    // k1=3,n2=2 : w^6
    v[OFF + 15] = twmul<INV>(v[OFF + 15], w9);                    // k1=3,n2=3
    dft4<INV>(v[OFF + 0], v[OFF + 1], v[OFF + 2], v[OFF + 3]);
    dft4<INV>(v[OFF + 4], v[OFF + 5], v[OFF + 6], v[OFF + 7]);
    dft4_rot2<INV>(v[OFF + 8], v[OFF + 9], v[OFF + 10], v[OFF + 11]);
    dft4<INV>(v[OFF + 12], v[OFF + 13], v[OFF + 14], v[OFF + 15]);
}
template <bool INV>
__device__ __forceinline__ void dft16(v2f (&v)[16]) {
    dft16_at<INV, 16, 0>(v);
}

// ------------------------------------------------------------------------------------------------
// 4096-point FFT, 256 threads, LDS buffer of FFT4096_LDS complex values.
// PAD(a) = a + a/16 spreads every pass's 16-lane access groups over all banks.
constexpr int FFT4096_LDS = 4096 + 256;
__device__ __forceinline__ int pad16(int a) { return a + (a >> 4); }

// Twiddle tables (global, L1/L2 resident): tw256[k] = exp(-2*pi*i*k/256), tw4096[k] = exp(-2*pi*i*k/4096)
struct Fft4096Tables {
    const v2f* tw256;
    const v2f* tw4096;
};

// Pass 1 (Ns = 1): thread j holds x[j + 256 t] in v[t]; writes y[16 j + t].  No twiddles.
template <bool INV>
__device__ __forceinline__ void fft4096_pass1(v2f (&v)[16], v2f* lds, int j) {
    dft16<INV>(v);
    const int base = 17 * j;  // pad16(16 j + t) = 17 j + t
#pragma unroll
    for (int t = 0; t < 16; ++t) lds[base + t] = v[DFT16_OUT(t)];
}
// Pass 2 (Ns = 16): reads y[j + 256 t], twiddle exp(-+2*pi*i*(j%16)*t/256), writes z[(j/16)*256 + j%16 + 16 t].
template <bool INV>
__device__ __forceinline__ void fft4096_pass2(v2f* lds, int j, const Fft4096Tables& tb) {
    v2f v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = lds[pad16(j + 256 * t)];
    const unsigned k = (unsigned)j & 15u;
#pragma unroll
    for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], tb.tw256[k * (unsigned)t]);
    dft16<INV>(v);
    fft_sync();  // every thread has read its inputs: in-place overwrite is safe
    const int base = (j >> 4) * 272 + (int)k;  // pad16((j/16)*256 + k + 16 t) = (j/16)*272 + k + 17 t
#pragma unroll
    for (int t = 0; t < 16; ++t) lds[base + 17 * t] = v[DFT16_OUT(t)];
}
// Pass 3 (Ns = 256): reads z[j + 256 t], twiddle exp(-+2*pi*i*j*t/4096); X[j + 256 t] is left in v[t].
template <bool INV>
__device__ __forceinline__ void fft4096_pass3(v2f (&out)[16], const v2f* lds, int j, const Fft4096Tables& tb) {
    v2f v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = lds[pad16(j + 256 * t)];
#pragma unroll
    for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], tb.tw4096[(unsigned)j * (unsigned)t]);
    dft16<INV>(v);
#pragma unroll
    for (int t = 0; t < 16; ++t) out[t] = v[DFT16_OUT(t)];
}
// Whole transform: x[j + 256 t] in v[t] on entry, X[j + 256 t] in v[t] on return.  The caller must
// have a barrier between any earlier use of `lds` and this call.
template <bool INV>
__device__ __forceinline__ void fft4096(v2f (&v)[16], v2f* lds, int j, const Fft4096Tables& tb) {
    fft4096_pass1<INV>(v, lds, j);
    fft_sync();
    fft4096_pass2<INV>(lds, j, tb);
    fft_sync();
    fft4096_pass3<INV>(v, lds, j, tb);
}

// ------------------------------------------------------------------------------------------------
// Twiddle-source policies for the fused STFT kernel.  Pass-2 twiddles exp(-2 pi i (j%16) t / 256) come
// either from the global table or from a 2 KiB LDS copy; pass-3 twiddles exp(-2 pi i j t / 4096) depend
// only on the thread and come either from the global table or from 15 VGPR pairs loaded once.
template <bool TW2_LDS, bool TW3_REGS>
struct TwiddleSource {
    const v2f* tw2;         // LDS copy when TW2_LDS, else the global tw256 table
    const v2f* tw3_global;  // global tw4096 table
    v2f tw3[TW3_REGS ? 15 : 1];
    unsigned j;
    __device__ __forceinline__ v2f w2(unsigned k, int t) const { return tw2[k * (unsigned)t]; }
    __device__ __forceinline__ v2f w3(int t) const {
        if constexpr (TW3_REGS) return tw3[t - 1];
        else return tw3_global[j * (unsigned)t];
    }
};

template <bool INV, class TW>
__device__ __forceinline__ void fft4096t_pass2(const v2f* src, v2f* dst, int j, const TW& tw, bool same_buffer) {
    v2f v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = src[pad16(j + 256 * t)];
    const unsigned k = (unsigned)j & 15u;
#pragma unroll
    for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], tw.w2(k, t));
    dft16<INV>(v);
    if (same_buffer) fft_sync();
    const int base = (j >> 4) * 272 + (int)k;
#pragma unroll
    for (int t = 0; t < 16; ++t) dst[base + 17 * t] = v[DFT16_OUT(t)];
}
template <bool INV, class TW>
__device__ __forceinline__ void fft4096t_pass3(v2f (&out)[16], const v2f* lds, int j, const TW& tw) {
    v2f v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = lds[pad16(j + 256 * t)];
#pragma unroll
    for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], tw.w3(t));
    dft16<INV>(v);
#pragma unroll
    for (int t = 0; t < 16; ++t) out[t] = v[DFT16_OUT(t)];
}
// x[j + 256 t] in v[t] -> X[j + 256 t] in v[t].  PINGPONG: writes `first`, then `second`, reads `second` last
// (caller guarantees nobody still reads `first` and `second` is free).  Otherwise in place in `first`.
template <bool INV, bool PINGPONG, class TW>
__device__ __forceinline__ void fft4096t(v2f (&v)[16], v2f* first, v2f* second, int j, const TW& tw) {
    fft4096_pass1<INV>(v, first, j);
    fft_sync();
    if constexpr (PINGPONG) {
        fft4096t_pass2<INV>(first, second, j, tw, false);
        fft_sync();
        fft4096t_pass3<INV>(v, second, j, tw);
    } else {
        fft4096t_pass2<INV>(first, first, j, tw, true);
        fft_sync();
        fft4096t_pass3<INV>(v, first, j, tw);
    }
}
// Two transforms at once on the two LDS buffers (in place, shared barriers).
template <bool INV, class TW>
__device__ __forceinline__ void fft4096t_dual(v2f (&v0)[16], v2f (&v1)[16], v2f* A, v2f* B, int j, const TW& tw) {
    fft4096_pass1<INV>(v0, A, j);
    fft4096_pass1<INV>(v1, B, j);
    fft_sync();
    {
        v2f a[16], b[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            a[t] = A[pad16(j + 256 * t)];
            b[t] = B[pad16(j + 256 * t)];
        }
        const unsigned k = (unsigned)j & 15u;
#pragma unroll
        for (int t = 1; t < 16; ++t) {
            const v2f w = tw.w2(k, t);
            a[t] = twmul<INV>(a[t], w);
            b[t] = twmul<INV>(b[t], w);
        }
        dft16<INV>(a);
        dft16<INV>(b);
        fft_sync();
        const int base = (j >> 4) * 272 + (int)k;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            A[base + 17 * t] = a[DFT16_OUT(t)];
            B[base + 17 * t] = b[DFT16_OUT(t)];
        }
    }
    fft_sync();
    {
        v2f a[16], b[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            a[t] = A[pad16(j + 256 * t)];
            b[t] = B[pad16(j + 256 * t)];
        }
#pragma unroll
        for (int t = 1; t < 16; ++t) {
            const v2f w = tw.w3(t);
            a[t] = twmul<INV>(a[t], w);
            b[t] = twmul<INV>(b[t], w);
        }
        dft16<INV>(a);
        dft16<INV>(b);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            v0[t] = a[DFT16_OUT(t)];
            v1[t] = b[DFT16_OUT(t)];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Generic radix-2 FFT over `n = 1 << logn` complex values in `a` (LDS or global), executed by all
// `nthreads` threads of the workgroup.  Same decimation-in-time order and twiddles (tw[k] =
// exp(-2*pi*i*k/n), k < n/2) as the CPU oracle's fft.hpp, without mul+add fusion.
__device__ __forceinline__ unsigned bitrev(unsigned x, unsigned bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// `tw_step` lets a table built for a larger size N' = n * tw_step serve this size.
__device__ inline void fft_radix2(v2f* a, unsigned n, unsigned logn, const v2f* tw, bool inverse, unsigned tid,
                                  unsigned nthreads, unsigned tw_step = 1) {
    if (n <= 1) return;
    fft_sync();
    for (unsigned i = tid; i < n; i += nthreads) {
        const unsigned r = bitrev(i, logn);
        if (i < r) {
            const v2f t = a[i];
            a[i] = a[r];
            a[r] = t;
        }
    }
    fft_sync();
    for (unsigned s = 1; s <= logn; ++s) {
        const unsigned half = 1u << (s - 1), stride = n >> s;
        for (unsigned b = tid; b < n / 2; b += nthreads) {
            const unsigned k = b & (half - 1);
            const unsigned base = ((b >> (s - 1)) << s) + k;
            v2f w = tw[k * stride * tw_step];
            if (inverse) w.y = -w.y;
            const v2f u = a[base], x = a[base + half];
            const v2f t{x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x};
            a[base] = u + t;
            a[base + half] = u - t;
        }
        fft_sync();
    }
}

// Any-length forward DFT by Bluestein's chirp-z on top of the radix-2 transform: with c[k] = exp(-i pi k^2 / n),
//   X[k] = c[k] * sum_j (x[j] c[j]) conj(c)[k - j]      (j k = (j^2 + k^2 - (k - j)^2) / 2)
// the sum is a cyclic convolution of length m = next_pow2(2n - 1) with the chirp filter b (b[k] = b[m - k] = conj(c[k]), k < n),
// whose transform `bf` = FFT_m(b) the host prepares in double precision.  `x` holds n values in and out; `scratch` m values.
// The reference plans any length (rustfft), so shapes that are not powers of two are part of the path (generic kernels only).
__device__ inline void fft_bluestein(v2f* x, unsigned n, v2f* scratch, const BluesteinPlan& bp, unsigned tid, unsigned nthreads) {
    fft_sync();
    for (unsigned i = tid; i < bp.m; i += nthreads) {
        v2f v{0.0f, 0.0f};
        if (i < n) {
            const v2f a = x[i], c = bp.chirp[i];
            v = v2f{a.x * c.x - a.y * c.y, a.x * c.y + a.y * c.x};
        }
        scratch[i] = v;
    }
    fft_radix2(scratch, bp.m, bp.log_m, bp.tw_m, false, tid, nthreads);
    for (unsigned i = tid; i < bp.m; i += nthreads) {
        const v2f a = scratch[i], f = bp.bf[i];
        scratch[i] = v2f{a.x * f.x - a.y * f.y, a.x * f.y + a.y * f.x};
    }
    fft_radix2(scratch, bp.m, bp.log_m, bp.tw_m, true, tid, nthreads);
    const float inv_m = 1.0f / (float)bp.m;
    for (unsigned k = tid; k < n; k += nthreads) {
        const v2f a = scratch[k], c = bp.chirp[k];
        x[k] = v2f{(a.x * c.x - a.y * c.y) * inv_m, (a.x * c.y + a.y * c.x) * inv_m};
    }
    fft_sync();
}
// forward transform of any length: radix-2 when it is a power of two, Bluestein otherwise
__device__ inline void fft_forward_any(v2f* x, unsigned n, unsigned logn, const v2f* tw, v2f* scratch, const BluesteinPlan& bp, unsigned tid,
                                       unsigned nthreads) {
    if (bp.m) fft_bluestein(x, n, scratch, bp, tid, nthreads);
    else fft_radix2(x, n, logn, tw, false, tid, nthreads);
}

}  // namespace omx
