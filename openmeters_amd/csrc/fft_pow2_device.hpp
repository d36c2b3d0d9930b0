// Size-templated variant of the register / LDS FFT of fft_device.hpp for N = 1024 ... 16384 complex points.
// One transform is carried by T = N / 16 threads, 16 complex values per thread: a 256-thread workgroup runs 4, 2 or 1
// transforms of 1024 / 2048 / 4096 points side by side; 8192 and 16384 points take a 512- / 1024-thread workgroup.
// Stockham passes: radix 16, 16, R3 = N / 256 (4, 8, 16) up to 4096 points; radix 16, 16, 16, R3 = N / 4096 (2, 4) above.
// In the last pass a thread does 16 / R3 butterflies.  Same padded LDS layout, same element convention as
// the 4096-point code: thread jf holds x[jf + T u] in v[u] on entry and X[jf + T u] in v[u] on return.
#pragma once
#include "fft_device.hpp"
#include "wave_device.hpp"
#include "fft_fused_device.hpp"

namespace omx {

template <int LOGN>
struct FftGeom {
    static_assert(LOGN >= 10 && LOGN <= 14, "N = 1024 ... 16384");
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 16;                 // threads per transform
    static constexpr int PASSES = LOGN <= 12 ? 3 : 4;  // radix 16 . 16 . R3   or   16 . 16 . 16 . R3
    static constexpr int R3 = PASSES == 3 ? N / 256 : N / 4096;  // radix of the LAST pass (2, 4, 8 or 16)
    static constexpr int M = 16 / R3;                // butterflies per thread in the last pass
    static constexpr int LDS = N + N / 16;           // padded complex slots per buffer
    static constexpr int WG = T > 256 ? T : 256;     // threads per workgroup (N = 8192: 512, N = 16384: 1024)
    static constexpr int FRAMES = WG / T;            // transforms per workgroup
};

// Barrier between the threads of ONE transform.  With T = 64 a transform lives in a single wavefront, whose LDS
// instructions execute in order: no s_barrier is needed, only a fence that keeps the compiler from reordering the LDS
// accesses — the 4 transforms of a workgroup then run free of each other.
template <int LOGN>
__device__ __forceinline__ void frame_sync() {
    if constexpr (FftGeom<LOGN>::T == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
#ifdef OMX_FRAME_SYNC_LDS_ONLY
        // (a translation unit whose transforms exchange data through LDS only: the barrier waits for this wavefront's LDS traffic and
        // leaves its global loads and stores in flight — __syncthreads() drains those too, vmcnt(0), at every pass of the transform)
        lds_workgroup_barrier();
#else
        __syncthreads();
#endif
    }
}

// natural-order 8-point DFT in place (radix-2 over two dft4)
template <bool INV>
__device__ __forceinline__ void dft8(v2f& a0, v2f& a1, v2f& a2, v2f& a3, v2f& a4, v2f& a5, v2f& a6, v2f& a7) {
    constexpr float H = 0.70710678118654752440f;
    const v2f hh{H, H}, nh{-H, -H};
    dft4<INV>(a0, a2, a4, a6);  // E[0..3]
    dft4<INV>(a1, a3, a5, a7);  // O[0..3]
    const v2f o1 = add_rot<INV>(a3, a3) * hh;  // w8^1 O1 = H (1 -+ i) O1
    const v2f o3 = sub_rot<INV>(a7, a7) * nh;  // w8^3 O3 = -H (1 +- i) O3
    const v2f e0 = a0, e1 = a2, e2 = a4, e3 = a6, o0 = a1, o2 = a5;
    a0 = e0 + o0;
    a4 = e0 - o0;
    a1 = e1 + o1;
    a5 = e1 - o1;
    a2 = add_rot<INV>(e2, o2);  // w8^2 = -+ i
    a6 = sub_rot<INV>(e2, o2);
    a3 = e3 + o3;
    a7 = e3 - o3;
}

// Twiddles: pass 2 from a 256-entry table (LDS copy), pass 3 resident in VGPRs.
// FFTP_FUSED (default): the twiddle products of every radix-16 pass ride the butterflies as fused multiply-adds
// (fft_fused_device.hpp: 75 / 97 packed operations per 16-point DFT without / with outer twiddles instead of 80 / 110)
#ifndef FFTP_FUSED
#define FFTP_FUSED 1
#endif
template <bool INV>
__device__ __forceinline__ void fftp_dft16(v2f (&v)[16]) {
#if FFTP_FUSED
    dft16_fused<INV>(v);
#else
    dft16<INV>(v);
#endif
}
// v[t] *= w(t) for t = 1 ... 15, then the DFT; `w` is called once per t
template <bool INV, class W>
__device__ __forceinline__ void fftp_tw_dft16(v2f (&v)[16], W&& w) {
#if FFTP_FUSED
    v2f ww[16];
#pragma unroll
    for (int t = 1; t < 16; ++t) ww[t] = w(t);
    ww[0] = ww[1];
    dft16_fused_tw<INV>(v, ww);
#else
#pragma unroll
    for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], w(t));
    dft16<INV>(v);
#endif
}
template <bool INV, class W>
__device__ __forceinline__ void fftp_tw_dft16_dual(v2f (&a)[16], v2f (&b)[16], W&& w) {
#if FFTP_FUSED
    v2f ww[16];
#pragma unroll
    for (int t = 1; t < 16; ++t) ww[t] = w(t);
    ww[0] = ww[1];
    dft16_fused_tw<INV>(a, ww);
    dft16_fused_tw<INV>(b, ww);
#else
#pragma unroll
    for (int t = 1; t < 16; ++t) {
        const v2f x = w(t);
        a[t] = twmul<INV>(a[t], x);
        b[t] = twmul<INV>(b[t], x);
    }
    dft16<INV>(a);
    dft16<INV>(b);
#endif
}

template <int LOGN>
struct TwiddlesPow2 {
    using G = FftGeom<LOGN>;
    const v2f* tw2;  // exp(-2 pi i k / 256), k < 256
    const v2f* twN;  // exp(-2 pi i k / N), k < N (global): pass-3 twiddles of the four-pass sizes are read from it at use
    v2f tw3[15];     // last pass: tw3[u - 1] = exp(-2 pi i t b / N), u = m + M t, b = jf + T m  (unused while t == 0)
    __device__ __forceinline__ void load(const v2f* table, unsigned jf) {
        twN = table;
#pragma unroll
        for (int u = 1; u < 16; ++u) {
            const unsigned t = (unsigned)(u / G::M), m = (unsigned)(u % G::M);
            tw3[u - 1] = t ? table[t * (jf + (unsigned)G::T * m)] : v2f{1.0f, 0.0f};
        }
    }
};

template <bool INV, int LOGN>
__device__ __forceinline__ void fftp_pass1(v2f (&v)[16], v2f* lds, int jf) {
    fftp_dft16<INV>(v);
    const int base = 17 * jf;  // pad16(16 jf + t)
#pragma unroll
    for (int t = 0; t < 16; ++t) lds[base + t] = v[DFT16_OUT(t)];
}
template <bool INV, int LOGN>
__device__ __forceinline__ void fftp_pass2(const v2f* src, v2f* dst, int jf, const TwiddlesPow2<LOGN>& tw) {
    using G = FftGeom<LOGN>;
    v2f v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = src[pad16(jf + G::T * t)];
    const unsigned k = (unsigned)jf & 15u;
    fftp_tw_dft16<INV>(v, [&](int t) { return tw.tw2[k * (unsigned)t]; });
    const int base = (jf >> 4) * 272 + (int)k;
#pragma unroll
    for (int t = 0; t < 16; ++t) dst[base + 17 * t] = v[DFT16_OUT(t)];
}
template <bool INV, int LOGN>
__device__ __forceinline__ void fftp_pass3(v2f (&out)[16], const v2f* lds, int jf, const TwiddlesPow2<LOGN>& tw) {
    using G = FftGeom<LOGN>;
    v2f v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = lds[pad16(jf + G::T * u)];
    if constexpr (G::R3 == 16) {
        fftp_tw_dft16<INV>(v, [&](int t) { return tw.tw3[t - 1]; });
#pragma unroll
        for (int t = 0; t < 16; ++t) out[t] = v[DFT16_OUT(t)];
        return;
    }
#pragma unroll
    for (int u = G::M; u < 16; ++u) v[u] = twmul<INV>(v[u], tw.tw3[u - 1]);
    if constexpr (G::R3 == 8) {
        dft8<INV>(v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14]);
        dft8<INV>(v[1], v[3], v[5], v[7], v[9], v[11], v[13], v[15]);
#pragma unroll
        for (int u = 0; u < 16; ++u) out[u] = v[u];
    } else if constexpr (G::R3 == 2) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            out[m] = v[m] + v[m + 8];
            out[m + 8] = v[m] - v[m + 8];
        }
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) dft4<INV>(v[m], v[m + 4], v[m + 8], v[m + 12]);
#pragma unroll
        for (int u = 0; u < 16; ++u) out[u] = v[u];
    }
}
// x[jf + T u] in v[u] -> X[jf + T u] in v[u].  Writes `first`, then `second`, reads `second` last (ping-pong); the caller
// guarantees nobody still reads `first` and that `second` is free.  All 256 threads of the workgroup must call it together.
template <bool INV, int LOGN>
__device__ __forceinline__ void fftp(v2f (&v)[16], v2f* first, v2f* second, int jf, const TwiddlesPow2<LOGN>& tw);
template <bool INV, int LOGN>
__device__ __forceinline__ void fftp_inplace(v2f (&v)[16], v2f* buf, int jf, const TwiddlesPow2<LOGN>& tw);

template <bool INV, int LOGN>
__device__ __forceinline__ void fftp(v2f (&v)[16], v2f* first, v2f* second, int jf, const TwiddlesPow2<LOGN>& tw) {
    if constexpr (FftGeom<LOGN>::PASSES == 4) {
        // four passes would end on `first`: run in place on `second` instead, so `second` is still the buffer read last.
        // The ping-pong contract only frees `second` one barrier into the transform, hence the barrier up front.
        frame_sync<LOGN>();
        fftp_inplace<INV, LOGN>(v, second, jf, tw);
        return;
    }
    fftp_pass1<INV, LOGN>(v, first, jf);
    frame_sync<LOGN>();
    fftp_pass2<INV, LOGN>(first, second, jf, tw);
    frame_sync<LOGN>();
    fftp_pass3<INV, LOGN>(v, second, jf, tw);
}
// Third pass of the four-pass sizes (Ns = 256, radix 16): reads y[jf + T t], twiddle exp(-+2 pi i t (jf % 256) / 4096), writes
// z[(jf / 256) * 4096 + jf % 256 + 256 t] in place (barrier between the loads and the stores).
template <bool INV, int LOGN>
__device__ __forceinline__ void fftp_mid3_inplace(v2f* buf, int jf, const TwiddlesPow2<LOGN>& tw) {
    using G = FftGeom<LOGN>;
    v2f a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) a[t] = buf[pad16(jf + G::T * t)];
    const unsigned k = (unsigned)jf & 255u;
    fftp_tw_dft16<INV>(a, [&](int t) { return tw.twN[(k * (unsigned)t) * (unsigned)(G::N / 4096)]; });
    frame_sync<LOGN>();
    const int base = (jf >> 8) * 4352 + (int)k + (int)(k >> 4);  // pad16(4096 q + k + 256 t) = 4352 q + k + k/16 + 272 t
#pragma unroll
    for (int t = 0; t < 16; ++t) buf[base + 272 * t] = a[DFT16_OUT(t)];
}
// Ping-pong transform of a SUBSET of a frame's threads (`act`), inside a frame whose barriers are sized by LOGSYNC: every
// thread of the frame reaches the barriers, only the active ones touch data.  Used by the zero-padded kernel, whose Hilbert
// transforms (window length W) are smaller than its windowed transforms (F = zp W).  Three-pass sizes only.
template <bool INV, int LOGN, int LOGSYNC>
__device__ __forceinline__ void fftp_masked(bool act, v2f (&v)[16], v2f* first, v2f* second, int jf, const TwiddlesPow2<LOGN>& tw) {
    static_assert(FftGeom<LOGN>::PASSES == 3, "three-pass sizes only");
    if (act) fftp_pass1<INV, LOGN>(v, first, jf);
    frame_sync<LOGSYNC>();
    if (act) fftp_pass2<INV, LOGN>(first, second, jf, tw);
    frame_sync<LOGSYNC>();
    if (act) fftp_pass3<INV, LOGN>(v, second, jf, tw);
}
// In place in one buffer (one extra barrier in pass 2): half the LDS of the ping-pong form, for kernels that run a single
// transform per frame slot (classic columns, spectrum).
template <bool INV, int LOGN>
__device__ __forceinline__ void fftp_inplace(v2f (&v)[16], v2f* buf, int jf, const TwiddlesPow2<LOGN>& tw) {
    using G = FftGeom<LOGN>;
    fftp_pass1<INV, LOGN>(v, buf, jf);
    frame_sync<LOGN>();
    {
        v2f a[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) a[t] = buf[pad16(jf + G::T * t)];
        const unsigned k = (unsigned)jf & 15u;
        fftp_tw_dft16<INV>(a, [&](int t) { return tw.tw2[k * (unsigned)t]; });
        frame_sync<LOGN>();
        const int base = (jf >> 4) * 272 + (int)k;
#pragma unroll
        for (int t = 0; t < 16; ++t) buf[base + 17 * t] = a[DFT16_OUT(t)];
    }
    frame_sync<LOGN>();
    if constexpr (G::PASSES == 4) {
        fftp_mid3_inplace<INV, LOGN>(buf, jf, tw);
        frame_sync<LOGN>();
    }
    fftp_pass3<INV, LOGN>(v, buf, jf, tw);
}
// Two transforms at once, in place on their own buffers (shared barriers).
template <bool INV, int LOGN>
__device__ __forceinline__ void fftp_dual(v2f (&v0)[16], v2f (&v1)[16], v2f* A, v2f* B, int jf, const TwiddlesPow2<LOGN>& tw) {
    using G = FftGeom<LOGN>;
    fftp_pass1<INV, LOGN>(v0, A, jf);
    fftp_pass1<INV, LOGN>(v1, B, jf);
    frame_sync<LOGN>();
    {
        v2f a[16], b[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            a[t] = A[pad16(jf + G::T * t)];
            b[t] = B[pad16(jf + G::T * t)];
        }
        const unsigned k = (unsigned)jf & 15u;
        fftp_tw_dft16_dual<INV>(a, b, [&](int t) { return tw.tw2[k * (unsigned)t]; });
        frame_sync<LOGN>();
        const int base = (jf >> 4) * 272 + (int)k;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            A[base + 17 * t] = a[DFT16_OUT(t)];
            B[base + 17 * t] = b[DFT16_OUT(t)];
        }
    }
    frame_sync<LOGN>();
    if constexpr (G::PASSES == 4) {
        v2f a[16], b[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            a[t] = A[pad16(jf + G::T * t)];
            b[t] = B[pad16(jf + G::T * t)];
        }
        const unsigned k = (unsigned)jf & 255u;
        fftp_tw_dft16_dual<INV>(a, b, [&](int t) { return tw.twN[(k * (unsigned)t) * (unsigned)(G::N / 4096)]; });
        frame_sync<LOGN>();
        const int base = (jf >> 8) * 4352 + (int)k + (int)(k >> 4);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            A[base + 272 * t] = a[DFT16_OUT(t)];
            B[base + 272 * t] = b[DFT16_OUT(t)];
        }
        frame_sync<LOGN>();
    }
    fftp_pass3<INV, LOGN>(v0, A, jf, tw);
    fftp_pass3<INV, LOGN>(v1, B, jf, tw);
}

}  // namespace omx
