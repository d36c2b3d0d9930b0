// K2 (round 2 form): fused reassigned STFT, W = F = 4096, H = 8192 — TWO consecutive columns of one stream per workgroup and
// FOUR 4096-point transforms per column (reference spectrogram/processor.rs:318-348, :439-488, :546-567).
//
// What changed against stft_reassigned_4096_kernel (stft_kernels.hip, kept as OMX_OPT_KERNEL_FORM = 1):
//   * The three windowed transforms FFT(w s), FFT(w' s), FFT(t w s) become TWO: every window the fused path serves is a
//     two-term cosine sum w[n] = c0 + c1 cos(2 pi n / W) (Hann, Hamming; window.rs:20-43), so with Z = FFT(s) and
//     Z2 = FFT((n - (W-1)/2) s):
//         FFT(w s)[k]    = c0 Z[k]  + c1/2 (Z[k-1]  + Z[k+1])
//         FFT(t w s)[k]  = c0 Z2[k] + c1/2 (Z2[k-1] + Z2[k+1])
//         FFT(w' s)[k]   = i c1 (pi/W) (Z[k-1] - Z[k+1])        (w' = spectral derivative of w = -c1 (2 pi / W) sin(2 pi n / W),
//                                                                processor.rs:569-599; DC / Nyquist zeroing never touches it)
//     Same accuracy against exact arithmetic as windowing in the time domain (tests/test_exact_f64.py; measured in numpy f32
//     before the kernel was written: power 5e-8, weighted f-hat 3e-11 ... 3e-8, t-hat 2e-7 of the bars' units either way).
//   * Both columns of the pair ride every transform as a DUAL transform (two independent dependency chains per wavefront,
//     shared barriers): the packed-real forward and the inverse of the Hilbert pair used to run as single transforms,
//     which cost 0.19 + 0.31 ms per 65 536 frames against 0.48 ms for a dual one (DESIGN §5 knock-out prices).
// LDS: two padded 4096-complex buffers (68 KiB) -> two workgroups = four columns in flight per CU (was two).
#include "stft_kernels.hpp"

#include "fft_device.hpp"

namespace omx {

namespace {

struct PairConsts {
    float bin_hz, max_hz, inv_2pi, inv_hop, latency_hops;
};

// spectrogram/processor.rs:459-485 for one bin (same statement order as reassign_bin in stft_kernels.hip)
__device__ __forceinline__ bool reassign_one(uint32_t i, v2f b, v2f d, v2f t, float norm, const PairConsts& c, omx_spectrogram_point& p) {
    const float pow = b.x * b.x + b.y * b.y;
    const float scaled_power = pow * norm;
    if (scaled_power < 1e-14f) return false;  // ANALYSIS_FLOOR_POWER (:69)
    const float inv_pow = 1.0f / pow;
    const float d_omega = -(d.y * b.x - d.x * b.y) * inv_pow;
    const float freq_hz = (float)i * c.bin_hz + d_omega * c.inv_2pi;
    if (!(freq_hz > 0.0f && c.max_hz - freq_hz > 0.0f)) return false;
    p.time_offset = (t.x * b.x + t.y * b.y) * inv_pow * c.inv_hop - c.latency_hops;
    p.freq_hz = freq_hz;
    p.power = scaled_power;
    return true;
}

__device__ __forceinline__ bool pair_block_to_stream_chunk(uint32_t n_streams, uint32_t chunks, uint32_t& s, uint32_t& chunk) {
    // XCD-aware map (same as block_to_stream_column): block b runs on XCD b % 8; stream s is pinned to XCD s % 8
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, q = b >> 3;
    s = (q / chunks) * 8u + xcd;
    chunk = q % chunks;
    return s < n_streams;
}

// Two transforms at once, in place in A and B (same arithmetic as fft4096t_dual).  Tried and dropped (same-box A/B, kernel ms
// per 65 536 frames, 1.72 as committed): LDS reads issued as single ds_read_b64 through inline asm instead of the
// ds_read2st64_b64 pairs hipcc forms (1.78: the forced full wait and 18 spilled registers cost more than the read2 penalty),
// volatile reads (3.08: 58 spilled registers), -amdgpu-sched-strategy=max-ilp (1.89: 16 spilled registers).
template <bool INV, class TW>
__device__ __forceinline__ void pair_dual(v2f (&v0)[16], v2f (&v1)[16], v2f* A, v2f* B, int j, const TW& tw) {
    fft4096_pass1<INV>(v0, A, j);
    fft4096_pass1<INV>(v1, B, j);
    __syncthreads();
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
        __syncthreads();
        const int base = (j >> 4) * 272 + (int)k;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            A[base + 17 * t] = a[DFT16_OUT(t)];
            B[base + 17 * t] = b[DFT16_OUT(t)];
        }
    }
    __syncthreads();
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

}  // namespace

__global__ __launch_bounds__(256, 2) void stft_reassigned_4096_pair_kernel(StftFastArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* A = reinterpret_cast<v2f*>(smem_raw);
    v2f* B = A + FFT4096_LDS;
    v2f* tw2_lds = B + FFT4096_LDS;                                // [256] exp(-2 pi i k / 256)
    uint32_t* scan = reinterpret_cast<uint32_t*>(tw2_lds + 256);   // [9][4] wave counts
    float* hil = reinterpret_cast<float*>(scan + 36);              // X[0]/2, X[4096]/2 of both columns

    const uint32_t chunks = (a.n_cols + 1u) / 2u;
    uint32_t s, chunk;
    if (!pair_block_to_stream_chunk(a.n_streams, chunks, s, chunk)) return;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const int lane = j & 63, wave = j >> 6;
    const float* ring = a.ring + (uint64_t)s * a.cap;
    const uint32_t mask32 = (uint32_t)(a.cap - 1);  // cap <= 2^30 (checked on the host)
    const uint32_t bytemask = mask32 << 2;
    const char* ring_bytes = reinterpret_cast<const char*>(ring);
    const long long last_nonzero = a.last_nonzero[s];
    const PairConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};

    const uint32_t col0 = chunk * 2u;
    const uint32_t n_cols_s = stft_cols(a, s);  // ragged banks: this stream's own column count
    if (col0 >= n_cols_s) return;
    const bool have1 = col0 + 1u < n_cols_s;
    const uint32_t col1 = have1 ? col0 + 1u : col0;  // an odd tail computes column 0 twice and stores it once
    const uint64_t tail_s = stft_tail(a, s);
    const uint64_t p0a = tail_s + (uint64_t)col0 * a.hop, p0b = tail_s + (uint64_t)col1 * a.hop;
    // silent fast path (:307-316): no non-zero sample at or after the front of the pending buffer
    const bool silent_a = last_nonzero < (long long)p0a, silent_b = last_nonzero < (long long)p0b;
    uint32_t* count_a = a.counts + (uint64_t)s * a.n_cols + col0;
    uint32_t* count_b = a.counts + (uint64_t)s * a.n_cols + col1;
    if (silent_a && (silent_b || !have1)) {  // column 1 starts later: silent_a implies silent_b
        if (j == 0) {
            *count_a = 0;
            if (have1) *count_b = 0;
        }
        return;
    }

    using TW = TwiddleSource<true, true>;
    TW tw;
    tw.j = ju;
    tw.tw3_global = a.tw4096;
    tw.tw2 = tw2_lds;
#pragma unroll
    for (int t = 1; t < 16; ++t) tw.tw3[t - 1] = a.tw4096[ju * (unsigned)t];
    tw2_lds[j] = a.tw256[ju];  // first read in pass 2 of the first transform, behind that transform's pass-1 barrier

    // ---- 1. packed real FFTs of the two 8192-sample windows -----------------------------------------------------------
    const uint32_t pa32 = (uint32_t)p0a, pb32 = (uint32_t)p0b;
    v2f va[16], vb[16];
    auto load_window = [&](v2f (&v)[16], uint64_t p0, uint32_t p32) {
        if ((p0 & 1ull) == 0) {  // pairs are 8-byte aligned and never straddle the ring wrap
#pragma unroll
            for (int t = 0; t < 16; ++t)
                v[t] = *reinterpret_cast<const v2f*>(ring_bytes + (((p32 + 2u * (ju + 256u * (unsigned)t)) << 2) & bytemask));
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint32_t q = p32 + 2u * (ju + 256u * (unsigned)t);
                v[t] = v2f{ring[q & mask32], ring[(q + 1u) & mask32]};
            }
        }
    };
    load_window(va, p0a, pa32);
    load_window(vb, p0b, pb32);
    pair_dual<false>(va, vb, A, B, j, tw);  // v[t] = Zf[j + 256 t]

    // ---- 2. Hilbert transform with ONE half-length inverse per column (derivation: stft_kernels.hip step 2) -----------
    __syncthreads();  // pass 3 of the dual transform still reads A and B
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        A[pad16(j + 256 * t)] = va[t];
        B[pad16(j + 256 * t)] = vb[t];
    }
    if (j == 0) {
        hil[0] = (va[0].x + va[0].y) * 0.5f;  // X[0] / 2
        hil[1] = (va[0].x - va[0].y) * 0.5f;  // X[4096] / 2
        hil[2] = (vb[0].x + vb[0].y) * 0.5f;
        hil[3] = (vb[0].x - vb[0].y) * 0.5f;
    }
    __syncthreads();
    v2f ya[16], yb[16];
    {
        v2f w8[16];  // exp(-2 pi i k / 8192) / 2, k = j + 256 t: one table read serves both columns
#pragma unroll
        for (int t = 0; t < 16; ++t) w8[t] = a.tw8192[ju + 256u * (unsigned)t];
        auto hilbert_spectrum = [&](v2f (&y)[16], const v2f (&v)[16], const v2f* X) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const unsigned k = (unsigned)(j + 256 * t);
                const v2f z = v[t], zr = X[pad16((int)((4096u - k) & 4095u))];
                const v2f sum{z.x + zr.x, z.y - zr.y}, dif{z.x - zr.x, z.y + zr.y};
                y[t] = cmulc(sum, w8[t]) - cmul(dif, w8[t]);
                if (t == 0 && k == 0) y[t] = v2f{0.0f, 0.0f};
            }
        };
        hilbert_spectrum(ya, va, A);
        __builtin_amdgcn_sched_barrier(0);  // one column at a time: interleaving the two loops doubles the live partner reads
        hilbert_spectrum(yb, vb, B);
        __builtin_amdgcn_sched_barrier(0);
    }
    const float half_x0a = hil[0], half_xna = hil[1], half_x0b = hil[2], half_xnb = hil[3];
    // the real part's samples, in flight during the inverse: Re analytic[n] = 4096 x[n] - X[0]/2 + X[4096] (-1)^n / 2
    float xra[16], xrb[16];
    {
        const uint32_t qa = pa32 + 2048u + ju, qb = pb32 + 2048u + ju;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            xra[t] = *reinterpret_cast<const float*>(ring_bytes + (((qa + 256u * (unsigned)t) << 2) & bytemask));
            xrb[t] = *reinterpret_cast<const float*>(ring_bytes + (((qb + 256u * (unsigned)t) << 2) & bytemask));
        }
    }
    __syncthreads();  // partners are read from the buffers the inverse is about to overwrite
    pair_dual<true>(ya, yb, A, B, j, tw);  // y[t] = (Im a[2m], Im a[2m+1]), m = j + 256 t

    // ---- 3. gather s[i] = analytic[2048 + i], i = j + 256 t, for both columns -------------------------------------------
    __syncthreads();
    float* imag_a = reinterpret_cast<float*>(A);
    float* imag_b = reinterpret_cast<float*>(B);
#pragma unroll
    for (int t = 4; t < 12; ++t) {
        *reinterpret_cast<v2f*>(imag_a + 2 * (j + 256 * t - 1024)) = ya[t];
        *reinterpret_cast<v2f*>(imag_b + 2 * (j + 256 * t - 1024)) = yb[t];
    }
    __syncthreads();
    v2f sa[16], sb[16];
    {
        const float par_a = (j & 1) ? -half_xna : half_xna, par_b = (j & 1) ? -half_xnb : half_xnb;  // (-1)^n: n = 2048 + i has j's parity
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            sa[t] = v2f{4096.0f * xra[t] - half_x0a + par_a, imag_a[j + 256 * t]};
            sb[t] = v2f{4096.0f * xrb[t] - half_x0b + par_b, imag_b[j + 256 * t]};
        }
    }
    const float c0 = a.win_c0, half_c1 = 0.5f * a.win_c1, dscale = a.win_c1 * (3.14159265358979323846f / 4096.0f);
    v2f* lin_z = A;   // [LIN_BINS] natural-order bins of Z  (slot 1 + k)
    v2f* lin_z2 = B;  // [LIN_BINS] natural-order bins of Z2

    // ---- 4. per column: Z = FFT(s), Z2 = FFT((n - 2047.5) s) as one dual transform; windows applied on the bins --------
    auto column = [&](const v2f (&sv)[16], bool silent, uint32_t col, uint32_t* count_out) {
        v2f z[16], z2[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float nc = (float)(j + 256 * t) - 2047.5f;  // compute_time_weighted's ramp (:601-608)
            z[t] = sv[t];
            z2[t] = v2f{sv[t].x * nc, sv[t].y * nc};
        }
        __syncthreads();  // the gather above / the previous column's neighbour reads still use A and B
        pair_dual<false>(z, z2, A, B, j, tw);
        float pn[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) pn[t] = a.bin_norm[(t < 8 || j == 0) ? ju + 256u * (unsigned)t : 0u];
        __syncthreads();  // pass 3 still reads A and B
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            lin_z[1 + j + 256 * t] = z[t];
            lin_z2[1 + j + 256 * t] = z2[t];
        }
        if (j == 255) {  // bin -1 = bin 4095
            lin_z[0] = z[15];
            lin_z2[0] = z2[15];
        }
        __syncthreads();

        omx_spectrogram_point pts[9];
        unsigned long long masks[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint32_t bin = (uint32_t)(j + 256 * t);
            bool keep = false;
            if ((t < 8 || j == 0) && !silent) {
                const v2f zm = lin_z[bin], zp = lin_z[bin + 2], z2m = lin_z2[bin], z2p = lin_z2[bin + 2];
                const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y}, z2s{z2m.x + z2p.x, z2m.y + z2p.y};
                const v2f bb{c0 * z[t].x + half_c1 * zs.x, c0 * z[t].y + half_c1 * zs.y};
                const v2f bd{-dscale * zd.y, dscale * zd.x};  // i c1 (pi / W) (Z[k-1] - Z[k+1])
                const v2f bt{c0 * z2[t].x + half_c1 * z2s.x, c0 * z2[t].y + half_c1 * z2s.y};
                keep = reassign_one(bin, bb, bd, bt, pn[t], rc, pts[t]);
            }
            masks[t] = __ballot(keep);
            if (lane == 0) scan[t * 4 + wave] = (uint32_t)__popcll(masks[t]);
        }
        __syncthreads();
        omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
        uint4 counts4[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) counts4[t] = *reinterpret_cast<const uint4*>(scan + t * 4);
        uint32_t running = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint32_t c[4] = {counts4[t].x, counts4[t].y, counts4[t].z, counts4[t].w};
            uint32_t before = running;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w < wave) before += c[w];
                running += c[w];
            }
            if ((masks[t] >> lane) & 1ull) {
                const uint32_t pos = before + (uint32_t)__popcll(masks[t] & ((1ull << lane) - 1ull));
                *reinterpret_cast<omx_spectrogram_point*>(reinterpret_cast<char*>(out) + pos * 12u) = pts[t];
            }
        }
        if (j == 0) *count_out = running;
    };
    column(sa, silent_a, col0, count_a);
    if (have1) column(sb, silent_b, col1, count_b);
}

void launch_stft_reassigned_4096_pair(const StftFastArgs& a, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return;
    const size_t lds = (size_t)(2 * FFT4096_LDS + 256) * sizeof(v2f) + 9 * 4 * sizeof(uint32_t) + 4 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_4096_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    const uint32_t chunks = (a.n_cols + 1u) / 2u;
    hipLaunchKernelGGL(stft_reassigned_4096_pair_kernel, dim3(stream_column_grid(a.n_streams, chunks)), dim3(256), lds, stream, a);
}

}  // namespace omx
