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

#include "buffer_device.hpp"
#include "fft_device.hpp"
#include "reassign_device.hpp"
#include "twiddle_run_device.hpp"
#include <cstdlib>

namespace omx {

namespace {

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
// w8[t] = base * exp(-2 pi i t / 32): the 8192-point twiddles of the Hilbert step as one table read per thread and 15 rotations by
// compile-time constants (a run of 16 table reads is a global round trip in the middle of the kernel with nothing to overlap it)
template <int TT>
__device__ __forceinline__ void w8_run(v2f (&w8)[16], v2f base) {
    if constexpr (TT < 16) {
        w8[TT] = rotate128<4 * TT>(base);
        w8_run<TT + 1>(w8, base);
    }
}
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
template <bool INV, class TW, class Hook = NoHook>
__device__ __forceinline__ void pair_dual(v2f (&v0)[16], v2f (&v1)[16], v2f* A, v2f* B, int j, const TW& tw, Hook before_first_barrier = Hook{}) {
    fft4096_pass1<INV>(v0, A, j);
    fft4096_pass1<INV>(v1, B, j);
    before_first_barrier();
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

// PHASES (tuning builds, OMX_K2_VARIANT=52): thread 0 adds the shader-clock cycles between phase marks to phases[0 ... 8]
template <bool PHASES>
__global__ __launch_bounds__(256, 2) void stft_reassigned_4096_pair_kernel(StftFastArgs a, unsigned long long* phases) {
    long long phase_t = 0;
    if constexpr (PHASES) phase_t = clock64();
    auto mark = [&](int i) {
        if constexpr (PHASES) {
            if (threadIdx.x == 0) {
                const long long now = clock64();
                atomicAdd(&phases[i], (unsigned long long)(now - phase_t));
                phase_t = now;
            }
        }
    };
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
    // per-stream words first, back to back (each is a dependent scalar load: one wait for the three, not three)
    const long long last_nonzero = a.last_nonzero[s];
    const uint32_t* cols_p = a.cols;
    const uint64_t* tails_p = a.tails;
    const uint32_t n_cols_s = cols_p ? cols_p[s] : a.n_cols;  // ragged banks: this stream's own column count ...
    const uint64_t tail_s = tails_p ? tails_p[s] : a.tail;    // ... and tail
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};

    const uint32_t col0 = chunk * 2u;
    if (col0 >= n_cols_s) return;
    const bool have1 = col0 + 1u < n_cols_s;
    const uint32_t col1 = have1 ? col0 + 1u : col0;  // an odd tail computes column 0 twice and stores it once
    const uint64_t p0a = tail_s + (uint64_t)col0 * a.hop, p0b = tail_s + (uint64_t)col1 * a.hop;
    uint32_t* count_a = a.counts + (uint64_t)s * a.n_cols + col0;
    uint32_t* count_b = a.counts + (uint64_t)s * a.n_cols + col1;

    // Everything the forward transforms need from memory is requested here, in one batch: the pass-2 twiddle this thread copies
    // to LDS, both windows, the resident pass-3 twiddles.
    using TW = TwiddleSource<true, true>;
    TW tw;
    tw.j = ju;
    tw.tw3_global = a.tw4096;
    tw.tw2 = tw2_lds;
    const v2f tw2_mine = a.tw256[ju];

    // ---- 1. packed real FFTs of the two 8192-sample windows -----------------------------------------------------------
    const uint32_t pa32 = (uint32_t)p0a, pb32 = (uint32_t)p0b;
    v2f va[16], vb[16];
    // Both windows lie in one piece of the ring (no wrap inside [p0a, p0b + 8192)) and pairs are 8-byte aligned: buffer loads, one
    // lane offset for the whole batch and the step in a scalar register (buffer_device.hpp); otherwise every index is wrapped by the mask.
    const uint32_t off_a = pa32 & mask32, hop_bytes = (pb32 - pa32) * 4u;
    const bool direct = (uint64_t)off_a + (uint64_t)(pb32 - pa32) + 8192ull <= a.cap && ((p0a | p0b) & 1ull) == 0;
    const GlobalBuffer windowb = global_buffer(ring + off_a, hop_bytes + 8192u * 4u);
    const GlobalBuffer tw4096b = global_buffer(a.tw4096, 4096u * 8u), tw8192b = global_buffer(a.tw8192, 4096u * 8u),
                       normb = global_buffer(a.bin_norm, 2049u * 4u);
    if (direct) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            va[t] = load_v2f(windowb, ju * 8u, 2048u * (unsigned)t);
            vb[t] = load_v2f(windowb, ju * 8u, hop_bytes + 2048u * (unsigned)t);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t qa = pa32 + 2u * (ju + 256u * (unsigned)t), qb = pb32 + 2u * (ju + 256u * (unsigned)t);
            va[t] = v2f{ring[qa & mask32], ring[(qa + 1u) & mask32]};
            vb[t] = v2f{ring[qb & mask32], ring[(qb + 1u) & mask32]};
        }
    }
#pragma unroll
    for (int t = 1; t < 16; ++t) tw.tw3[t - 1] = load_v2f(tw4096b, ju * 8u * (unsigned)t, 0);
    const v2f w8_base = load_v2f(tw8192b, ju * 8u, 0);  // exp(-2 pi i j / 8192) / 2: element j + 256 t of the Hilbert step's twiddles is it times exp(-2 pi i t / 32)
    float pn[9];  // bin normalisation of this thread's bins: the same for both columns, requested with the first batch
#pragma unroll
    for (int t = 0; t < 9; ++t) pn[t] = load_f32(normb, ju * 4u, 1024u * (unsigned)t);  // (t = 8, j > 0: past the table, reads 0, not used)
    // silent fast path (:307-316): no non-zero sample at or after the front of the pending buffer.  Decided AFTER the first batch of
    // loads is in flight: `last_nonzero` is one more dependent scalar round trip, which now overlaps the vector loads instead of
    // preceding them (a silent pair wastes its loads; it is the rare case)
    const bool silent_a = last_nonzero < (long long)p0a, silent_b = last_nonzero < (long long)p0b;
    if (silent_a && (silent_b || !have1)) {  // column 1 starts later: silent_a implies silent_b
        if (j == 0) {
            *count_a = 0;
            if (have1) *count_b = 0;
        }
        return;
    }
    mark(0);
    // (the LDS copy of the pass-2 twiddles is first read in pass 2, behind the pass-1 barrier)
    pair_dual<false>(va, vb, A, B, j, tw, [&] { tw2_lds[j] = tw2_mine; });  // v[t] = Zf[j + 256 t]
    mark(1);

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
        // exp(-2 pi i k / 8192) / 2, k = j + 256 t: one table read serves both columns.  (Requested earlier — with the first batch, or
        // ahead of the natural-order copy — the kernel is 1-2 % SLOWER, same-box A/B.)
        v2f w8[16];
        w8_run<0>(w8, w8_base);
        // partner Zf[(4096 - k) & 4095] of k = j + 256 t sits at pad16(4096 - j) - 272 t (4096 - j is not a multiple of 16 for
        // j > 0 ... and pad16 is linear across multiples of 256 anyway); thread 0's partners 4096 - 256 t sit at 4352 - 272 t, and
        // its t = 0 read (one slot past the block, inside the allocation) is not used
        const int part = (j ? pad16(4096 - j) : 4352) - 272 * 15;
        auto hilbert_spectrum = [&](v2f (&y)[16], const v2f (&v)[16], const v2f* X) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const unsigned k = (unsigned)(j + 256 * t);
                const v2f z = v[t], zr = X[part + 272 * (15 - t)];
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
    if (direct) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            xra[t] = load_f32(windowb, ju * 4u, 8192u + 1024u * (unsigned)t);
            xrb[t] = load_f32(windowb, ju * 4u, hop_bytes + 8192u + 1024u * (unsigned)t);
        }
    } else {
        const uint32_t qa = pa32 + 2048u + ju, qb = pb32 + 2048u + ju;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            xra[t] = *reinterpret_cast<const float*>(ring_bytes + (((qa + 256u * (unsigned)t) << 2) & bytemask));
            xrb[t] = *reinterpret_cast<const float*>(ring_bytes + (((qb + 256u * (unsigned)t) << 2) & bytemask));
        }
    }
    __syncthreads();  // partners are read from the buffers the inverse is about to overwrite
    mark(2);
    pair_dual<true>(ya, yb, A, B, j, tw);  // y[t] = (Im a[2m], Im a[2m+1]), m = j + 256 t
    mark(3);

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
    mark(4);
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
        mark(5);
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

        // Branch-free, neighbour reads first: written with the reference's early returns, every bin was a serial chain of
        // `LDS read - wait - arithmetic - branch - LDS read - wait`, each wait exposed (the phase cost as much as a dual transform).
        // Every lane computes every bin it touches; what the early returns decided goes into the keep flag.
        omx_spectrogram_point pts[9];
        unsigned long long masks[9];
#pragma unroll
        for (int h = 0; h < 9; h += 3) {  // three bins at a time: 24 registers of neighbours in flight
            v2f nzm[3], nzp[3], nz2m[3], nz2p[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int bin = j + 256 * (h + u);  // (t = 8: only thread 0's bin exists; the others read slots inside the buffer and drop the result)
                nzm[u] = lin_z[bin];
                nzp[u] = lin_z[bin + 2];
                nz2m[u] = lin_z2[bin];
                nz2p[u] = lin_z2[bin + 2];
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int t = h + u;
                const uint32_t bin = (uint32_t)(j + 256 * t);
                const v2f zm = nzm[u], zp = nzp[u], z2m = nz2m[u], z2p = nz2p[u];
                const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y}, z2s{z2m.x + z2p.x, z2m.y + z2p.y};
                const v2f bb{c0 * z[t].x + half_c1 * zs.x, c0 * z[t].y + half_c1 * zs.y};
                const v2f bd{-dscale * zd.y, dscale * zd.x};  // i c1 (pi / W) (Z[k-1] - Z[k+1])
                const v2f bt{c0 * z2[t].x + half_c1 * z2s.x, c0 * z2[t].y + half_c1 * z2s.y};
                const bool keep = reassign_flat(bin, bb, bd, bt, pn[t], rc, pts[t]) && (t < 8 || j == 0) && !silent;
                masks[t] = __ballot(keep);
                if (lane == 0) scan[t * 4 + wave] = (uint32_t)__popcll(masks[t]);
            }
        }
        __syncthreads();
        omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
        // exclusive prefix of the 36 wave counts ([t][wave] row-major = bin order), by every wavefront for itself: one LDS read,
        // six DPP adds, then one v_readlane per t — 5 VALU instructions per t where summing the counts per thread took 22
        const uint32_t cnt = lane < 36 ? scan[lane] : 0u;
        const uint32_t inc = wave_inclusive_sum(cnt);
        const uint32_t exc = inc - cnt;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const uint32_t running = (uint32_t)__builtin_amdgcn_readlane((int)inc, 35);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)exc, 4 * t + wave_u);
            if ((masks[t] >> lane) & 1ull) {
                const uint32_t pos = before + lanes_below(masks[t]);
                *reinterpret_cast<omx_spectrogram_point*>(reinterpret_cast<char*>(out) + pos * 12u) = pts[t];
            }
        }
        if (j == 0) *count_out = running;
        mark(6);
    };
    column(sa, silent_a, col0, count_a);
    if (have1) column(sb, silent_b, col1, count_b);
}

void launch_stft_reassigned_4096_pair(const StftFastArgs& a, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return;
    const size_t lds = (size_t)(2 * FFT4096_LDS + 256) * sizeof(v2f) + 9 * 4 * sizeof(uint32_t) + 4 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_4096_pair_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    const uint32_t chunks = (a.n_cols + 1u) / 2u;
#ifdef OMX_TUNING
    static const bool phases = [] {
        const char* e = getenv("OMX_K2_VARIANT");
        return e && atoi(e) == 52;
    }();
    if (phases) {
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_4096_pair_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr2 = true;
        }
        hipLaunchKernelGGL(stft_reassigned_4096_pair_kernel<true>, dim3(stream_column_grid(a.n_streams, chunks)), dim3(256), lds, stream, a,
                           k2_phase_buffer());
        return;
    }
#endif
    hipLaunchKernelGGL(stft_reassigned_4096_pair_kernel<false>, dim3(stream_column_grid(a.n_streams, chunks)), dim3(256), lds, stream, a,
                       (unsigned long long*)nullptr);
}

}  // namespace omx
