// 16384-point transforms of the reassigned path (window 16384, or 1024 ... 8192 zero-padded to 16384) as FOUR 4096-point transforms
// of the tuned kernel (two dual transforms) plus a radix-4 step in registers: 256 threads per column and two workgroups per CU,
// where the size-templated kernels ran 1024 threads per column, four-pass in-place transforms and one workgroup per CU
// (22 us per transform and workgroup; reference spectrogram/processor.rs:318-348, :439-488, :546-567).
//
// A thread holds 64 complex values of a transform, in one of two index conventions (cf. stft8192_kernels.hip):
//   "interleaved"  x[4 (j + 256 t) + r], r < 4, t < 16     (four neighbouring values per t: 32-byte loads / stores)
//   "natural"      X[j + 256 t'], t' = t + 16 q < 64
// Decimation in time, interleaved -> natural:   F_r = FFT4096(x[4m + r]),  X[k + 4096 q] = sum_r (-i)^(rq) w^(rk) F_r[k]
// Decimation in frequency, natural -> interleaved:  g_r[k] = (sum_q (+i)^(rq) y[k + 4096 q]) w^(-rk),  out[4n + r] = IFFT4096(g_r)[n]
// (w = exp(-2 pi i / 16384)).  The analytic slice and the spectra still travel between the kernels through the HBM scratch
// (launch_big in stft_pow2_kernels.hip): 128 registers hold ONE transform's values, so Z and Z2 cannot both stay resident.
#define OMX_FRAME_SYNC_LDS_ONLY 1  // this file's kernels exchange data between their threads through LDS only (fft_device.hpp: fft_sync)
#include "stft_kernels.hpp"

#include <type_traits>

#include "buffer_device.hpp"
#include "fft_device.hpp"
#include "reassign_device.hpp"
#include "twiddle_run_device.hpp"

namespace omx {

namespace {

// pass-2 twiddles from the LDS copy, pass-3 twiddles exp(-+2 pi i j t / 4096) = T16384[4 j t] read at use
struct Tw16k {
    const v2f* tw2;
    GlobalBuffer t16384;
    unsigned j32;  // 4 j * 8 bytes
    __device__ __forceinline__ v2f w2(unsigned k, int t) const { return tw2[k * (unsigned)t]; }
    __device__ __forceinline__ v2f w3(int t) const { return load_v2f(t16384, j32 * (unsigned)t, 0); }
};

// w^(r k), k = j + 256 t, r = 1, 2, 3, as w^(r j) (three table reads per thread, once) times exp(-2 pi i r t / 64) (compile-time): a table read
// per element is a global round trip per t once the registers are full, and these kernels live at the register limit
struct TwRun {
    v2f b1, b2, b3;  // w^j, w^(2j), w^(3j)
    __device__ __forceinline__ void load(const GlobalBuffer& T, unsigned ju) {
        b1 = load_v2f(T, ju * 8u, 0);
        b2 = load_v2f(T, ju * 16u, 0);
        b3 = load_v2f(T, ju * 24u, 0);
    }
    template <int TT>
    __device__ __forceinline__ void at(v2f& w1, v2f& w2, v2f& w3) const {
        w1 = rotate128<2 * TT>(b1);
        w2 = rotate128<4 * TT>(b2);
        w3 = rotate128<6 * TT>(b3);
    }
};

template <int NQ, int TT>
__device__ __forceinline__ void dit4_one(v2f (&f0)[16], v2f (&f1)[16], v2f (&f2)[16], v2f (&f3)[16], const TwRun& run) {
    constexpr int t = TT;
    v2f w1, w2, w3;
    run.at<TT>(w1, w2, w3);
    v2f a0 = f0[t], a1 = cmul(f1[t], w1), a2 = cmul(f2[t], w2), a3 = cmul(f3[t], w3);
    if constexpr (NQ == 4) {
        dft4<false>(a0, a1, a2, a3);
        f0[t] = a0;
        f1[t] = a1;
        f2[t] = a2;
        f3[t] = a3;
    } else {  // q = 0, 1 only (bins 0 ... 8191)
        const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
        f0[t] = t0 + t2;
        f1[t] = add_rot<false>(t1, d);
        if (t == 0) f2[t] = t0 - t2;                 // bins 8192 + j (the halo above N/2: threads 0 ... 16)
        if (t == 15) f3[t] = sub_rot<false>(t1, d);  // bins 12288 + 3840 + j (the halo below 0: threads 240 ... 255)
    }
}
// radix-4 step of the decimation in time, outputs q = 0 ... NQ-1 (f0 ... f3 in, X[k + 4096 q] out in place of f_q)
template <int NQ, int TT = 0>
__device__ __forceinline__ void dit4_combine(v2f (&f0)[16], v2f (&f1)[16], v2f (&f2)[16], v2f (&f3)[16], const TwRun& run) {
    if constexpr (TT < 16) {
        dit4_one<NQ, TT>(f0, f1, f2, f3, run);
        dit4_combine<NQ, TT + 1>(f0, f1, f2, f3, run);
    }
}
// radix-4 step of the decimation in frequency (inverse): g_r = (sum_q (+i)^(rq) y_q) conj(w^(r k))
template <int TT = 0>
__device__ __forceinline__ void dif4_split(v2f (&f0)[16], v2f (&f1)[16], v2f (&f2)[16], v2f (&f3)[16], const TwRun& run) {
    if constexpr (TT < 16) {
        v2f w1, w2, w3;
        run.at<TT>(w1, w2, w3);
        v2f a0 = f0[TT], a1 = f1[TT], a2 = f2[TT], a3 = f3[TT];
        dft4<true>(a0, a1, a2, a3);
        f0[TT] = a0;
        f1[TT] = cmulc(a1, w1);
        f2[TT] = cmulc(a2, w2);
        f3[TT] = cmulc(a3, w3);
        dif4_split<TT + 1>(f0, f1, f2, f3, run);
    }
}

template <int Q, int TT = 0, class F>
__device__ __forceinline__ void hilbert_run(v2f (&x)[16], const v2f (&zr)[16], v2f w8_base, const F& hilbert) {
    if constexpr (TT < 16) {
        x[TT] = hilbert(x[TT], zr[TT], rotate128<TT + 16 * Q>(w8_base));
        hilbert_run<Q, TT + 1>(x, zr, w8_base, hilbert);
    }
}

}  // namespace

// ================================================================================================
// hilbert_16k_kernel: packed real FFT of the 32768-sample window, single-inverse Hilbert, analytic slice -> sv[frame][16384]
// ================================================================================================
// imag_only: the slice row holds the 16384 imaginary parts as floats, then X[0]/2 and X[16384]/2 (the consumer rebuilds the real
// parts from the ring: windowed_reassign_16k_kernel); otherwise complex values (windowed_16k_kernel)
__global__ __launch_bounds__(256, 2) void hilbert_16k_kernel(StftFastArgs a, BigScratch sc, int imag_only) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* A = reinterpret_cast<v2f*>(smem_raw);
    v2f* B = A + FFT4096_LDS;
    v2f* tw2_lds = B + FFT4096_LDS;  // [256]
    const uint32_t item = sc.first + blockIdx.x;
    const uint32_t s = item / a.n_cols, col = item % a.n_cols;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const uint64_t p0 = stft_tail(a, s) + (uint64_t)col * a.hop;
    if (col >= stft_cols(a, s) || a.last_nonzero[s] < (long long)p0) return;  // past the stream's count / silent column: reassign_big_kernel emits it empty
    const float* ring = a.ring + (uint64_t)s * a.cap;
    const uint32_t mask32 = (uint32_t)(a.cap - 1);
    const uint32_t p32 = (uint32_t)p0;
    const GlobalBuffer T = global_buffer(a.tw4096, 16384u * 8u), W8 = global_buffer(a.tw8192, 16384u * 8u);
    const Tw16k tw{tw2_lds, T, 32u * ju};
    tw2_lds[j] = a.tw256[ju];

    // ---- forward: z[m] = (x[2m], x[2m+1]), thread j reads m = 4 (j + 256 t) + r: 8 consecutive samples per t ---------------------------
    const uint32_t off0 = p32 & mask32;
    const bool direct = (uint64_t)off0 + 32768ull <= a.cap && (p0 & 1ull) == 0;
    const GlobalBuffer window = global_buffer(ring + off0, 32768u * 4u);
    auto load_pair = [&](v2f (&x0)[16], v2f (&x1)[16], int r0) {  // sub-sequences r0, r0 + 1
        if (direct) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                x0[t] = load_v2f(window, ju * 32u + 8u * (unsigned)r0, 8192u * (unsigned)t);
                x1[t] = load_v2f(window, ju * 32u + 8u * (unsigned)r0 + 8u, 8192u * (unsigned)t);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint32_t q = p32 + 8u * (ju + 256u * (unsigned)t) + 2u * (unsigned)r0;
                x0[t] = v2f{ring[q & mask32], ring[(q + 1u) & mask32]};
                x1[t] = v2f{ring[(q + 2u) & mask32], ring[(q + 3u) & mask32]};
            }
        }
    };
    v2f f0[16], f1[16], f2[16], f3[16];
    load_pair(f0, f1, 0);
    fft4096t_dual<false>(f0, f1, A, B, j, tw);
    load_pair(f2, f3, 2);
    fft_sync();  // pass 3 of the previous dual still reads A and B
    fft4096t_dual<false>(f2, f3, A, B, j, tw);
    TwRun run;
    run.load(T, ju);
    dit4_combine<4>(f0, f1, f2, f3, run);  // f_q[t] = Zf[j + 256 t + 4096 q]

    // ---- Hilbert spectrum: y[k] from Zf[k], Zf[(16384 - k) & 16383] -----------------------------------------------------------------------
    // Quarter q's partners lie in quarter 3 - q, at offset 4096 - (j + 256 t) = slot pad16(4096 - j) - 272 t of that quarter's copy
    // (thread 0: 4352 - 272 t), so the quarters cross LDS in two rounds: (0 in A, 3 in B), then (1 in A, 2 in B).  The four elements
    // k = 0, 4096, 8192, 12288 (thread 0, t = 0) pair among themselves and are handled in registers.
    const v2f top1 = f1[0], top2 = f2[0], top3 = f3[0];            // (only thread 0's are used)
    float* hil = reinterpret_cast<float*>(tw2_lds + 256);
    if (j == 0) {
        hil[0] = (f0[0].x + f0[0].y) * 0.5f;  // X[0] / 2
        hil[1] = (f0[0].x - f0[0].y) * 0.5f;  // X[16384] / 2
    }
    const v2f w8_base = load_v2f(W8, ju * 8u, 0);  // exp(-2 pi i j / 32768) / 2; element k = j + 256 t' takes it times exp(-2 pi i t' / 128)
    auto hilbert = [&](v2f z, v2f zr, v2f w8) {
        const v2f sum{z.x + zr.x, z.y - zr.y}, dif{z.x - zr.x, z.y + zr.y};
        return cmulc(sum, w8) - cmul(dif, w8);
    };
    const int part = (j ? pad16(4096 - j) : 4352) - 272 * 15;
    auto exchange = [&](v2f (&lo)[16], v2f (&hi)[16], auto q_lo_c, auto q_hi_c, v2f self_lo, v2f self_hi) {
        constexpr int Q_LO = decltype(q_lo_c)::value, Q_HI = decltype(q_hi_c)::value;
        fft_sync();  // earlier reads of A and B are done
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            A[pad16(j + 256 * t)] = lo[t];
            B[pad16(j + 256 * t)] = hi[t];
        }
        fft_sync();
        v2f zr_lo[16], zr_hi[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            zr_lo[t] = B[part + 272 * (15 - t)];
            zr_hi[t] = A[part + 272 * (15 - t)];
        }
        if (j == 0) {
            zr_lo[0] = self_lo;
            zr_hi[0] = self_hi;
        }
        hilbert_run<Q_LO>(lo, zr_lo, w8_base, hilbert);
        hilbert_run<Q_HI>(hi, zr_hi, w8_base, hilbert);
    };
    exchange(f0, f3, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, f0[0], top1);  // k = 0: its own partner (forced to 0 below); k = 12288 <-> 4096
    exchange(f1, f2, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, top3, top2);   // k = 4096 <-> 12288; k = 8192: its own partner
    if (j == 0) f0[0] = v2f{0.0f, 0.0f};

    // ---- inverse, decimation in frequency: g_r[k] = (sum_q (+i)^(rq) y[k + 4096 q]) conj(w^(rk)), four 4096-point inverses ----------------
    dif4_split(f0, f1, f2, f3, run);
    const float hx0 = hil[0], hxn = hil[1];
    fft_sync();  // the partner reads are done
    fft4096t_dual<true>(f0, f1, A, B, j, tw);
    fft_sync();
    fft4096t_dual<true>(f2, f3, A, B, j, tw);  // f_r[t] = (Im a[2m], Im a[2m+1]), m = 4 (j + 256 t) + r

    // ---- analytic slice: samples 8 (j + 256 t) ... + 7 of the window, t = 4 ... 11, are slice elements i = 8 (j + 256 (t - 4)) ... + 7 ------
    v2f* out = sc.sv + (uint64_t)blockIdx.x * 16384u;
    if (imag_only) {
        float* outf = reinterpret_cast<float*>(out);
        if (j == 0) {
            outf[16384] = hx0;
            outf[16385] = hxn;
        }
#pragma unroll
        for (int t = 4; t < 12; ++t) {
            float4* dst = reinterpret_cast<float4*>(outf + 8 * (j + 256 * (t - 4)));
            dst[0] = float4{f0[t].x, f0[t].y, f1[t].x, f1[t].y};
            dst[1] = float4{f2[t].x, f2[t].y, f3[t].x, f3[t].y};
        }
        return;
    }
#pragma unroll
    for (int t = 4; t < 12; ++t) {
        float xr[8];
        if (direct) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const v2f x2 = load_v2f(window, ju * 32u + 8u * (unsigned)e, 8192u * (unsigned)t);
                xr[2 * e] = x2.x;
                xr[2 * e + 1] = x2.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[e] = ring[(p32 + 8u * (ju + 256u * (unsigned)t) + (unsigned)e) & mask32];
        }
        // Re analytic[n] = 16384 x[n] - X[0]/2 + X[16384] (-1)^n / 2
        const v2f im[4] = {f0[t], f1[t], f2[t], f3[t]};
        float4* dst = reinterpret_cast<float4*>(out + 8 * (j + 256 * (t - 4)));
#pragma unroll
        for (int r = 0; r < 4; ++r)
            dst[r] = float4{16384.0f * xr[2 * r] - hx0 + hxn, im[r].x, 16384.0f * xr[2 * r + 1] - hx0 - hxn, im[r].y};
    }
}

// ================================================================================================
// windowed_16k_kernel: (frame, q) -> one 16384-point transform of the windowed / ramped analytic slice, bins 0 ... 8192 (+ halo)
// ================================================================================================
__global__ __launch_bounds__(256, 2) void windowed_16k_kernel(StftFastArgs a, BigScratch sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* A = reinterpret_cast<v2f*>(smem_raw);
    v2f* B = A + FFT4096_LDS;
    v2f* tw2_lds = B + FFT4096_LDS;
    const uint32_t item = sc.first + blockIdx.x, q = blockIdx.y;
    const uint32_t s = item / a.n_cols, col = item % a.n_cols;
    if (col >= stft_cols(a, s) || a.last_nonzero[s] < (long long)(stft_tail(a, s) + (uint64_t)col * a.hop)) return;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const GlobalBuffer T = global_buffer(a.tw4096, 16384u * 8u);
    const Tw16k tw{tw2_lds, T, 32u * ju};
    tw2_lds[j] = a.tw256[ju];
    const uint32_t W = a.window_size;  // == 16384 unless the window is zero-padded to the transform (:334-342)
    const bool bins = a.win_terms == 2;  // Hann / Hamming: q = 0 -> Z = FFT(s), q = 1 -> Z2 = FFT((n - c) s); the window is applied on the
                                         // bins by reassign_big_kernel (see stft4096_pair_kernels.hip)
    // reads past the window return 0: that IS the zero padding (:559-567)
    const GlobalBuffer slice = global_buffer(sc.sv + (uint64_t)blockIdx.x * W, W * 8u);
    const GlobalBuffer win = global_buffer(q == 1 ? a.dwindow : a.window, W * 4u);
    const float center = (float)(W - 1u) * 0.5f;
    auto load_pair = [&](v2f (&x0)[16], v2f (&x1)[16], int r0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = r0 + e;
                const v2f x = load_v2f(slice, ju * 32u + 8u * (unsigned)r, 8192u * (unsigned)t);
                const float i = (float)(4 * (j + 256 * t) + r);
                float w = bins ? 1.0f : load_f32(win, ju * 16u + 4u * (unsigned)r, 4096u * (unsigned)t);
                if (bins ? q == 1 : q == 2) w = (i - center) * w;  // compute_time_weighted (:601-608)
                (e ? x1 : x0)[t] = v2f{x.x * w, x.y * w};
            }
        }
    };
    v2f f0[16], f1[16], f2[16], f3[16];
    load_pair(f0, f1, 0);
    fft4096t_dual<false>(f0, f1, A, B, j, tw);
    load_pair(f2, f3, 2);
    fft_sync();  // pass 3 of the previous dual still reads A and B
    fft4096t_dual<false>(f2, f3, A, B, j, tw);
    TwRun run;
    run.load(T, ju);
    dit4_combine<2>(f0, f1, f2, f3, run);  // f0[t] = X[j + 256 t], f1[t] = X[4096 + j + 256 t]; f2[0] = X[8192 + j]; f3[15] = X[16128 + j]
    if (bins) {  // bins -kBigHalo ... 8192 + kBigHalo (row slot = bin + kBigHalo): the window's cosine shifts by F / W <= 16 bins
        v2f* out = sc.spec + ((uint64_t)q * sc.count + blockIdx.x) * kBigRow<14> + kBigHalo;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            out[j + 256 * t] = f0[t];
            out[4096 + j + 256 * t] = f1[t];
        }
        if (j <= kBigHalo) out[8192 + j] = f2[0];
        if (j >= 256 - kBigHalo) out[j - 256] = f3[15];  // bin 16384 - m sits at slot -m
        return;
    }
    v2f* out = sc.spec + ((uint64_t)q * sc.count + blockIdx.x) * 8193u;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        out[j + 256 * t] = f0[t];
        out[4096 + j + 256 * t] = f1[t];
    }
    if (j == 0) out[8192] = f2[0];
}

// ================================================================================================
// windowed_reassign_16k_kernel (two-term cosine windows): per column, Z2 = FFT((n - c) s) and Z = FFT(s) one after the other, the
// window applied on the bins, reassignment and ordered compaction — what windowed_16k_kernel x 2 + reassign_big_kernel did through
// 264 KB of spectra in HBM per column.  A spectrum's bins 0 ... 8192 (+ the halo the window's shift needs) cross LDS as ONE natural-
// order copy in both buffers (8225 slots); FFT(t w s) waits for the second transform in a 64 KB global row that only this workgroup
// touches.  IMAG: the slice row holds imaginary parts (hilbert_16k_kernel, W = 16384); otherwise complex values of a window
// W < 16384 that is zero-padded to the transform (reads past the window return 0).
// ================================================================================================
template <bool IMAG>
__global__ __launch_bounds__(256, 2) void windowed_reassign_16k_kernel(StftFastArgs a, BigScratch sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* A = reinterpret_cast<v2f*>(smem_raw);
    v2f* B = A + FFT4096_LDS;
    v2f* tw2_lds = B + FFT4096_LDS;
    uint32_t* scan = reinterpret_cast<uint32_t*>(tw2_lds + 256);  // [9][4]
    v2f* lin = A + kBigHalo;  // natural-order bins -16 ... 8208 over both buffers
    const uint32_t item = sc.first + blockIdx.x;
    const uint32_t s = item / a.n_cols, col = item % a.n_cols;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const int lane = j & 63, wave = j >> 6;
    uint32_t* count_out = a.counts + (uint64_t)s * a.n_cols + col;
    const uint64_t p0 = stft_tail(a, s) + (uint64_t)col * a.hop;
    if (col >= stft_cols(a, s) || a.last_nonzero[s] < (long long)p0) {  // past the stream's count / silent column (:307-316)
        if (j == 0) *count_out = 0;
        return;
    }
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};
    const GlobalBuffer T = global_buffer(a.tw4096, 16384u * 8u), normb = global_buffer(a.bin_norm, 8193u * 4u);
    const Tw16k tw{tw2_lds, T, 32u * ju};
    tw2_lds[j] = a.tw256[ju];
    TwRun run;
    run.load(T, ju);
    const uint32_t W = a.window_size;  // == 16384 unless the window is zero-padded to the transform (:334-342)
    const float center = (float)(W - 1u) * 0.5f;
    // slice element i = 4 (j + 256 t) + r
    const float* ring = a.ring + (uint64_t)s * a.cap;
    const uint32_t mask32 = (uint32_t)(a.cap - 1);
    const uint32_t p32 = (uint32_t)p0;
    const float* rowf = reinterpret_cast<const float*>(sc.sv + (uint64_t)blockIdx.x * W);
    const GlobalBuffer slice = global_buffer(rowf, IMAG ? 16384u * 4u : W * 8u);
    const uint32_t off_s = (p32 + 8192u) & mask32;  // the slice's first sample (IMAG: W = 16384, slice = samples 8192 ... 24575 of the window)
    const bool direct = (uint64_t)off_s + 16384ull <= a.cap && (p0 & 1ull) == 0;
    const GlobalBuffer real = global_buffer(ring + off_s, 16384u * 4u);
    float hx0 = 0.0f, hxn = 0.0f;
    if constexpr (IMAG) {
        hx0 = rowf[16384];
        hxn = rowf[16385];
    }
    auto load_pair = [&](v2f (&x0)[16], v2f (&x1)[16], int r0, bool ramp) {  // elements 4 (j + 256 t) + r0, + r0 + 1
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            v2f e0, e1;
            if constexpr (IMAG) {
                const v2f im = load_v2f(slice, ju * 16u + 4u * (unsigned)r0, 4096u * (unsigned)t);
                v2f re;
                if (direct) re = load_v2f(real, ju * 16u + 4u * (unsigned)r0, 4096u * (unsigned)t);
                else {
                    const uint32_t qx = p32 + 8192u + 4u * (ju + 256u * (unsigned)t) + (unsigned)r0;
                    re = v2f{ring[qx & mask32], ring[(qx + 1u) & mask32]};
                }
                // Re analytic[n] = 16384 x[n] - X[0]/2 + X[16384] (-1)^n / 2; the parity of n = 8192 + i is that of r0 (even here)
                e0 = v2f{16384.0f * re.x - hx0 + hxn, im.x};
                e1 = v2f{16384.0f * re.y - hx0 - hxn, im.y};
            } else {
                e0 = load_v2f(slice, ju * 32u + 8u * (unsigned)r0, 8192u * (unsigned)t);
                e1 = load_v2f(slice, ju * 32u + 8u * (unsigned)r0 + 8u, 8192u * (unsigned)t);
            }
            if (ramp) {  // compute_time_weighted's ramp (:601-608)
                const float n0 = (float)(4 * (j + 256 * t) + r0) - center;
                e0 = v2f{e0.x * n0, e0.y * n0};
                e1 = v2f{e1.x * (n0 + 1.0f), e1.y * (n0 + 1.0f)};
            }
            x0[t] = e0;
            x1[t] = e1;
        }
    };
    // one transform of the slice, bins -16 ... 8208 left in `lin`
    auto spectrum = [&](bool ramp) {
        v2f f0[16], f1[16], f2[16], f3[16];
        load_pair(f0, f1, 0, ramp);
        fft4096t_dual<false>(f0, f1, A, B, j, tw);
        load_pair(f2, f3, 2, ramp);
        fft_sync();  // pass 3 of the previous dual still reads A and B
        fft4096t_dual<false>(f2, f3, A, B, j, tw);
        dit4_combine<2>(f0, f1, f2, f3, run);  // f0[t] = X[j + 256 t], f1[t] = X[4096 + j + 256 t]; f2[0] = X[8192 + j]; f3[15] = X[16128 + j]
        fft_sync();  // pass 3 still reads A and B
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            lin[j + 256 * t] = f0[t];
            lin[4096 + j + 256 * t] = f1[t];
        }
        if (j <= kBigHalo) lin[8192 + j] = f2[0];
        if (j >= 256 - kBigHalo) lin[j - 256] = f3[15];  // bin 16384 - m sits at -m
        fft_sync();
    };
    // window on the bins: cos(2 pi n / W) shifts an F-point spectrum by F / W bins
    const int shift = (int)(16384u / W);
    const float c0 = a.win_c0, half_c1 = 0.5f * a.win_c1, dscale = a.win_c1 * (3.14159265358979323846f / (float)W);
    v2f* park = sc.spec + (uint64_t)blockIdx.x * kBigRow<14>;  // FFT(t w s) of bins 0 ... 8192, written and read by this workgroup only

    fft_sync();  // tw2_lds
    spectrum(true);
#pragma unroll
    for (int t = 0; t < 33; ++t) {
        if (t == 32 && j != 0) break;
        const int bin = j + 256 * t;
        const v2f zc = lin[bin], zm = lin[bin - shift], zp = lin[bin + shift];
        park[bin] = v2f{c0 * zc.x + half_c1 * (zm.x + zp.x), c0 * zc.y + half_c1 * (zm.y + zp.y)};
    }
    fft_sync();  // those reads are done before the next transform's pass 1 writes
    spectrum(false);

    // reassignment + ordered compaction, eight (the last time nine) bins per thread at a time
    omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    uint32_t emitted = 0;
    auto chunk = [&](auto first_c, auto count_c) {
        constexpr int T0 = decltype(first_c)::value, NT = decltype(count_c)::value;
        omx_spectrogram_point pts[NT];
        unsigned long long masks[NT];
        v2f bt[NT];
        float pn[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const bool mine = T0 + u < 32 || j == 0;
            bt[u] = park[mine ? j + 256 * (T0 + u) : 0];
            pn[u] = load_f32(normb, ju * 4u, 1024u * (unsigned)(T0 + u));  // (t = 32, j > 0: past the table, reads 0, not used)
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int t = T0 + u;
            const int bin = j + 256 * t;  // (t = 32: only thread 0's bin exists; the others read slots inside the buffer and drop the result)
            const v2f zc = lin[bin], zm = lin[bin - shift], zp = lin[bin + shift];
            const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y};
            const v2f bb{c0 * zc.x + half_c1 * zs.x, c0 * zc.y + half_c1 * zs.y};
            const v2f bd{-dscale * zd.y, dscale * zd.x};  // i c1 (pi / W) (Z[k - F/W] - Z[k + F/W])
            const bool keep = reassign_flat((uint32_t)bin, bb, bd, bt[u], pn[u], rc, pts[u]) && (t < 32 || j == 0);
            masks[u] = __ballot(keep);
            if (lane == 0) scan[u * 4 + wave] = (uint32_t)__popcll(masks[u]);
        }
        fft_sync();
        const uint32_t cnt = lane < 4 * NT ? scan[lane] : 0u;
        const uint32_t inc = wave_inclusive_sum(cnt);
        const uint32_t exc = inc - cnt;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const uint32_t before = emitted + (uint32_t)__builtin_amdgcn_readlane((int)exc, 4 * u + wave_u);
            if ((masks[u] >> lane) & 1ull)
                *reinterpret_cast<omx_spectrogram_point*>(reinterpret_cast<char*>(out) + (before + lanes_below(masks[u])) * 12u) = pts[u];
        }
        emitted += (uint32_t)__builtin_amdgcn_readlane((int)inc, 4 * NT - 1);
        fft_sync();  // the wave counts are rewritten by the next chunk
    };
    chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
    chunk(std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{});
    chunk(std::integral_constant<int, 16>{}, std::integral_constant<int, 8>{});
    chunk(std::integral_constant<int, 24>{}, std::integral_constant<int, 9>{});
    if (j == 0) *count_out = emitted;
}

static size_t lds_16k() { return (size_t)(2 * FFT4096_LDS + 256) * sizeof(v2f) + 9 * 4 * sizeof(uint32_t) + 4 * sizeof(float); }
void launch_hilbert_16k(const StftFastArgs& a, const BigScratch& sc, bool imag_only, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hilbert_16k_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_16k());
        attr_set = true;
    }
    hipLaunchKernelGGL(hilbert_16k_kernel, dim3(sc.count), dim3(256), lds_16k(), stream, a, sc, imag_only ? 1 : 0);
}
void launch_windowed_16k(const StftFastArgs& a, const BigScratch& sc, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(windowed_16k_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_16k());
        attr_set = true;
    }
    hipLaunchKernelGGL(windowed_16k_kernel, dim3(sc.count, a.win_terms == 2 ? 2 : 3), dim3(256), lds_16k(), stream, a, sc);
}

// two-term cosine windows: both transforms, the window on the bins, reassignment and compaction in one kernel
void launch_windowed_reassign_16k(const StftFastArgs& a, const BigScratch& sc, bool imag_only, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(windowed_reassign_16k_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_16k());
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(windowed_reassign_16k_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_16k());
        attr_set = true;
    }
    if (imag_only) hipLaunchKernelGGL(windowed_reassign_16k_kernel<true>, dim3(sc.count), dim3(256), lds_16k(), stream, a, sc);
    else hipLaunchKernelGGL(windowed_reassign_16k_kernel<false>, dim3(sc.count), dim3(256), lds_16k(), stream, a, sc);
}

}  // namespace omx
