// K2 for W = F = 8192 (H = 16384): one column per 256-thread workgroup, every 8192-point transform as ONE dual 4096-point
// transform of the tuned kernel plus a radix-2 step in registers (reference spectrogram/processor.rs:318-348, :439-488, :546-567).
//
// A thread holds 32 complex values of a transform.  Two index conventions alternate, so that no transform needs a transposition:
//   "interleaved"  thread j holds x[2 (j + 256 t) + r], r = 0 / 1, t < 16   (two neighbouring values per t: 16-byte loads)
//   "natural"      thread j holds X[j + 256 t'], t' < 32                    (t' = t + 16 q for the two halves q = 0 / 1)
// Decimation in time takes interleaved input to natural output:   F_r = FFT4096(x[2m + r]),  X[k + 4096 q] = F_0[k] + (-1)^q w^k F_1[k]
// Decimation in frequency takes natural input to interleaved output:  g_r[k] = (y[k] + (-1)^r y[k + 4096]) w^(-rk),  out[2n + r] = IFFT4096(g_r)[n]
// (w = exp(-2 pi i / 8192)).  The packed-real forward transform reads the ring interleaved and leaves the spectrum natural (what the
// Hilbert step's partner exchange wants); the inverse runs DIF and leaves (Im a[2m], Im a[2m+1]) interleaved; the analytic slice goes
// through LDS as floats (as in the 4096 kernel) and is read back interleaved; Z and Z2 run DIT again and leave their bins natural,
// which is the order the compaction needs.  Four dual transforms per column — the same as the 4096 kernel spends on a PAIR of columns,
// for the same number of bins (4097 against 2 x 2049): the size-templated kernel this replaces ran 512 threads per column with
// four-pass transforms and one workgroup per CU (11.2 M frames/s).
// Windows: the two-term cosine sums (Hann, Hamming), applied on the bins (see stft4096_pair_kernels.hip); other windows stay on
// the size-templated kernel.
#define OMX_FRAME_SYNC_LDS_ONLY 1  // this file's kernels exchange data between their threads through LDS only (fft_device.hpp: fft_sync)
#include "stft_kernels.hpp"

#include "buffer_device.hpp"
#include "fft_device.hpp"
#include "reassign_device.hpp"

#include <type_traits>

namespace omx {

__global__ __launch_bounds__(256, 2) void stft_reassigned_8192_kernel(StftFastArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* A = reinterpret_cast<v2f*>(smem_raw);
    v2f* B = A + FFT4096_LDS;                                      // (A and B are contiguous: the Hilbert step uses them as one 8704-slot buffer)
    v2f* tw2_lds = B + FFT4096_LDS;                                // [256] exp(-2 pi i k / 256)
    uint32_t* scan = reinterpret_cast<uint32_t*>(tw2_lds + 256);   // [9][4] wave counts of one half of the bins
    float* hil = reinterpret_cast<float*>(scan + 36);              // X[0]/2, X[8192]/2

    // XCD-aware map (same as block_to_stream_column): block b runs on XCD b % 8; stream s is pinned to XCD s % 8
    const uint32_t blk = blockIdx.x, xcd = blk & 7u, bq = blk >> 3;
    const uint32_t s = (bq / a.n_cols) * 8u + xcd, col = bq % a.n_cols;
    if (s >= a.n_streams) return;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const int lane = j & 63, wave = j >> 6;
    const float* ring = a.ring + (uint64_t)s * a.cap;
    const uint32_t mask32 = (uint32_t)(a.cap - 1);  // cap <= 2^30 (checked on the host)
    const long long last_nonzero = a.last_nonzero[s];
    const uint32_t n_cols_s = stft_cols(a, s);  // ragged banks: this stream's own column count
    const uint64_t tail_s = stft_tail(a, s);
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};
    if (col >= n_cols_s) return;
    const uint64_t p0 = tail_s + (uint64_t)col * a.hop;
    uint32_t* count_out = a.counts + (uint64_t)s * a.n_cols + col;
    if (last_nonzero < (long long)p0) {  // silent fast path (:307-316)
        if (j == 0) *count_out = 0;
        return;
    }
    const v2f* T8192 = a.tw4096;  // exp(-2 pi i k / 8192), k < 8192 (the host's table for W = 8192)

    // pass-3 twiddles exp(-2 pi i j t / 4096) = T8192[2 j t] are read at use (resident, they cost 30 registers this kernel does not have)
    struct TW {
        const v2f* tw2;
        GlobalBuffer t8192;
        unsigned j16;
        __device__ __forceinline__ v2f w2(unsigned k, int t) const { return tw2[k * (unsigned)t]; }
        __device__ __forceinline__ v2f w3(int t) const { return load_v2f(t8192, j16 * (unsigned)t, 0); }
    };
    const GlobalBuffer T8192b = global_buffer(T8192, 8192u * 8u), W16384b = global_buffer(a.tw8192, 8192u * 8u),
                       normb = global_buffer(a.bin_norm, 4097u * 4u);
    const TW tw{tw2_lds, T8192b, 16u * ju};
    const v2f tw2_mine = a.tw256[ju];

    // ---- 1. packed real FFT of the 16384-sample window: z[m] = (x[2m], x[2m+1]), m = 2 (j + 256 t) + r ------------------------
    const uint32_t p32 = (uint32_t)p0;
    v2f v0[16], v1[16];
    // the 16384 samples of the window lie in one piece of the ring (no wrap) and pairs are 8-byte aligned: buffer loads off one
    // lane offset; otherwise every index is wrapped by the mask
    const uint32_t off0 = p32 & mask32;
    const bool direct = (uint64_t)off0 + 16384ull <= a.cap && (p0 & 1ull) == 0;
    const GlobalBuffer windowb = global_buffer(ring + off0, 16384u * 4u);
    if (direct) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            v0[t] = load_v2f(windowb, ju * 16u, 4096u * (unsigned)t);
            v1[t] = load_v2f(windowb, ju * 16u + 8u, 4096u * (unsigned)t);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t q = p32 + 4u * (ju + 256u * (unsigned)t);
            v0[t] = v2f{ring[q & mask32], ring[(q + 1u) & mask32]};
            v1[t] = v2f{ring[(q + 2u) & mask32], ring[(q + 3u) & mask32]};
        }
    }
    v2f wk[16];  // w^k, k = j + 256 t: the radix-2 twiddles of every transform of this column
#pragma unroll
    for (int t = 0; t < 16; ++t) wk[t] = load_v2f(T8192b, ju * 8u, 2048u * (unsigned)t);
    tw2_lds[j] = tw2_mine;  // first read in pass 2 of the first transform, behind that transform's pass-1 barrier
    fft4096t_dual<false>(v0, v1, A, B, j, tw);  // F_0[k], F_1[k], k = j + 256 t
    v2f xl[16], xh[16];                          // Zf[k], Zf[k + 4096]
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const v2f m = cmul(v1[t], wk[t]);
        xl[t] = v0[t] + m;
        xh[t] = v0[t] - m;
    }

    // ---- 2. Hilbert transform with ONE half-length inverse (derivation: stft_kernels.hip step 2, N = 8192) -----------------------
    fft_sync();  // pass 3 of the dual transform still reads A and B
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        A[pad16(j + 256 * t)] = xl[t];
        A[pad16(j + 256 * t + 4096)] = xh[t];
    }
    if (j == 0) {
        hil[0] = (xl[0].x + xl[0].y) * 0.5f;  // X[0] / 2
        hil[1] = (xl[0].x - xl[0].y) * 0.5f;  // X[8192] / 2
    }
    fft_sync();
    v2f yl[16], yh[16];
    {
        // partner Zf[(8192 - k) & 8191] of k = j + 256 t' sits at pad16(8192 - j) - 272 t' (thread 0: 8704 - 272 t'; its t' = 0 read
        // lands one slot past the two buffers, inside the allocation, and is not used)
        const int part = (j ? pad16(8192 - j) : 8704) - 272 * 31;
        auto hilbert_half = [&](v2f (&y)[16], const v2f (&x)[16], int half) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int tp = t + 16 * half;
                const v2f w8 = load_v2f(W16384b, ju * 8u, 2048u * (unsigned)tp);  // exp(-2 pi i k / 16384) / 2
                const v2f z = x[t], zr = A[part + 272 * (31 - tp)];
                const v2f sum{z.x + zr.x, z.y - zr.y}, dif{z.x - zr.x, z.y + zr.y};
                y[t] = cmulc(sum, w8) - cmul(dif, w8);
                if (tp == 0 && j == 0) y[t] = v2f{0.0f, 0.0f};
            }
        };
        hilbert_half(yl, xl, 0);
        hilbert_half(yh, xh, 1);
    }
    const float half_x0 = hil[0], half_xn = hil[1];
    // inverse, decimation in frequency: g_0 = y[k] + y[k + 4096], g_1 = (y[k] - y[k + 4096]) conj(w^k)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const v2f lo = yl[t], hi = yh[t];
        v0[t] = lo + hi;
        v1[t] = cmulc(lo - hi, wk[t]);
    }
    fft_sync();  // partners are read from the buffers the inverse is about to overwrite
    fft4096t_dual<true>(v0, v1, A, B, j, tw);  // v_r[t] = (Im a[2m], Im a[2m+1]), m = 2 (j + 256 t) + r

    // ---- 3. analytic slice s[i] = analytic[4096 + i]: imaginary parts through LDS, read back interleaved -------------------------
    fft_sync();  // pass 3 of the inverse reads all over A and B
    float* imag = reinterpret_cast<float*>(A);  // 8192 floats
#pragma unroll
    for (int t = 4; t < 12; ++t) {  // samples 4 (j + 256 t) ... + 3 of the 16384-sample window; the slice is [4096, 12288)
        float4 q4{v0[t].x, v0[t].y, v1[t].x, v1[t].y};
        *reinterpret_cast<float4*>(imag + 4 * (j + 256 * (t - 4))) = q4;
    }
    v2f xr[16];  // the real part's samples x[p0 + 4096 + 2 (j + 256 t) + r]
    {
        const uint32_t qx = p32 + 4096u + 2u * ju;
        if (direct) {
#pragma unroll
            for (int t = 0; t < 16; ++t) xr[t] = load_v2f(windowb, ju * 8u, 16384u + 2048u * (unsigned)t);
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) xr[t] = v2f{ring[(qx + 512u * (unsigned)t) & mask32], ring[(qx + 512u * (unsigned)t + 1u) & mask32]};
        }
    }
    fft_sync();
    v2f s0[16], s1[16];  // s[2 (j + 256 t)], s[2 (j + 256 t) + 1]
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const v2f im = *reinterpret_cast<const v2f*>(imag + 2 * (j + 256 * t));
        // Re analytic[n] = 8192 x[n] - X[0]/2 + X[8192] (-1)^n / 2, n = 4096 + i: the parity of i = r
        s0[t] = v2f{8192.0f * xr[t].x - half_x0 + half_xn, im.x};
        s1[t] = v2f{8192.0f * xr[t].y - half_x0 - half_xn, im.y};
    }
    const float c0 = a.win_c0, half_c1 = 0.5f * a.win_c1, dscale = a.win_c1 * (3.14159265358979323846f / 8192.0f);
    v2f* lin_z = A;   // natural-order bins of Z: slot 1 + k, k = -1 ... 4097
    v2f* lin_z2 = B;  // natural-order bins of Z2

    // ---- 4. Z2 = FFT((n - 4095.5) s) first (its bins wait in LDS), then Z = FFT(s) -------------------------------------------------
    // bins k = j + 256 t (t < 16) come from the q = 0 half; bins 4096, 4097 (threads 0, 1) and -1 = 8191 (thread 255) from q = 1
    auto spectrum = [&](v2f (&f0)[16], v2f (&f1)[16], v2f* lin) {
        fft4096t_dual<false>(f0, f1, A, B, j, tw);
        v2f w[16];  // w^k again (not held across the transform: 32 registers it needs)
#pragma unroll
        for (int t = 0; t < 16; ++t) w[t] = load_v2f(T8192b, ju * 8u, 2048u * (unsigned)t);
        fft_sync();  // pass 3 still reads A and B
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const v2f m = cmul(f1[t], w[t]);
            lin[1 + j + 256 * t] = f0[t] + m;
            if (t == 0 && j < 2) lin[1 + 4096 + j] = f0[t] - m;  // bins 4096, 4097
            if (t == 15 && j == 255) lin[0] = f0[t] - m;         // bin 8191 = -1
        }
    };
    {
        v2f f0[16], f1[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float nc = (float)(2 * (j + 256 * t)) - 4095.5f;  // compute_time_weighted's ramp (:601-608)
            f0[t] = v2f{s0[t].x * nc, s0[t].y * nc};
            f1[t] = v2f{s1[t].x * (nc + 1.0f), s1[t].y * (nc + 1.0f)};
        }
        fft_sync();  // the gather above still reads A
        spectrum(f0, f1, lin_z2);
    }
    // (Z's pass 1 writes A and B: the Z2 copy in B must be consumed first -> t = FFT(t w s) is formed now and parked in registers)
    fft_sync();
    v2f bt[17];
#pragma unroll
    for (int t = 0; t < 17; ++t) {
        const int bin = j + 256 * t;  // (t = 16: only thread 0's bin exists; the others read slots inside the buffer and drop the result)
        const v2f z2m = lin_z2[bin], z2c = lin_z2[bin + 1], z2p = lin_z2[bin + 2];
        const v2f z2s{z2m.x + z2p.x, z2m.y + z2p.y};
        bt[t] = v2f{c0 * z2c.x + half_c1 * z2s.x, c0 * z2c.y + half_c1 * z2s.y};
    }
    fft_sync();  // those reads are done before Z's pass 1 writes
    spectrum(s0, s1, lin_z);
    fft_sync();

    // ---- 5. reassignment + ordered compaction (bins j + 256 t, t < 16, and bin 4096 on thread 0), in two halves of the bins ---------
    omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    uint32_t emitted = 0;
    auto half = [&](auto first_c, auto count_c) {
        constexpr int T0 = decltype(first_c)::value, NT = decltype(count_c)::value;
        omx_spectrogram_point pts[NT];
        unsigned long long masks[NT];
        float pn[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) pn[u] = load_f32(normb, ju * 4u, 1024u * (unsigned)(T0 + u));  // (t = 16, j > 0: past the table, reads 0, not used)
#pragma unroll
        for (int h = 0; h < NT; h += 3) {  // three bins at a time: the neighbour reads of a group are issued together
            v2f nzm[3], nzc[3], nzp[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (h + u >= NT) continue;
                const int bin = j + 256 * (T0 + h + u);
                nzm[u] = lin_z[bin];
                nzc[u] = lin_z[bin + 1];
                nzp[u] = lin_z[bin + 2];
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = h + u;
                if (i >= NT) continue;
                const int t = T0 + i;
                const uint32_t bin = (uint32_t)(j + 256 * t);
                const v2f zm = nzm[u], zp = nzp[u], zc = nzc[u];
                const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y};
                const v2f bb{c0 * zc.x + half_c1 * zs.x, c0 * zc.y + half_c1 * zs.y};
                const v2f bd{-dscale * zd.y, dscale * zd.x};  // i c1 (pi / W) (Z[k-1] - Z[k+1])
                const bool keep = reassign_flat(bin, bb, bd, bt[t], pn[i], rc, pts[i]) && (t < 16 || j == 0);
                masks[i] = __ballot(keep);
                if (lane == 0) scan[i * 4 + wave] = (uint32_t)__popcll(masks[i]);
            }
        }
        fft_sync();
        // exclusive prefix of this half's wave counts ([t][wave] row-major = bin order)
        const uint32_t cnt = lane < 4 * NT ? scan[lane] : 0u;
        const uint32_t inc = wave_inclusive_sum(cnt);
        const uint32_t exc = inc - cnt;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const uint32_t before = emitted + (uint32_t)__builtin_amdgcn_readlane((int)exc, 4 * i + wave_u);
            if ((masks[i] >> lane) & 1ull)
                *reinterpret_cast<omx_spectrogram_point*>(reinterpret_cast<char*>(out) + (before + lanes_below(masks[i])) * 12u) = pts[i];
        }
        emitted += (uint32_t)__builtin_amdgcn_readlane((int)inc, 4 * NT - 1);
    };
    half(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
    fft_sync();  // the wave counts are rewritten
    half(std::integral_constant<int, 8>{}, std::integral_constant<int, 9>{});
    if (j == 0) *count_out = emitted;
}

void launch_stft_reassigned_8192(const StftFastArgs& a, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return;
    const size_t lds = (size_t)(2 * FFT4096_LDS + 256) * sizeof(v2f) + 9 * 4 * sizeof(uint32_t) + 4 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_8192_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(stft_reassigned_8192_kernel, dim3(stream_column_grid(a.n_streams, a.n_cols)), dim3(256), lds, stream, a);
}

}  // namespace omx
