// K2p: the fused reassigned STFT of stft_kernels.hip, templated on the window length W = N = 1024, 2048 (and 4096, kept as a
// cross-check of the tuned 4096 kernel).  A frame is carried by T = N / 16 threads, so a 256-thread workgroup processes
// 256 / T consecutive columns of one stream side by side, each with its own pair of LDS buffers.  Same algorithm, same
// operation order per bin (reference src/visuals/spectrogram/processor.rs:318-348, 439-488, 546-567):
//   packed real FFT of the 2N-sample window -> Hilbert with one half-length inverse -> analytic slice ->
//   three windowed FFTs (w, w', t w) -> per-bin reassignment -> ordered compaction.
#include <mutex>
#include <cstdlib>

#include "buffer_device.hpp"
#define OMX_FRAME_SYNC_LDS_ONLY 1  // every kernel of this file exchanges data between its threads through LDS only (global scratch is
                                   // written by one kernel and read by the next): fft_pow2_device.hpp, frame_sync
#include "fft_pow2_device.hpp"
#include "reassign_device.hpp"
#include "twiddle_run_device.hpp"
#include "stft_kernels.hpp"

namespace omx {

// ordered compaction, shared by the kernels of this file: `masks[t]` = ballot of the kept bins jf + T t of this wavefront, `scan` =
// this frame's [9][WPF] wave counts in LDS (written before the frame barrier).  The exclusive prefix over the counts ([t][wave]
// row-major = bin order) is taken by every wavefront for itself with six DPP adds and read per t with v_readlane; returns the
// column's point count.  WPF = 8 (8192 points) has 72 counts: the 64 of t < 8 are scanned, t = 8 (bin N/2, thread 0) follows them.
template <int WPF>
__device__ __forceinline__ uint32_t store_ordered(const unsigned long long (&masks)[9], const omx_spectrogram_point (&pts)[9], const uint32_t* scan,
                                                  int lane, int wf, bool in_range, omx_spectrogram_point* out) {
    static_assert(WPF <= 8 || WPF == 16, "one wavefront's lanes hold the counts of t < 8");
    const int wf_u = __builtin_amdgcn_readfirstlane(wf);
    uint32_t running = 0;
    if constexpr (WPF <= 8) {
        constexpr int NCNT = WPF < 8 ? 9 * WPF : 64;
        const uint32_t cnt = lane < NCNT ? scan[lane] : 0u;
        const uint32_t inc = wave_inclusive_sum(cnt);
        const uint32_t exc = inc - cnt;
        running = (uint32_t)__builtin_amdgcn_readlane((int)inc, NCNT - 1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            uint32_t before;
            if (WPF == 8 && t == 8) before = running;  // bin N/2 comes last
            else before = (uint32_t)__builtin_amdgcn_readlane((int)exc, t * WPF + wf_u);
            if (in_range && ((masks[t] >> lane) & 1ull))
                *reinterpret_cast<omx_spectrogram_point*>(reinterpret_cast<char*>(out) + (before + lanes_below(masks[t])) * 12u) = pts[t];
        }
        if constexpr (WPF == 8) running += scan[8 * WPF];
    } else {  // 16384 points: 144 counts, summed per thread
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            uint32_t before = running;
            for (int w = 0; w < WPF; ++w) {
                const uint32_t c = scan[t * WPF + w];
                if (w < wf) before += c;
                running += c;
            }
            if (in_range && ((masks[t] >> lane) & 1ull))
                *reinterpret_cast<omx_spectrogram_point*>(reinterpret_cast<char*>(out) + (before + lanes_below(masks[t])) * 12u) = pts[t];
        }
    }
    return running;
}

// BINS: the window is applied on the bins (two-term cosine-sum windows, Hann / Hamming): Z = FFT(s) and Z2 = FFT((n - c) s) as ONE
// dual transform, then FFT(w s)[k] = c0 Z[k] + c1/2 (Z[k-1] + Z[k+1]), FFT(t w s) the same combination of Z2, and
// FFT(w' s)[k] = i c1 (pi / N) (Z[k-1] - Z[k+1]) — four transforms per column instead of five, no window tables
// (derivation and accuracy: stft4096_pair_kernels.hip).
template <int LOGN, bool BINS>
__global__ __launch_bounds__(FftGeom<LOGN>::WG, FftGeom<LOGN>::WG == 256 ? 2 : 1) void stft_reassigned_pow2_kernel(StftFastArgs a) {
    using G = FftGeom<LOGN>;
    constexpr int N = G::N, T = G::T, F = G::FRAMES, WPF = T / 64;  // waves per frame
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* lds = reinterpret_cast<v2f*>(smem_raw);
    v2f* tw2_lds = lds + 2 * F * G::LDS;                              // [256]
    uint32_t* scan_all = reinterpret_cast<uint32_t*>(tw2_lds + 256);  // [F][9][WPF]
    float* hil_all = reinterpret_cast<float*>(scan_all + F * 9 * WPF);  // [F][2]

    const uint32_t chunks = (a.n_cols + F - 1) / F;
    // XCD-aware block -> (stream, chunk) map (see block_to_stream_column in stft_kernels.hip)
    const uint32_t blk = blockIdx.x, xcd = blk & 7u, q = blk >> 3;
    const uint32_t s = (q / chunks) * 8u + xcd, chunk = q % chunks;
    if (s >= a.n_streams) return;
    const int fs = threadIdx.x / T, jf = threadIdx.x % T;
    const unsigned ju = (unsigned)jf;
    const int lane = threadIdx.x & 63, wf = jf >> 6;
    v2f* A = lds + (2 * fs) * G::LDS;
    v2f* B = A + G::LDS;
    uint32_t* scan = scan_all + fs * 9 * WPF;
    float* hil = hil_all + fs * 2;
    const uint32_t col_raw = chunk * F + (uint32_t)fs;
    const uint32_t n_cols_s = stft_cols(a, s);  // ragged banks: this stream's own column count
    if (chunk * F >= n_cols_s) return;           // (whole workgroup: every frame slot is past it)
    const bool in_range = col_raw < n_cols_s;
    const uint32_t col = in_range ? col_raw : n_cols_s - 1u;  // idle frame slots shadow the last column (barriers stay uniform)

    const char* ring_bytes = reinterpret_cast<const char*>(a.ring + (uint64_t)s * a.cap);
    const uint32_t bytemask = (uint32_t)(a.cap - 1) << 2;
    const long long last_nonzero = a.last_nonzero[s];
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};
    const uint64_t p0 = stft_tail(a, s) + (uint64_t)col * a.hop;
    const uint32_t p32 = (uint32_t)p0;
    uint32_t* count_out = a.counts + (uint64_t)s * a.n_cols + col;
    // silent fast path (:307-316).  The first frame slot has the smallest p0: if it is silent, all of them are.
    const uint64_t p0_first = stft_tail(a, s) + (uint64_t)(chunk * F) * a.hop;
    if (last_nonzero < (long long)p0_first) {
        if (jf == 0 && in_range) *count_out = 0;
        return;
    }
    const bool silent = last_nonzero < (long long)p0;  // this slot only: computed anyway, emitted empty

    TwiddlesPow2<LOGN> tw;
    tw.tw2 = tw2_lds;
    tw.load(a.tw4096, ju);  // `tw4096` carries exp(-2 pi i k / N) for this N
    if (threadIdx.x < 256) tw2_lds[threadIdx.x] = a.tw256[threadIdx.x];

    // ---- 1. packed real FFT of the 2N-sample window (loads for the Hilbert build issued alongside) -----------------
    v2f v[16], w2n[16];
    if ((p0 & 1ull) == 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
            v[t] = *reinterpret_cast<const v2f*>(ring_bytes + (((p32 + 2u * (ju + (unsigned)T * (unsigned)t)) << 2) & bytemask));
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t qq = p32 + 2u * (ju + (unsigned)T * (unsigned)t);
            v[t] = v2f{*reinterpret_cast<const float*>(ring_bytes + ((qq << 2) & bytemask)),
                       *reinterpret_cast<const float*>(ring_bytes + (((qq + 1u) << 2) & bytemask))};
        }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) w2n[t] = a.tw8192[ju + (unsigned)T * (unsigned)t];  // exp(-2 pi i k / 2N)
    lds_workgroup_barrier();  // tw2_lds (shared by every frame slot)
    fftp<false, LOGN>(v, A, B, jf, tw);  // v[t] = Zf[jf + T t]; last read: B

    // ---- 2. Hilbert transform with one half-length inverse (see stft_kernels.hip for the derivation) ------------------
    //   Re analytic[n] = N x[n] - X[0]/2 + X[N](-1)^n / 2 ;  Im analytic = inverse REAL FFT of -i X[k]
#pragma unroll
    for (int t = 0; t < 16; ++t) A[pad16(jf + T * t)] = v[t];
    if (jf == 0) {
        hil[0] = (v[0].x + v[0].y) * 0.5f;  // X[0] / 2
        hil[1] = (v[0].x - v[0].y) * 0.5f;  // X[N] / 2   (N = the half-length: the real transform has 2N points)
    }
    frame_sync<LOGN>();
    v2f y[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const unsigned k = ju + (unsigned)T * (unsigned)t;
        const v2f z = v[t];
        const v2f zr = A[pad16((int)(((unsigned)N - k) & (unsigned)(N - 1)))];
        const v2f sum{z.x + zr.x, z.y - zr.y};   // Zf[k] + conj Zf[N-k]  (the 1/2 lives in the twiddle table)
        const v2f dif{z.x - zr.x, z.y + zr.y};   // Zf[k] - conj Zf[N-k]
        y[t] = cmulc(sum, w2n[t]) - cmul(dif, w2n[t]);
        if (k == 0) y[t] = v2f{0.0f, 0.0f};
    }
    const float half_x0 = hil[0], half_xn = hil[1];
    float pw[16], pdw[16], pxr[16];
    {
        const uint32_t qe = p32 + (uint32_t)(N / 2) + ju;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if constexpr (!BINS) {
                pw[t] = a.window[ju + (unsigned)T * (unsigned)t];
                pdw[t] = a.dwindow[ju + (unsigned)T * (unsigned)t];
            } else {
                pw[t] = pdw[t] = 0.0f;
            }
            pxr[t] = *reinterpret_cast<const float*>(ring_bytes + (((qe + (unsigned)T * (unsigned)t) << 2) & bytemask));
        }
    }
    // B was last read before the barrier above; A is released by pass 1's barrier (pass 2 writes it)
    fftp<true, LOGN>(y, B, A, jf, tw);  // y[t] = (Im a[2m], Im a[2m+1]), m = jf + T t; last read: A

    // ---- 3. analytic slice s[i] = analytic[N/2 + i], i = jf + T t ---------------------------------------------------------
    float* imag = reinterpret_cast<float*>(B);  // N floats: Im analytic[N/2 .. 3N/2)
#pragma unroll
    for (int t = 4; t < 12; ++t) *reinterpret_cast<v2f*>(imag + 2 * (jf + T * t - N / 4)) = y[t];
    frame_sync<LOGN>();
    const float parity = (jf & 1) ? -half_xn : half_xn;  // n = N/2 + i has the parity of jf (N/2 and T are even)
    v2f vb[16], vd[16], vt[16];
    constexpr float CENTER = (float)(N - 1) * 0.5f;
    v2f bb[9], bd[9];
    float pn[9];
    if constexpr (!BINS) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const v2f sv{(float)N * pxr[t] - half_x0 + parity, imag[jf + T * t]};
            const float w = pw[t], dw = pdw[t];
            const float wt = ((float)(jf + T * t) - CENTER) * w;  // compute_time_weighted (:601-608)
            vb[t] = v2f{sv.x * w, sv.y * w};
            vd[t] = v2f{sv.x * dw, sv.y * dw};
            vt[t] = v2f{sv.x * wt, sv.y * wt};
        }
        frame_sync<LOGN>();  // imag[] (in B) is consumed
        fftp_dual<false, LOGN>(vb, vd, A, B, jf, tw);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            bb[t] = vb[t];
            bd[t] = vd[t];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) pn[t] = a.bin_norm[(t < 8 || jf == 0) ? ju + (unsigned)T * (unsigned)t : 0u];
        frame_sync<LOGN>();  // the paired transform's last pass still reads A and B
        fftp<false, LOGN>(vt, A, B, jf, tw);
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const v2f sv{(float)N * pxr[t] - half_x0 + parity, imag[jf + T * t]};
            const float nc = (float)(jf + T * t) - CENTER;  // compute_time_weighted's ramp (:601-608)
            vb[t] = sv;                                     // Z  = FFT(s)
            vd[t] = v2f{sv.x * nc, sv.y * nc};              // Z2 = FFT((n - c) s)
        }
        frame_sync<LOGN>();  // imag[] (in B) is consumed
        fftp_dual<false, LOGN>(vb, vd, A, B, jf, tw);
#pragma unroll
        for (int t = 0; t < 9; ++t) pn[t] = a.bin_norm[(t < 8 || jf == 0) ? ju + (unsigned)T * (unsigned)t : 0u];
        frame_sync<LOGN>();  // the transform's last pass still reads A and B
        // natural-order copy of bins -1 ... N/2 + T of both spectra (slot 1 + k = bin k, slot 0 = bin -1 = bin N - 1)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            A[1 + jf + T * t] = vb[t];
            B[1 + jf + T * t] = vd[t];
        }
        if (jf == T - 1) {
            A[0] = vb[15];
            B[0] = vd[15];
        }
        frame_sync<LOGN>();
        const float c0 = a.win_c0, half_c1 = 0.5f * a.win_c1, dscale = a.win_c1 * (3.14159265358979323846f / (float)N);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint32_t bin = ju + (unsigned)T * (unsigned)t;
            const bool mine = t < 8 || jf == 0;
            const uint32_t at = mine ? bin : 0u;
            const v2f zm = A[at], zp = A[at + 2], z2m = B[at], z2p = B[at + 2];
            const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y}, z2s{z2m.x + z2p.x, z2m.y + z2p.y};
            bb[t] = v2f{c0 * vb[t].x + half_c1 * zs.x, c0 * vb[t].y + half_c1 * zs.y};
            bd[t] = v2f{-dscale * zd.y, dscale * zd.x};  // i c1 (pi / N) (Z[k-1] - Z[k+1])
            vt[t] = v2f{c0 * vd[t].x + half_c1 * z2s.x, c0 * vd[t].y + half_c1 * z2s.y};
        }
    }

    // ---- 4. reassignment + ordered compaction (bins jf + T t, t < 8, and bin N/2 on thread 0) ----------------------------
    omx_spectrogram_point pts[9];
    unsigned long long masks[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint32_t bin = ju + (unsigned)T * (unsigned)t;
        const bool keep = reassign_flat(bin, bb[t], bd[t], vt[t], pn[t], rc, pts[t]) && (t < 8 || jf == 0) && !silent;
        masks[t] = __ballot(keep);
        if (lane == 0) scan[t * WPF + wf] = (uint32_t)__popcll(masks[t]);
    }
    frame_sync<LOGN>();
    omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
    const uint32_t running = store_ordered<WPF>(masks, pts, scan, lane, wf, in_range, out);
    if (jf == 0 && in_range) *count_out = running;
}

// ================================================================================================
// K2p-pair: the pair form of the tuned 4096 kernel (stft4096_pair_kernels.hip) for W = F = 1024 / 2048, Hann / Hamming: a frame slot
// (T = N/16 threads) carries TWO consecutive columns, so the packed-real forward transform and the Hilbert inverse run as dual
// transforms too (two dependency chains per wavefront between shared syncs) — two dual-transform phases per column where the
// one-column-per-slot kernel above spends three transform phases.  With T = 64 (1024 points) a slot is one wavefront and no phase
// of it needs a workgroup barrier.  2048 / hop 64 is the reference's default spectrogram shape.
// ================================================================================================
template <int TT>
__device__ __forceinline__ void half_twiddle_run(v2f (&w8)[16], v2f base) {  // base * exp(-2 pi i t / 32): element jf + T t of exp(-2 pi i k / 2N) / 2
    if constexpr (TT < 16) {
        w8[TT] = rotate128<4 * TT>(base);
        half_twiddle_run<TT + 1>(w8, base);
    }
}

// (superseded by the tri kernels below since round 4: compiled into the tuning library and into tools/build_ab.sh's POW2_FORCE_PAIR builds only)
#if defined(OMX_TUNING) || defined(POW2_FORCE_PAIR)
template <int LOGN>
__global__ __launch_bounds__(256, 2) void stft_reassigned_pow2_pair_kernel(StftFastArgs a) {
    using G = FftGeom<LOGN>;
    constexpr int N = G::N, T = G::T, F = G::FRAMES, WPF = T / 64;  // F pair slots per workgroup
    static_assert(LOGN == 10 || LOGN == 11, "1024 / 2048 points");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* lds = reinterpret_cast<v2f*>(smem_raw);
    v2f* tw2_lds = lds + 2 * F * G::LDS;                              // [256]
    uint32_t* scan_all = reinterpret_cast<uint32_t*>(tw2_lds + 256);  // [F][9][WPF]
    float* hil_all = reinterpret_cast<float*>(scan_all + F * 9 * WPF);  // [F][4]

    const uint32_t pairs = (a.n_cols + 1u) / 2u, chunks = (pairs + F - 1) / F;
    const uint32_t blk = blockIdx.x, xcd = blk & 7u, q = blk >> 3;
    const uint32_t s = (q / chunks) * 8u + xcd, chunk = q % chunks;
    if (s >= a.n_streams) return;
    const int fs = threadIdx.x / T, jf = threadIdx.x % T;
    const unsigned ju = (unsigned)jf;
    const int lane = threadIdx.x & 63, wf = jf >> 6;
    v2f* A = lds + (2 * fs) * G::LDS;
    v2f* B = A + G::LDS;
    uint32_t* scan = scan_all + fs * 9 * WPF;
    float* hil = hil_all + fs * 4;
    const uint32_t n_cols_s = stft_cols(a, s), pairs_s = (n_cols_s + 1u) / 2u;
    if (chunk * F >= pairs_s) return;  // (whole workgroup: every slot is past this stream's columns)
    const uint32_t pair_raw = chunk * F + (uint32_t)fs;
    const bool in_range = pair_raw < pairs_s;
    const uint32_t pair = in_range ? pair_raw : pairs_s - 1u;  // idle slots shadow the last pair (syncs stay uniform)
    const uint32_t col0 = 2u * pair;
    const bool have1 = col0 + 1u < n_cols_s;
    const uint32_t col1 = have1 ? col0 + 1u : col0;  // an odd tail computes column 0 twice and stores it once

    const float* ring = a.ring + (uint64_t)s * a.cap;
    const uint32_t mask32 = (uint32_t)(a.cap - 1);
    const long long last_nonzero = a.last_nonzero[s];
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};
    const uint64_t tail_s = stft_tail(a, s);
    const uint64_t p0a = tail_s + (uint64_t)col0 * a.hop, p0b = tail_s + (uint64_t)col1 * a.hop;
    // silent fast path (:307-316).  The first slot's first column has the smallest p0: if it is silent, every column here is.
    const uint64_t p0_first = tail_s + (uint64_t)(2u * chunk * F) * a.hop;
    if (last_nonzero < (long long)p0_first) {
        if (jf == 0 && in_range) {
            a.counts[(uint64_t)s * a.n_cols + col0] = 0;
            if (have1) a.counts[(uint64_t)s * a.n_cols + col1] = 0;
        }
        return;
    }
    const bool silent_a = last_nonzero < (long long)p0a, silent_b = last_nonzero < (long long)p0b;  // computed anyway, emitted empty

    TwiddlesPow2<LOGN> tw;
    tw.tw2 = tw2_lds;
    tw.load(a.tw4096, ju);  // `tw4096` carries exp(-2 pi i k / N) for this N
    if (threadIdx.x < 256) tw2_lds[threadIdx.x] = a.tw256[threadIdx.x];
    const v2f w8_base = a.tw8192[ju];  // exp(-2 pi i jf / 2N) / 2
    float pn[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) pn[t] = a.bin_norm[(t < 8 || jf == 0) ? ju + (unsigned)T * (unsigned)t : 0u];

    // ---- 1. packed real FFTs of the two 2N-sample windows ------------------------------------------------------------------------------
    const uint32_t pa32 = (uint32_t)p0a, pb32 = (uint32_t)p0b;
    v2f va[16], vb[16];
    if (((p0a | p0b) & 1ull) == 0) {  // pairs are 8-byte aligned and never straddle the ring wrap
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            va[t] = *reinterpret_cast<const v2f*>(ring + ((pa32 + 2u * (ju + (unsigned)T * (unsigned)t)) & mask32));
            vb[t] = *reinterpret_cast<const v2f*>(ring + ((pb32 + 2u * (ju + (unsigned)T * (unsigned)t)) & mask32));
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t qa = pa32 + 2u * (ju + (unsigned)T * (unsigned)t), qb = pb32 + 2u * (ju + (unsigned)T * (unsigned)t);
            va[t] = v2f{ring[qa & mask32], ring[(qa + 1u) & mask32]};
            vb[t] = v2f{ring[qb & mask32], ring[(qb + 1u) & mask32]};
        }
    }
    lds_workgroup_barrier();  // tw2_lds (shared by every slot)
    fftp_dual<false, LOGN>(va, vb, A, B, jf, tw);  // v[t] = Zf[jf + T t]

    // ---- 2. Hilbert transform with one half-length inverse per column (derivation: stft_kernels.hip step 2) ---------------------------
    frame_sync<LOGN>();  // pass 3 of the dual transform still reads A and B
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        A[pad16(jf + T * t)] = va[t];
        B[pad16(jf + T * t)] = vb[t];
    }
    if (jf == 0) {
        hil[0] = (va[0].x + va[0].y) * 0.5f;  // X[0] / 2
        hil[1] = (va[0].x - va[0].y) * 0.5f;  // X[N] / 2
        hil[2] = (vb[0].x + vb[0].y) * 0.5f;
        hil[3] = (vb[0].x - vb[0].y) * 0.5f;
    }
    frame_sync<LOGN>();
    v2f ya[16], yb[16];
    {
        v2f w8[16];
        half_twiddle_run<0>(w8, w8_base);
        // partner Zf[(N - k) & (N - 1)] of k = jf + T t sits at pad16(N - jf) - (T + T/16) t (thread 0: pad16(N) - ...; its t = 0 read
        // lands one slot past the buffer, inside the allocation, and is not used)
        constexpr int PS = T + T / 16;
        const int part = (jf ? pad16(N - jf) : N + N / 16) - PS * 15;
        auto hilbert_spectrum = [&](v2f (&y)[16], const v2f (&v)[16], const v2f* X) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const v2f z = v[t], zr = X[part + PS * (15 - t)];
                const v2f sum{z.x + zr.x, z.y - zr.y}, dif{z.x - zr.x, z.y + zr.y};
                y[t] = cmulc(sum, w8[t]) - cmul(dif, w8[t]);
                if (t == 0 && jf == 0) y[t] = v2f{0.0f, 0.0f};
            }
        };
        hilbert_spectrum(ya, va, A);
        hilbert_spectrum(yb, vb, B);
    }
    const float half_x0a = hil[0], half_xna = hil[1], half_x0b = hil[2], half_xnb = hil[3];
    float xra[16], xrb[16];  // the real part's samples, in flight during the inverse
    {
        const uint32_t qa = pa32 + (uint32_t)(N / 2) + ju, qb = pb32 + (uint32_t)(N / 2) + ju;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            xra[t] = ring[(qa + (unsigned)T * (unsigned)t) & mask32];
            xrb[t] = ring[(qb + (unsigned)T * (unsigned)t) & mask32];
        }
    }
    frame_sync<LOGN>();  // partners are read from the buffers the inverse is about to overwrite
    fftp_dual<true, LOGN>(ya, yb, A, B, jf, tw);  // y[t] = (Im a[2m], Im a[2m+1]), m = jf + T t

    // ---- 3. analytic slices s[i] = analytic[N/2 + i], i = jf + T t -------------------------------------------------------------------------
    frame_sync<LOGN>();
    float* imag_a = reinterpret_cast<float*>(A);
    float* imag_b = reinterpret_cast<float*>(B);
#pragma unroll
    for (int t = 4; t < 12; ++t) {
        *reinterpret_cast<v2f*>(imag_a + 2 * (jf + T * t - N / 4)) = ya[t];
        *reinterpret_cast<v2f*>(imag_b + 2 * (jf + T * t - N / 4)) = yb[t];
    }
    frame_sync<LOGN>();
    v2f sa[16], sb[16];
    {
        const float par_a = (jf & 1) ? -half_xna : half_xna, par_b = (jf & 1) ? -half_xnb : half_xnb;  // n = N/2 + i has jf's parity
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            sa[t] = v2f{(float)N * xra[t] - half_x0a + par_a, imag_a[jf + T * t]};
            sb[t] = v2f{(float)N * xrb[t] - half_x0b + par_b, imag_b[jf + T * t]};
        }
    }
    const float c0 = a.win_c0, half_c1 = 0.5f * a.win_c1, dscale = a.win_c1 * (3.14159265358979323846f / (float)N);
    constexpr float CENTER = (float)(N - 1) * 0.5f;

    // ---- 4. per column: Z = FFT(s), Z2 = FFT((n - c) s) as one dual transform; windows applied on the bins -------------------------------
    auto column = [&](const v2f (&sv)[16], bool silent, bool store, uint32_t col) {
        v2f z[16], z2[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float nc = (float)(jf + T * t) - CENTER;  // compute_time_weighted's ramp (:601-608)
            z[t] = sv[t];
            z2[t] = v2f{sv[t].x * nc, sv[t].y * nc};
        }
        frame_sync<LOGN>();  // the gather above / the previous column's neighbour reads still use A and B
        fftp_dual<false, LOGN>(z, z2, A, B, jf, tw);
        frame_sync<LOGN>();  // pass 3 still reads A and B
#pragma unroll
        for (int t = 0; t < 9; ++t) {  // natural-order copy of bins -1 ... N/2 + T (slot 1 + k = bin k, slot 0 = bin -1 = bin N - 1)
            A[1 + jf + T * t] = z[t];
            B[1 + jf + T * t] = z2[t];
        }
        if (jf == T - 1) {
            A[0] = z[15];
            B[0] = z2[15];
        }
        frame_sync<LOGN>();
        omx_spectrogram_point pts[9];
        unsigned long long masks[9];
#pragma unroll
        for (int h = 0; h < 9; h += 3) {  // three bins at a time: the neighbour reads of a group are issued together
            v2f nzm[3], nzp[3], nz2m[3], nz2p[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int bin = jf + T * (h + u);  // (t = 8: only thread 0's bin exists; the others read slots inside the buffer)
                nzm[u] = A[bin];
                nzp[u] = A[bin + 2];
                nz2m[u] = B[bin];
                nz2p[u] = B[bin + 2];
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int t = h + u;
                const uint32_t bin = ju + (unsigned)T * (unsigned)t;
                const v2f zm = nzm[u], zp = nzp[u], z2m = nz2m[u], z2p = nz2p[u];
                const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y}, z2s{z2m.x + z2p.x, z2m.y + z2p.y};
                const v2f bb{c0 * z[t].x + half_c1 * zs.x, c0 * z[t].y + half_c1 * zs.y};
                const v2f bd{-dscale * zd.y, dscale * zd.x};  // i c1 (pi / N) (Z[k-1] - Z[k+1])
                const v2f bt{c0 * z2[t].x + half_c1 * z2s.x, c0 * z2[t].y + half_c1 * z2s.y};
                const bool keep = reassign_flat(bin, bb, bd, bt, pn[t], rc, pts[t]) && (t < 8 || jf == 0) && !silent;
                masks[t] = __ballot(keep);
                if (lane == 0) scan[t * WPF + wf] = (uint32_t)__popcll(masks[t]);
            }
        }
        frame_sync<LOGN>();
        omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
        const uint32_t running = store_ordered<WPF>(masks, pts, scan, lane, wf, store, out);
        if (jf == 0 && store) a.counts[(uint64_t)s * a.n_cols + col] = running;
    };
    column(sa, silent_a, in_range, col0);
    column(sb, silent_b, in_range && have1, col1);
}

template <int LOGN>
static void launch_pow2_pair(const StftFastArgs& a, hipStream_t stream) {
    using G = FftGeom<LOGN>;
    constexpr int F = G::FRAMES, WPF = G::T / 64;
    const size_t lds = (size_t)(2 * F * G::LDS + 256) * sizeof(v2f) + (size_t)F * 9 * WPF * sizeof(uint32_t) + (size_t)F * 4 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_pow2_pair_kernel<LOGN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    const uint32_t pairs = (a.n_cols + 1u) / 2u, chunks = (pairs + F - 1) / F;
    hipLaunchKernelGGL((stft_reassigned_pow2_pair_kernel<LOGN>), dim3(stream_column_grid(a.n_streams, chunks)), dim3(G::WG), lds, stream, a);
}
#endif  // OMX_TUNING || POW2_FORCE_PAIR

// ================================================================================================
// K1p: fused classic column (reference spectrogram/processor.rs:350-380, window.rs:66-88) for W = F = 1024 / 2048 / 4096.
// Two consecutive columns share one complex FFT (column 2p in the real part, 2p+1 in the imaginary part); a frame slot is
// T = N/16 threads, so a workgroup emits 2 * 256/T columns.  One LDS buffer per slot (in-place transform) -> 4 workgroups
// per CU.  HBM-bound by design: hop*C*4 B in, (N/2+1)*2 B out per column.
// ================================================================================================
// Hardware log2 (v_log_f32, 1 ulp) instead of libm's denormal-safe logf: zero / denormal / NaN power all land on the -140 dB
// clamp exactly as `power > 0 ? max(ln p * LN_TO_DB, floor) : floor` does, and since db >= -140 the rounded value is >= 1680,
// where floor(x + 0.5) is exactly round-half-away (x + 0.5 is representable): 6 VALU per bin instead of ~30.
__device__ __forceinline__ uint16_t classic_code(float power) {  // level.rs:28-34 + processor.rs:103-108
    const float db = fmaxf(__builtin_amdgcn_logf(power) * 3.0102999566f, -140.0f);
    const float v = fminf(floorf((db + 144.0f) * (65535.0f / 156.0f) + 0.5f), 65535.0f);
    return (uint16_t)v;
}

template <int LOGN>
__global__ __launch_bounds__(FftGeom<LOGN>::WG) void stft_classic_pow2_kernel(StftFastArgs a, uint16_t* __restrict__ codes) {
    using G = FftGeom<LOGN>;
    constexpr int N = G::N, T = G::T, F = G::FRAMES, WPF = T / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* lds = reinterpret_cast<v2f*>(smem_raw);                      // [F][G::LDS]
    v2f* tw2_lds = lds + F * G::LDS;                                  // [256]
    float (*wave_sum)[2][WPF] = reinterpret_cast<float (*)[2][WPF]>(tw2_lds + 256);  // [F][2][WPF]
    float (*wave_max)[2][WPF] = wave_sum + F;                                        // [F][2][WPF]
    const uint32_t pairs = (a.n_cols + 1) / 2, chunks = (pairs + F - 1) / F;
    const uint32_t blk = blockIdx.x, xcd = blk & 7u, q = blk >> 3;
    const uint32_t s = (q / chunks) * 8u + xcd, chunk = q % chunks;
    if (s >= a.n_streams) return;
    const int fs = threadIdx.x / T, jf = threadIdx.x % T;
    const unsigned ju = (unsigned)jf;
    const int wf = jf >> 6;
    v2f* buf = lds + fs * G::LDS;
    const uint32_t n_cols_s = stft_cols(a, s), pairs_s = (n_cols_s + 1u) / 2u;  // ragged banks: this stream's own column count
    if (chunk * F >= pairs_s) return;
    const uint32_t pair_raw = chunk * F + (uint32_t)fs;
    const bool in_range = pair_raw < pairs_s;
    const uint32_t pair = in_range ? pair_raw : pairs_s - 1u;
    const uint32_t col_a = 2u * pair;
    const bool has_b = col_a + 1u < n_cols_s;

    const char* ring_bytes = reinterpret_cast<const char*>(a.ring + (uint64_t)s * a.cap);
    const uint32_t bytemask = (uint32_t)(a.cap - 1) << 2;
    const uint32_t p32 = (uint32_t)(stft_tail(a, s) + (uint64_t)col_a * a.hop);
    // zero padding (window W < transform N, `processor.rs:350-368`): element i = jf + T t of the frame is sample i of the window
    // for i < W and 0 beyond; loads use a clamped index and the selects sit where the values are consumed
    const uint32_t Wn = a.window_size ? a.window_size : (uint32_t)N;
    float xa[16], xb[16], w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t i = ju + (unsigned)T * (unsigned)t;
        const uint32_t qq = p32 + (i < Wn ? i : 0u);
        xa[t] = *reinterpret_cast<const float*>(ring_bytes + ((qq << 2) & bytemask));
        xb[t] = *reinterpret_cast<const float*>(ring_bytes + (((qq + (has_b ? a.hop : 0u)) << 2) & bytemask));
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t i = ju + (unsigned)T * (unsigned)t;
        w[t] = a.window[i < Wn ? i : 0u];
    }
    TwiddlesPow2<LOGN> tw;
    tw.tw2 = tw2_lds;
    tw.load(a.tw4096, ju);
    if (threadIdx.x < 256) tw2_lds[threadIdx.x] = a.tw256[threadIdx.x];
    float norm[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) norm[t] = a.bin_norm[(t < 8 || jf == 0) ? ju + (unsigned)T * (unsigned)t : 0u];
    // window.rs:76-79: the column's mean is the reference's sequential f32 sum over the window / W — taken by window_sums_seq_kernel ahead of
    // this launch (a.col_sums; round 6: a tree sum here before, 8e-5 of the column maximum from the reference on a hop with a large offset)
    const float* cs = a.col_sums + (uint64_t)s * a.n_cols + col_a;
    const float ta = cs[0], tb = has_b ? cs[1] : 0.0f;
    lds_workgroup_barrier();  // tw2_lds (shared by every frame slot)
    const float mean_a = ta / (float)Wn, mean_b = tb / (float)Wn;  // mean over the window (window.rs:80-84)
    v2f v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const bool inside = ju + (unsigned)T * (unsigned)t < Wn;
        v[t] = v2f{inside ? (xa[t] - mean_a) * w[t] : 0.0f, (inside && has_b) ? (xb[t] - mean_b) * w[t] : 0.0f};
    }
    // Level equalisation (round 5).  The two columns ride one complex transform, and the split X_a = (Z[k] + conj Z[N-k]) / 2 cancels
    // column b's spectrum only as far as the computed transform of b is Hermitian — to ~4e-7 of b's LARGEST bin.  A column far below
    // its partner (an onset, a release: soak seeds 21028002 / 21093005 / 21109003, 97 / 64 / 42 dB apart) therefore came out 211 / 6 / 2
    // codes off where the reference, one real transform per column, has no such coupling.  Each column is scaled by the power of two
    // that brings its largest windowed sample to [1, 2) — exact in f32 — and its powers are scaled back by the exact inverse: the
    // coupling is then relative to comparable levels whatever the columns' own levels were.
    float pa = 0.0f, pb = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        pa = fmaxf(pa, fabsf(v[t].x));
        pb = fmaxf(pb, fabsf(v[t].y));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        pa = fmaxf(pa, __shfl_xor(pa, off));
        pb = fmaxf(pb, __shfl_xor(pb, off));
    }
    if ((jf & 63) == 0) {
        wave_max[fs][0][wf] = pa;
        wave_max[fs][1][wf] = pb;
    }
    frame_sync<LOGN>();
#pragma unroll
    for (int i = 0; i < WPF; ++i) {
        pa = fmaxf(pa, wave_max[fs][0][i]);
        pb = fmaxf(pb, wave_max[fs][1][i]);
    }
    // frexp exponent e: p = m 2^e with m in [0.5, 1); a silent or non-finite column keeps its scale
    const int ea = (pa > 0.0f && pa < INFINITY) ? __builtin_amdgcn_frexp_expf(pa) : 0, eb = (pb > 0.0f && pb < INFINITY) ? __builtin_amdgcn_frexp_expf(pb) : 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = v2f{ldexpf(v[t].x, -ea), ldexpf(v[t].y, -eb)};
    fftp_inplace<false, LOGN>(v, buf, jf, tw);
    frame_sync<LOGN>();
#pragma unroll
    for (int t = 0; t < 16; ++t) buf[pad16(jf + T * t)] = v[t];
    frame_sync<LOGN>();
    if (!in_range) return;
    uint16_t* out_a = codes + ((uint64_t)s * a.n_cols + col_a) * a.column_stride;
    uint16_t* out_b = out_a + a.column_stride;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t == 8 && jf != 0) break;
        const uint32_t k = ju + (unsigned)T * (unsigned)t;
        const v2f z = v[t];
        const v2f zr = buf[pad16((int)(((unsigned)N - k) & (unsigned)(N - 1)))];
        const v2f xa_k{(z.x + zr.x) * 0.5f, (z.y - zr.y) * 0.5f};  // (Z + conj Zr) / 2
        const v2f xb_k{(z.y + zr.y) * 0.5f, (zr.x - z.x) * 0.5f};  // (Z - conj Zr) / (2i)
        out_a[k] = classic_code(ldexpf((xa_k.x * xa_k.x + xa_k.y * xa_k.y) * norm[t], 2 * ea));
        if (has_b) out_b[k] = classic_code(ldexpf((xb_k.x * xb_k.x + xb_k.y * xb_k.y) * norm[t], 2 * eb));
    }
}

template <int LOGN>
static void launch_classic(const StftFastArgs& a, uint16_t* codes, hipStream_t stream) {
    using G = FftGeom<LOGN>;
    constexpr int F = G::FRAMES, WPF = G::T / 64;
    const size_t lds = (size_t)(F * G::LDS + 256) * sizeof(v2f) + (size_t)F * 4 * WPF * sizeof(float);  // + wave sums, wave maxima
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_classic_pow2_kernel<LOGN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const uint32_t pairs = (a.n_cols + 1) / 2, chunks = (pairs + F - 1) / F;
    hipLaunchKernelGGL(stft_classic_pow2_kernel<LOGN>, dim3(stream_column_grid(a.n_streams, chunks)), dim3(G::WG), lds, stream, a, codes);
}
// window.rs:76-79: every classic column's sequential f32 window sum, ahead of the transform kernel that divides it by the window length
static void launch_classic_window_sums(const StftFastArgs& a, uint32_t window, hipStream_t stream) {
    WindowSumArgs w{};
    w.ring[0] = a.ring;
    w.n_rings = 1;
    w.cap = a.cap;
    w.tail = a.tail;
    w.tails = a.tails;
    w.hops = a.cols;
    w.hop = a.hop;
    w.window = window;
    w.first_hop = 0;
    w.n_hops = a.n_cols;
    w.n_streams = a.n_streams;
    w.sums = a.col_sums;
    launch_window_sums(w, stream);
}
void launch_stft_classic_pow2(const StftFastArgs& a, uint16_t* codes, uint32_t fft_size, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return;
    launch_classic_window_sums(a, a.window_size ? a.window_size : fft_size, stream);
    switch (fft_size) {
        case 1024: launch_classic<10>(a, codes, stream); break;
        case 2048: launch_classic<11>(a, codes, stream); break;
        case 4096: launch_classic<12>(a, codes, stream); break;
        case 8192: launch_classic<13>(a, codes, stream); break;
        case 16384: launch_classic<14>(a, codes, stream); break;
        default: break;
    }
}

// ================================================================================================
// K2z: the same fused reassigned STFT with zero padding (the GUI offers 2 ... 32x, `ui/settings/spectrogram.rs:13`): window
// length W = 2^LOGW, transform length F = 2^LOGF = zp W.  The Hilbert pair works on 2W samples (a W-point complex packing,
// `processor.rs:546-557` with H = next_pow2(2W)) and is carried by the first W/16 threads of the frame; the analytic slice
// (W values) is then spread over all F/16 threads, zero-extended to F (`apply_complex_window`, :559-567) and the three
// windowed transforms, the reassignment and the compaction run exactly as in the unpadded kernel, over F/2 + 1 bins.
// ================================================================================================
// BINS: Hann / Hamming, the window applied on the bins (cos(2 pi n / W) shifts an F-point spectrum by F / W bins): two windowed
// transforms (Z, Z2 as one dual) instead of three, no window tables
template <int LOGW, int LOGF, bool BINS>
__global__ __launch_bounds__(FftGeom<LOGF>::WG, FftGeom<LOGF>::WG == 256 ? 2 : 1) void stft_reassigned_zp_kernel(
    StftFastArgs a, const v2f* __restrict__ twF) {
    using GW = FftGeom<LOGW>;
    using G = FftGeom<LOGF>;
    static_assert(LOGF > LOGW && LOGF <= 13 && GW::PASSES == 3, "W <= 4096 < F <= 8192, or W < F <= 4096");
    constexpr int W = GW::N, TW = GW::T, N = G::N, T = G::T, F = G::FRAMES, WPF = T / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* lds = reinterpret_cast<v2f*>(smem_raw);
    v2f* tw2_lds = lds + 2 * F * G::LDS;                              // [256]
    uint32_t* scan_all = reinterpret_cast<uint32_t*>(tw2_lds + 256);  // [F][9][WPF]
    float* hil_all = reinterpret_cast<float*>(scan_all + F * 9 * WPF);  // [F][2]

    const uint32_t chunks = (a.n_cols + F - 1) / F;
    const uint32_t blk = blockIdx.x, xcd = blk & 7u, q = blk >> 3;
    const uint32_t s = (q / chunks) * 8u + xcd, chunk = q % chunks;
    if (s >= a.n_streams) return;
    const int fs = threadIdx.x / T, jf = threadIdx.x % T;
    const unsigned ju = (unsigned)jf;
    const int lane = threadIdx.x & 63, wf = jf >> 6;
    const bool hact = jf < TW;  // the threads that carry the W-point Hilbert transforms
    v2f* A = lds + (2 * fs) * G::LDS;
    v2f* B = A + G::LDS;
    uint32_t* scan = scan_all + fs * 9 * WPF;
    float* hil = hil_all + fs * 2;
    const uint32_t col_raw = chunk * F + (uint32_t)fs;
    const uint32_t n_cols_s = stft_cols(a, s);
    if (chunk * F >= n_cols_s) return;
    const bool in_range = col_raw < n_cols_s;
    const uint32_t col = in_range ? col_raw : n_cols_s - 1u;

    const char* ring_bytes = reinterpret_cast<const char*>(a.ring + (uint64_t)s * a.cap);
    const uint32_t bytemask = (uint32_t)(a.cap - 1) << 2;
    const long long last_nonzero = a.last_nonzero[s];
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};
    const uint64_t p0 = stft_tail(a, s) + (uint64_t)col * a.hop;
    const uint32_t p32 = (uint32_t)p0;
    uint32_t* count_out = a.counts + (uint64_t)s * a.n_cols + col;
    const uint64_t p0_first = stft_tail(a, s) + (uint64_t)(chunk * F) * a.hop;
    if (last_nonzero < (long long)p0_first) {  // silent fast path (:307-316): the first slot has the smallest p0
        if (jf == 0 && in_range) *count_out = 0;
        return;
    }
    const bool silent = last_nonzero < (long long)p0;

    TwiddlesPow2<LOGW> twh;  // Hilbert transforms (size W): `a.tw4096` = exp(-2 pi i k / W), `a.tw8192` = exp(-2 pi i k / 2W)
    twh.tw2 = tw2_lds;
    twh.load(a.tw4096, hact ? ju : 0u);
    if (threadIdx.x < 256) tw2_lds[threadIdx.x] = a.tw256[threadIdx.x];

    // ---- 1. packed real FFT of the 2W-sample window (W/16 threads) -----------------------------------------------------------
    const unsigned jh = hact ? ju : 0u;  // idle threads shadow thread 0's addresses (loads stay unconditional and in range)
    v2f v[16], w2n[16];
    if ((p0 & 1ull) == 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
            v[t] = *reinterpret_cast<const v2f*>(ring_bytes + (((p32 + 2u * (jh + (unsigned)TW * (unsigned)t)) << 2) & bytemask));
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t qq = p32 + 2u * (jh + (unsigned)TW * (unsigned)t);
            v[t] = v2f{*reinterpret_cast<const float*>(ring_bytes + ((qq << 2) & bytemask)),
                       *reinterpret_cast<const float*>(ring_bytes + (((qq + 1u) << 2) & bytemask))};
        }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) w2n[t] = a.tw8192[jh + (unsigned)TW * (unsigned)t];
    lds_workgroup_barrier();  // tw2_lds (shared by every frame slot)
    fftp_masked<false, LOGW, LOGF>(hact, v, A, B, jf, twh);

    // ---- 2. Hilbert transform with one half-length inverse ---------------------------------------------------------------------
    if (hact) {
#pragma unroll
        for (int t = 0; t < 16; ++t) A[pad16(jf + TW * t)] = v[t];
        if (jf == 0) {
            hil[0] = (v[0].x + v[0].y) * 0.5f;  // X[0] / 2
            hil[1] = (v[0].x - v[0].y) * 0.5f;  // X[W] / 2
        }
    }
    frame_sync<LOGF>();
    v2f y[16];
    if (hact) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned k = ju + (unsigned)TW * (unsigned)t;
            const v2f z = v[t];
            const v2f zr = A[pad16((int)(((unsigned)W - k) & (unsigned)(W - 1)))];
            const v2f sum{z.x + zr.x, z.y - zr.y};   // Zf[k] + conj Zf[N-k]  (the 1/2 lives in the twiddle table)
            const v2f dif{z.x - zr.x, z.y + zr.y};   // Zf[k] - conj Zf[N-k]
            y[t] = cmulc(sum, w2n[t]) - cmul(dif, w2n[t]);
            if (k == 0) y[t] = v2f{0.0f, 0.0f};
        }
    }
    const float half_x0 = hil[0], half_xn = hil[1];
    // window tables and the real part's samples for THIS thread's slice elements i = jf + T u (i < W), in flight during the inverse
    float pw[16], pdw[16], pxr[16];
    {
        const uint32_t qe = p32 + (uint32_t)(W / 2);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned i = ju + (unsigned)T * (unsigned)u;
            const unsigned ic = i < (unsigned)W ? i : 0u;  // clamped: the value is discarded below when i >= W
            if constexpr (!BINS) {
                pw[u] = a.window[ic];
                pdw[u] = a.dwindow[ic];
            } else {
                pw[u] = pdw[u] = 0.0f;
            }
            pxr[u] = *reinterpret_cast<const float*>(ring_bytes + (((qe + ic) << 2) & bytemask));
        }
    }
    fftp_masked<true, LOGW, LOGF>(hact, y, B, A, jf, twh);  // y[t] = (Im a[2m], Im a[2m+1]), m = jf + TW t

    // ---- 3. analytic slice s[i] = analytic[W/2 + i], i < W, spread over the F/16 threads of the frame ---------------------
    float* imag = reinterpret_cast<float*>(B);  // W floats
    if (hact) {
#pragma unroll
        for (int t = 4; t < 12; ++t) *reinterpret_cast<v2f*>(imag + 2 * (jf + TW * t - W / 4)) = y[t];
    }
    TwiddlesPow2<LOGF> tw;  // windowed transforms (size F)
    tw.tw2 = tw2_lds;
    tw.load(twF, ju);
    frame_sync<LOGF>();
    const float parity = (jf & 1) ? -half_xn : half_xn;  // n = W/2 + i has the parity of jf (W/2 and T are even)
    v2f vb[16], vd[16], vt[16];
    constexpr float CENTER = (float)(W - 1) * 0.5f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int i = jf + T * u;
        v2f sv{0.0f, 0.0f};
        float w = 0.0f, dw = 0.0f, wt = 0.0f;
        if (i < W) {  // compile-time for most u: i < W  <=>  u < 16 / zp (T u is a multiple of T, jf < T)
            sv = v2f{(float)W * pxr[u] - half_x0 + parity, imag[i]};
            if constexpr (BINS) {
                w = 1.0f;                    // Z  = FFT(s)
                dw = (float)i - CENTER;      // Z2 = FFT((n - c) s): compute_time_weighted's ramp (:601-608)
            } else {
                w = pw[u];
                dw = pdw[u];
                wt = ((float)i - CENTER) * w;  // compute_time_weighted (:601-608)
            }
        }
        vb[u] = v2f{sv.x * w, sv.y * w};   // zero beyond the window (:563-566)
        vd[u] = v2f{sv.x * dw, sv.y * dw};
        vt[u] = v2f{sv.x * wt, sv.y * wt};
    }
    frame_sync<LOGF>();  // imag[] (in B) is consumed
    fftp_dual<false, LOGF>(vb, vd, A, B, jf, tw);
    v2f bb[9], bd[9];
    float pn[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) pn[t] = a.bin_norm[(t < 8 || jf == 0) ? ju + (unsigned)T * (unsigned)t : 0u];
    frame_sync<LOGF>();  // the paired transform's last pass still reads A and B
    if constexpr (!BINS) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            bb[t] = vb[t];
            bd[t] = vd[t];
        }
        fftp<false, LOGF>(vt, A, B, jf, tw);
    } else {
        // natural-order copy of bins -SH ... F/2 + T of both spectra (slot SH + k = bin k; bins -m = F - m sit below slot SH)
        constexpr int SH = N / W;  // F / W
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            A[SH + jf + T * t] = vb[t];
            B[SH + jf + T * t] = vd[t];
        }
        if (jf >= T - SH) {
            A[SH - (T - jf)] = vb[15];
            B[SH - (T - jf)] = vd[15];
        }
        frame_sync<LOGF>();
        const float c0 = a.win_c0, half_c1 = 0.5f * a.win_c1, dscale = a.win_c1 * (3.14159265358979323846f / (float)W);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint32_t bin = ju + (unsigned)T * (unsigned)t;
            const bool mine = t < 8 || jf == 0;
            const uint32_t at = (mine ? bin : 0u) + (unsigned)SH;
            const v2f zm = A[at - SH], zp = A[at + SH], z2m = B[at - SH], z2p = B[at + SH];
            const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y}, z2s{z2m.x + z2p.x, z2m.y + z2p.y};
            bb[t] = v2f{c0 * vb[t].x + half_c1 * zs.x, c0 * vb[t].y + half_c1 * zs.y};
            bd[t] = v2f{-dscale * zd.y, dscale * zd.x};  // i c1 (pi / W) (Z[k - F/W] - Z[k + F/W])
            vt[t] = v2f{c0 * vd[t].x + half_c1 * z2s.x, c0 * vd[t].y + half_c1 * z2s.y};
        }
    }

    // ---- 4. reassignment + ordered compaction (bins jf + T t, t < 8, and bin F/2 on thread 0) ----------------------------
    omx_spectrogram_point pts[9];
    unsigned long long masks[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint32_t bin = ju + (unsigned)T * (unsigned)t;
        const bool keep = reassign_flat(bin, bb[t], bd[t], vt[t], pn[t], rc, pts[t]) && (t < 8 || jf == 0) && !silent;
        masks[t] = __ballot(keep);
        if (lane == 0) scan[t * WPF + wf] = (uint32_t)__popcll(masks[t]);
    }
    frame_sync<LOGF>();
    omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
    const uint32_t running = store_ordered<WPF>(masks, pts, scan, lane, wf, in_range, out);
    if (jf == 0 && in_range) *count_out = running;
}

template <int LOGW, int LOGF>
static void launch_zp(const StftFastArgs& a, const v2f* twF, hipStream_t stream) {
    using G = FftGeom<LOGF>;
    constexpr int F = G::FRAMES, WPF = G::T / 64;
    const size_t lds = (size_t)(2 * F * G::LDS + 256) * sizeof(v2f) + (size_t)F * 9 * WPF * sizeof(uint32_t) + (size_t)F * 2 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_zp_kernel<LOGW, LOGF, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_zp_kernel<LOGW, LOGF, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const uint32_t chunks = (a.n_cols + F - 1) / F;
    if (a.win_terms == 2)  // Hann / Hamming: window applied on the bins, four transforms per column
        hipLaunchKernelGGL((stft_reassigned_zp_kernel<LOGW, LOGF, true>), dim3(stream_column_grid(a.n_streams, chunks)), dim3(G::WG), lds, stream, a, twF);
    else
        hipLaunchKernelGGL((stft_reassigned_zp_kernel<LOGW, LOGF, false>), dim3(stream_column_grid(a.n_streams, chunks)), dim3(G::WG), lds, stream, a, twF);
}

// window 1024 / 2048 / 4096 zero-padded to 2048 / 4096 / 8192: `a.tw4096` = exp(-2 pi i k / W), `a.tw8192` = exp(-2 pi i k / 2W), twF = exp(-2 pi i k / F)
bool launch_stft_reassigned_zp(const StftFastArgs& a, uint32_t window, uint32_t fft_size, const v2f* twF, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return true;
    if (window == 1024 && fft_size == 2048) launch_zp<10, 11>(a, twF, stream);
    else if (window == 1024 && fft_size == 4096) launch_zp<10, 12>(a, twF, stream);
    else if (window == 2048 && fft_size == 4096) launch_zp<11, 12>(a, twF, stream);
    else if (window == 1024 && fft_size == 8192) launch_zp<10, 13>(a, twF, stream);
    else if (window == 2048 && fft_size == 8192) launch_zp<11, 13>(a, twF, stream);
    else if (window == 4096 && fft_size == 8192) launch_zp<12, 13>(a, twF, stream);
    else return false;
    return true;
}

// ================================================================================================
// K2b: reassigned columns for W = F = 16384 — the one GUI size whose fused form does not fit a CU (two 139 KiB LDS buffers, or
// 1024 threads under a 128-VGPR cap).  Same arithmetic as the fused kernels, cut into three kernels that hand the analytic
// slice and the three spectra over through an HBM scratch (0.45 MB per frame), processed in chunks of frames:
//   hilbert_big_kernel     packed real FFT of the 2N-sample window, single-IFFT Hilbert, analytic slice -> sv[frame][N]
//   windowed_big_kernel    (frame, q): FFT_N(sv * {w, w', t w}[q]) -> spec[q][frame][N/2 + 1]
//   reassign_big_kernel    per-bin reassignment + ordered compaction
// One frame per 1024-thread workgroup, one in-place 16384-point transform at a time.
// ================================================================================================
template <int LOGN>
__global__ __launch_bounds__(FftGeom<LOGN>::WG) void hilbert_big_kernel(StftFastArgs a, BigScratch sc) {
    using G = FftGeom<LOGN>;
    constexpr int N = G::N, T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* buf = reinterpret_cast<v2f*>(smem_raw);         // [G::LDS]
    v2f* tw2_lds = buf + G::LDS;                          // [256]
    float* hil = reinterpret_cast<float*>(tw2_lds + 256);  // [2]
    const uint32_t item = sc.first + blockIdx.x;
    const uint32_t s = item / a.n_cols, col = item % a.n_cols;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const uint64_t p0 = stft_tail(a, s) + (uint64_t)col * a.hop;
    if (col >= stft_cols(a, s) || a.last_nonzero[s] < (long long)p0) return;  // past the stream's count / silent column (:307-316): reassign_big_kernel emits it empty
    const char* ring_bytes = reinterpret_cast<const char*>(a.ring + (uint64_t)s * a.cap);
    const uint32_t bytemask = (uint32_t)(a.cap - 1) << 2;
    const uint32_t p32 = (uint32_t)p0;
    TwiddlesPow2<LOGN> tw;
    tw.tw2 = tw2_lds;
    tw.load(a.tw4096, ju);
    for (unsigned i = threadIdx.x; i < 256u; i += (unsigned)T) tw2_lds[i] = a.tw256[i];  // launched with T threads (T >= 64)
    v2f v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t qq = p32 + 2u * (ju + (unsigned)T * (unsigned)t);
        v[t] = v2f{*reinterpret_cast<const float*>(ring_bytes + ((qq << 2) & bytemask)),
                   *reinterpret_cast<const float*>(ring_bytes + (((qq + 1u) << 2) & bytemask))};
    }
    lds_workgroup_barrier();  // tw2_lds
    fftp_inplace<false, LOGN>(v, buf, j, tw);
    lds_workgroup_barrier();
#pragma unroll
    for (int t = 0; t < 16; ++t) buf[pad16(j + T * t)] = v[t];
    if (j == 0) {
        hil[0] = (v[0].x + v[0].y) * 0.5f;  // X[0] / 2
        hil[1] = (v[0].x - v[0].y) * 0.5f;  // X[N] / 2
    }
    lds_workgroup_barrier();
    v2f y[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const unsigned k = ju + (unsigned)T * (unsigned)t;
        const v2f z = v[t];
        const v2f zr = buf[pad16((int)(((unsigned)N - k) & (unsigned)(N - 1)))];
        const v2f sum{z.x + zr.x, z.y - zr.y};   // Zf[k] + conj Zf[N-k]  (the 1/2 lives in the twiddle table)
        const v2f dif{z.x - zr.x, z.y + zr.y};   // Zf[k] - conj Zf[N-k]
        const v2f w = a.tw8192[k];  // exp(-2 pi i k / 2N)
        y[t] = cmulc(sum, w) - cmul(dif, w);
        if (k == 0) y[t] = v2f{0.0f, 0.0f};
    }
    const float half_x0 = hil[0], half_xn = hil[1];
    lds_workgroup_barrier();  // partners are read from the buffer the inverse is about to overwrite
    fftp_inplace<true, LOGN>(y, buf, j, tw);  // y[t] = (Im a[2m], Im a[2m+1]), m = j + T t
    lds_workgroup_barrier();
    float* imag = reinterpret_cast<float*>(buf);  // N floats: Im analytic[N/2 .. 3N/2)
#pragma unroll
    for (int t = 4; t < 12; ++t) *reinterpret_cast<v2f*>(imag + 2 * (j + T * t - N / 4)) = y[t];
    lds_workgroup_barrier();
    const float parity = (j & 1) ? -half_xn : half_xn;
    v2f* out = sc.sv + (uint64_t)blockIdx.x * N;
    const uint32_t qe = p32 + (uint32_t)(N / 2) + ju;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const float xr = *reinterpret_cast<const float*>(ring_bytes + (((qe + (unsigned)T * (unsigned)u) << 2) & bytemask));
        out[j + T * u] = v2f{(float)N * xr - half_x0 + parity, imag[j + T * u]};
    }
}

template <int LOGN>
__global__ __launch_bounds__(FftGeom<LOGN>::WG) void windowed_big_kernel(StftFastArgs a, BigScratch sc) {
    using G = FftGeom<LOGN>;
    constexpr int N = G::N, T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* buf = reinterpret_cast<v2f*>(smem_raw);
    v2f* tw2_lds = buf + G::LDS;
    const uint32_t item = sc.first + blockIdx.x, q = blockIdx.y;
    const uint32_t s = item / a.n_cols, col = item % a.n_cols;
    if (col >= stft_cols(a, s) || a.last_nonzero[s] < (long long)(stft_tail(a, s) + (uint64_t)col * a.hop)) return;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    TwiddlesPow2<LOGN> tw;
    tw.tw2 = tw2_lds;
    tw.load(a.tw4096, ju);
    if (threadIdx.x < 256) tw2_lds[threadIdx.x] = a.tw256[threadIdx.x];
    const uint32_t W = a.window_size;  // == N unless the window is zero-padded to the transform (:334-342)
    const v2f* sv = sc.sv + (uint64_t)blockIdx.x * W;
    const bool bins = a.win_terms == 2;  // Hann / Hamming: q = 0 -> Z = FFT(s), q = 1 -> Z2 = FFT((n - c) s); the window is applied
                                         // on the bins by reassign_big_kernel (see stft4096_pair_kernels.hip)
    const float* win = q == 1 ? a.dwindow : a.window;
    const float center = (float)(W - 1u) * 0.5f;
    v2f v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const uint32_t i = ju + (unsigned)T * (unsigned)u;
        const uint32_t ic = i < W ? i : 0u;  // unconditional loads, selected afterwards
        const v2f x = sv[ic];
        float w = bins ? 1.0f : win[ic];
        if (bins ? q == 1 : q == 2) w = ((float)ic - center) * w;  // compute_time_weighted (:601-608)
        v[u] = i < W ? v2f{x.x * w, x.y * w} : v2f{0.0f, 0.0f};
    }
    lds_workgroup_barrier();  // tw2_lds
    fftp_inplace<false, LOGN>(v, buf, j, tw);
    if (bins) {  // bins -kBigHalo ... N/2 + kBigHalo (row slot = bin + kBigHalo): the window's cosine shifts by F / W <= 16 bins
        v2f* out = sc.spec + ((uint64_t)q * sc.count + blockIdx.x) * kBigRow<LOGN> + kBigHalo;
#pragma unroll
        for (int u = 0; u < 8; ++u) out[j + T * u] = v[u];
        if (j <= kBigHalo) out[N / 2 + j] = v[8];
        if (j >= T - kBigHalo) out[j - T] = v[15];  // bin N - m sits at slot -m
        return;
    }
    v2f* out = sc.spec + ((uint64_t)q * sc.count + blockIdx.x) * (N / 2 + 1);
#pragma unroll
    for (int u = 0; u < 8; ++u) out[j + T * u] = v[u];
    if (j == 0) out[N / 2] = v[8];
}

template <int LOGN>
__global__ __launch_bounds__(FftGeom<LOGN>::WG) void reassign_big_kernel(StftFastArgs a, BigScratch sc) {
    using G = FftGeom<LOGN>;
    constexpr int N = G::N, T = G::T, WPF = T / 64;
    __shared__ uint32_t scan[9 * WPF];
    const uint32_t item = sc.first + blockIdx.x;
    const uint32_t s = item / a.n_cols, col = item % a.n_cols;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const int lane = j & 63, wf = j >> 6;
    uint32_t* count_out = a.counts + (uint64_t)s * a.n_cols + col;
    if (col >= stft_cols(a, s) || a.last_nonzero[s] < (long long)(stft_tail(a, s) + (uint64_t)col * a.hop)) {
        if (j == 0) *count_out = 0;
        return;
    }
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};
    const bool bins = a.win_terms == 2;
    const uint64_t row = bins ? (uint64_t)kBigRow<LOGN> : (uint64_t)(N / 2 + 1);
    const uint64_t per = (uint64_t)sc.count * row;
    const v2f* sb = sc.spec + (uint64_t)blockIdx.x * row + (bins ? kBigHalo : 0);
    // window on the bins: cos(2 pi n / W) shifts an F-point spectrum by F / W bins
    const int shift = (int)((uint32_t)N / a.window_size);
    const float c0 = a.win_c0, half_c1 = 0.5f * a.win_c1, dscale = a.win_c1 * (3.14159265358979323846f / (float)a.window_size);
    omx_spectrogram_point pts[9];
    unsigned long long masks[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint32_t bin = (t < 8 || j == 0) ? ju + (unsigned)T * (unsigned)t : 0u;
        v2f b, d, tt;
        if (bins) {
            const v2f* z = sb + bin;
            const v2f* z2 = sb + per + bin;
            const v2f zc = z[0], zm = z[-shift], zp = z[shift], z2c = z2[0], z2m = z2[-shift], z2p = z2[shift];
            b = v2f{c0 * zc.x + half_c1 * (zm.x + zp.x), c0 * zc.y + half_c1 * (zm.y + zp.y)};
            d = v2f{-dscale * (zm.y - zp.y), dscale * (zm.x - zp.x)};  // i c1 (pi / W) (Z[k - F/W] - Z[k + F/W])
            tt = v2f{c0 * z2c.x + half_c1 * (z2m.x + z2p.x), c0 * z2c.y + half_c1 * (z2m.y + z2p.y)};
        } else {
            b = sb[bin];
            d = sb[per + bin];
            tt = sb[2 * per + bin];
        }
        const float norm = a.bin_norm[bin];
        bool keep = false;
        if (t < 8 || j == 0) keep = reassign_flat(bin, b, d, tt, norm, rc, pts[t]);
        masks[t] = __ballot(keep);
        if (lane == 0) scan[t * WPF + wf] = (uint32_t)__popcll(masks[t]);
    }
    lds_workgroup_barrier();
    omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
    uint32_t running = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        uint32_t before = running;
        for (int w = 0; w < WPF; ++w) {
            const uint32_t c = scan[t * WPF + w];
            if (w < wf) before += c;
            running += c;
        }
        if ((masks[t] >> lane) & 1ull) {
            const uint32_t pos = before + (uint32_t)__popcll(masks[t] & ((1ull << lane) - 1ull));
            *reinterpret_cast<omx_spectrogram_point*>(reinterpret_cast<char*>(out) + pos * 12u) = pts[t];
        }
    }
    if (j == 0) *count_out = running;
}

// bytes of scratch per frame of a chunk, and the launcher (frames [first, first + count) of the call)
uint64_t stft_big_scratch_bytes_per_frame() { return (uint64_t)(16384 + 3 * 8193 + 4 * kBigHalo) * sizeof(v2f); }
// LOGW = window (Hilbert pair on 2W samples), LOGF = transform; W == F unless zero-padded
template <int LOGW, int LOGF>
static void launch_big(const StftFastArgs& a, const v2f* twF, void* scratch, uint32_t first, uint32_t count, hipStream_t stream) {
    using GW = FftGeom<LOGW>;
    using GF = FftGeom<LOGF>;
    BigScratch sc{};
    sc.sv = reinterpret_cast<v2f*>(scratch);
    sc.spec = sc.sv + (uint64_t)count * GW::N;
    sc.first = first;
    sc.count = count;
    const size_t lds_w = (size_t)(GW::LDS + 256) * sizeof(v2f) + 2 * sizeof(float);
    const size_t lds_f = (size_t)(GF::LDS + 256) * sizeof(v2f) + 2 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hilbert_big_kernel<LOGW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(windowed_big_kernel<LOGF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
        attr_set = true;
    }
    StftFastArgs af = a;  // the windowed transforms run at F points: their twiddles are exp(-2 pi i k / F)
    if (twF) af.tw4096 = twF;
    const bool fused = LOGF == 14 && a.win_terms == 2;  // Hann / Hamming at 16384 points: windowed transforms + reassignment in one kernel
    if constexpr (LOGW == 14) launch_hilbert_16k(a, sc, fused, stream);  // stft16384_kernels.hip: 256 threads per frame, 4 x 4096-point transforms
    else hipLaunchKernelGGL(hilbert_big_kernel<LOGW>, dim3(count), dim3(GW::T), lds_w, stream, a, sc);
    if (fused) {
        launch_windowed_reassign_16k(af, sc, LOGW == 14, stream);
        return;
    }
    if constexpr (LOGF == 14) launch_windowed_16k(af, sc, stream);
    else hipLaunchKernelGGL(windowed_big_kernel<LOGF>, dim3(count, a.win_terms == 2 ? 2 : 3), dim3(GF::T), lds_f, stream, af, sc);
    hipLaunchKernelGGL(reassign_big_kernel<LOGF>, dim3(count), dim3(GF::T), 0, stream, af, sc);
}
void launch_stft_reassigned_16384(const StftFastArgs& a, void* scratch, uint32_t first, uint32_t count, hipStream_t stream) {
    if (count == 0) return;
    launch_big<14, 14>(a, nullptr, scratch, first, count, stream);
}
// window 1024 ... 8192 zero-padded to a 16384-point transform (zero padding 16 / 8 / 4 / 2 of the GUI): same three kernels
bool launch_stft_reassigned_zp_16384(const StftFastArgs& a, uint32_t window, const v2f* twF, void* scratch, uint32_t first, uint32_t count,
                                     hipStream_t stream) {
    if (count == 0) return true;
    switch (window) {
        case 1024: launch_big<10, 14>(a, twF, scratch, first, count, stream); return true;
        case 2048: launch_big<11, 14>(a, twF, scratch, first, count, stream); return true;
        case 4096: launch_big<12, 14>(a, twF, scratch, first, count, stream); return true;
        case 8192: launch_big<13, 14>(a, twF, scratch, first, count, stream); return true;
        default: return false;
    }
}
// ================================================================================================
// Zero padding beyond 16384 points (the GUI offers up to 32x of 1024 ... 16384-point windows: F = zp W up to 524288,
// reference src/ui/settings/spectrogram.rs:13, processor.rs:231).  A W-sample slice zero-padded to F = zp W points is zp
// independent W-point transforms of modulated copies of the slice:
//     X[zp q + r] = sum_n x[n] e^{-2 pi i n (zp q + r) / F} = FFT_W( x[n] e^{-2 pi i n r / F} )[q],     r = 0 ... zp - 1
// so no transform here is longer than the window.  Same three steps as above:
//   hilbert_big_kernel        analytic slice of every frame -> scratch                                        (unchanged)
//   windowed_residue_kernel   workgroup = (frame, window w / w' / t w, residue r): window x modulation in the time domain
//                             (processor.rs:334-342, :569-608), one W-point transform, bins zp q + r of that spectrum's row
//   reassign_stream_kernel    per-bin reassignment + ordered compaction over the F / 2 + 1 bins, 1024 at a time
// ================================================================================================
template <int LOGW>
__global__ __launch_bounds__(FftGeom<LOGW>::WG) void windowed_residue_kernel(StftFastArgs a, BigScratch sc, const v2f* __restrict__ twF, uint32_t zp) {
    using G = FftGeom<LOGW>;
    constexpr int N = G::N, T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* buf = reinterpret_cast<v2f*>(smem_raw);
    v2f* tw2_lds = buf + G::LDS;
    const uint32_t item = sc.first + blockIdx.x, q = blockIdx.y, r = blockIdx.z;
    const uint32_t s = item / a.n_cols, col = item % a.n_cols;
    if (col >= stft_cols(a, s) || a.last_nonzero[s] < (long long)(stft_tail(a, s) + (uint64_t)col * a.hop)) return;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    TwiddlesPow2<LOGW> tw;
    tw.tw2 = tw2_lds;
    tw.load(a.tw4096, ju);  // exp(-2 pi i k / W)
    for (unsigned i = threadIdx.x; i < 256u; i += (unsigned)T) tw2_lds[i] = a.tw256[i];
    const uint32_t F = (uint32_t)N * zp;
    const v2f* sv = sc.sv + (uint64_t)blockIdx.x * N;
    const float* win = q == 1 ? a.dwindow : a.window;
    const float center = (float)(N - 1) * 0.5f;
    v2f v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const uint32_t i = ju + (unsigned)T * (unsigned)u;
        const v2f x = sv[i];
        float w = win[i];
        if (q == 2) w = ((float)i - center) * w;  // compute_time_weighted (:601-608)
        const v2f m = twF[(i * r) & (F - 1u)];     // exp(-2 pi i n r / F)
        v[u] = cmul(v2f{x.x * w, x.y * w}, m);
    }
    lds_workgroup_barrier();  // tw2_lds
    fftp_inplace<false, LOGW>(v, buf, j, tw);
    v2f* out = sc.spec + ((uint64_t)q * sc.count + blockIdx.x) * (F / 2u + 1u);
#pragma unroll
    for (int u = 0; u < 8; ++u) out[(uint64_t)(ju + (unsigned)T * (unsigned)u) * zp + r] = v[u];   // bin zp q + r, q < W / 2
    if (j == 0 && r == 0) out[F / 2u] = v[8];                                                          // q = W / 2: the Nyquist bin
}

__global__ __launch_bounds__(1024) void reassign_stream_kernel(StftFastArgs a, BigScratch sc, uint32_t bins) {
    __shared__ uint32_t scan[16];
    const uint32_t item = sc.first + blockIdx.x;
    const uint32_t s = item / a.n_cols, col = item % a.n_cols;
    const unsigned ju = threadIdx.x;
    const int lane = (int)(ju & 63u), wf = (int)(ju >> 6);
    uint32_t* count_out = a.counts + (uint64_t)s * a.n_cols + col;
    if (col >= stft_cols(a, s) || a.last_nonzero[s] < (long long)(stft_tail(a, s) + (uint64_t)col * a.hop)) {
        if (ju == 0) *count_out = 0;
        return;
    }
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};
    const uint64_t per = (uint64_t)sc.count * bins;
    const v2f* sb = sc.spec + (uint64_t)blockIdx.x * bins;
    omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
    uint32_t running = 0;
    for (uint32_t b0 = 0; b0 < bins; b0 += 1024u) {
        const uint32_t bin = b0 + ju;
        const bool valid = bin < bins;
        const uint32_t bc = valid ? bin : 0u;
        omx_spectrogram_point pt;
        const bool keep = reassign_flat(bc, sb[bc], sb[per + bc], sb[2 * per + bc], a.bin_norm[bc], rc, pt) && valid;
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) scan[wf] = (uint32_t)__popcll(mask);
        lds_workgroup_barrier();
        uint32_t before = running, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t c = scan[w];
            if (w < wf) before += c;
            total += c;
        }
        if (keep) {
            const uint32_t pos = before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            *reinterpret_cast<omx_spectrogram_point*>(reinterpret_cast<char*>(out) + (uint64_t)pos * 12u) = pt;
        }
        running += total;
        lds_workgroup_barrier();  // scan[] is rewritten by the next tile
    }
    if (ju == 0) *count_out = running;
}

// Classic (non-reassigned) columns beyond 16384 points (processor.rs:350-380): the same decomposition on the DC-removed windowed REAL
// slice — workgroup = (column, residue r): (x - mean) w e^{-2 pi i n r / F}, one W-point transform, bins zp q + r <= F / 2 -> u16 codes.
// (The modulated slice is complex: the two-columns-per-transform packing of stft_classic_pow2_kernel does not apply.)
template <int LOGW>
__global__ __launch_bounds__(FftGeom<LOGW>::WG) void classic_residue_kernel(StftFastArgs a, uint16_t* __restrict__ codes, const v2f* __restrict__ twF,
                                                                           uint32_t zp) {
    using G = FftGeom<LOGW>;
    constexpr int N = G::N, T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* buf = reinterpret_cast<v2f*>(smem_raw);
    v2f* tw2_lds = buf + G::LDS;
    const uint32_t item = blockIdx.x, r = blockIdx.y;
    const uint32_t s = item / a.n_cols, col = item % a.n_cols;
    if (col >= stft_cols(a, s)) return;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const char* ring_bytes = reinterpret_cast<const char*>(a.ring + (uint64_t)s * a.cap);
    const uint32_t bytemask = (uint32_t)(a.cap - 1) << 2;
    const uint32_t p32 = (uint32_t)(stft_tail(a, s) + (uint64_t)col * a.hop);
    const uint32_t F = (uint32_t)N * zp;
    TwiddlesPow2<LOGW> tw;
    tw.tw2 = tw2_lds;
    tw.load(a.tw4096, ju);  // exp(-2 pi i k / W)
    for (unsigned i = threadIdx.x; i < 256u; i += (unsigned)T) tw2_lds[i] = a.tw256[i];
    float x[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) x[t] = *reinterpret_cast<const float*>(ring_bytes + (((p32 + ju + (unsigned)T * (unsigned)t) << 2) & bytemask));
    lds_workgroup_barrier();  // tw2_lds
    const float total = a.col_sums[(uint64_t)s * a.n_cols + col];  // the reference's sequential f32 sum (window_sum_kernels.hip)
    const float mean = total / (float)N;  // window.rs:76-79
    v2f v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t i = ju + (unsigned)T * (unsigned)t;
        const float xw = (x[t] - mean) * a.window[i];
        const v2f m = twF[(i * r) & (F - 1u)];
        v[t] = v2f{xw * m.x, xw * m.y};
    }
    fftp_inplace<false, LOGW>(v, buf, j, tw);
    uint16_t* out = codes + ((uint64_t)s * a.n_cols + col) * a.column_stride;
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        if (u == 8 && (j != 0 || r != 0)) break;  // q = W / 2 is the Nyquist bin, residue 0 only
        const uint32_t k = (ju + (unsigned)T * (unsigned)u) * zp + r;
        const v2f z = v[u];
        out[k] = classic_code((z.x * z.x + z.y * z.y) * a.bin_norm[k]);
    }
}
template <int LOGW>
static void launch_classic_residue_w(const StftFastArgs& a, uint16_t* codes, const v2f* twF, uint32_t zp, hipStream_t stream) {
    using G = FftGeom<LOGW>;
    const size_t lds = (size_t)(G::LDS + 256) * sizeof(v2f) + 16 * sizeof(float);
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(classic_residue_kernel<LOGW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    hipLaunchKernelGGL(classic_residue_kernel<LOGW>, dim3(a.n_streams * a.n_cols, zp), dim3(G::T), lds, stream, a, codes, twF, zp);
}
bool launch_stft_classic_residue(const StftFastArgs& a, uint16_t* codes, uint32_t window, uint32_t zp, const v2f* twF, hipStream_t stream) {
    if (a.n_streams == 0 || a.n_cols == 0) return true;
    if (window == 1024 || window == 2048 || window == 4096 || window == 8192 || window == 16384) launch_classic_window_sums(a, window, stream);
    switch (window) {
        case 1024: launch_classic_residue_w<10>(a, codes, twF, zp, stream); return true;
        case 2048: launch_classic_residue_w<11>(a, codes, twF, zp, stream); return true;
        case 4096: launch_classic_residue_w<12>(a, codes, twF, zp, stream); return true;
        case 8192: launch_classic_residue_w<13>(a, codes, twF, zp, stream); return true;
        case 16384: launch_classic_residue_w<14>(a, codes, twF, zp, stream); return true;
        default: return false;
    }
}

uint64_t stft_residue_scratch_bytes_per_frame(uint32_t window, uint32_t fft_size) {
    return ((uint64_t)window + 3ull * ((uint64_t)fft_size / 2 + 1)) * sizeof(v2f);
}
template <int LOGW>
static void launch_residue(const StftFastArgs& a, const v2f* twF, uint32_t zp, void* scratch, uint32_t first, uint32_t count, hipStream_t stream) {
    using GW = FftGeom<LOGW>;
    BigScratch sc{};
    sc.sv = reinterpret_cast<v2f*>(scratch);
    sc.spec = sc.sv + (uint64_t)count * GW::N;
    sc.first = first;
    sc.count = count;
    const size_t lds_w = (size_t)(GW::LDS + 256) * sizeof(v2f) + 2 * sizeof(float);
    static std::once_flag attr_once;  // (one per window size)
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hilbert_big_kernel<LOGW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(windowed_residue_kernel<LOGW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);
    });
    if constexpr (LOGW == 14) launch_hilbert_16k(a, sc, false, stream);
    else hipLaunchKernelGGL(hilbert_big_kernel<LOGW>, dim3(count), dim3(GW::T), lds_w, stream, a, sc);
    hipLaunchKernelGGL(windowed_residue_kernel<LOGW>, dim3(count, 3, zp), dim3(GW::T), lds_w, stream, a, sc, twF, zp);
    hipLaunchKernelGGL(reassign_stream_kernel, dim3(count), dim3(1024), 0, stream, a, sc, (uint32_t)GW::N * zp / 2u + 1u);
}
// window 1024 ... 16384 zero-padded to zp * window > 16384 points; frames [first, first + count) of the call
bool launch_stft_reassigned_residue(const StftFastArgs& a, uint32_t window, uint32_t zp, const v2f* twF, void* scratch, uint32_t first, uint32_t count,
                                    hipStream_t stream) {
    if (count == 0) return true;
    switch (window) {
        case 1024: launch_residue<10>(a, twF, zp, scratch, first, count, stream); return true;
        case 2048: launch_residue<11>(a, twF, zp, scratch, first, count, stream); return true;
        case 4096: launch_residue<12>(a, twF, zp, scratch, first, count, stream); return true;
        case 8192: launch_residue<13>(a, twF, zp, scratch, first, count, stream); return true;
        case 16384: launch_residue<14>(a, twF, zp, scratch, first, count, stream); return true;
        default: return false;
    }
}
// tuning only (OMX_K2_VARIANT=31): the 4096-point shape through the same three kernels, to price the fused kernel against
// simple high-occupancy ones
void launch_stft_reassigned_4096_split(const StftFastArgs& a, void* scratch, uint32_t first, uint32_t count, hipStream_t stream) {
    if (count == 0) return;
    launch_big<12, 12>(a, nullptr, scratch, first, count, stream);
}

template <int LOGN>
static void launch_pow2(const StftFastArgs& a, hipStream_t stream) {
    using G = FftGeom<LOGN>;
    const bool bins = a.win_terms == 2;  // Hann / Hamming: window applied on the bins, four transforms per column
    constexpr int F = G::FRAMES, WPF = G::T / 64;
    const size_t lds = (size_t)(2 * F * G::LDS + 256) * sizeof(v2f) + (size_t)F * 9 * WPF * sizeof(uint32_t) + (size_t)F * 2 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_pow2_kernel<LOGN, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_pow2_kernel<LOGN, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const uint32_t chunks = (a.n_cols + F - 1) / F;
    if (bins) hipLaunchKernelGGL((stft_reassigned_pow2_kernel<LOGN, true>), dim3(stream_column_grid(a.n_streams, chunks)), dim3(G::WG), lds, stream, a);
    else hipLaunchKernelGGL((stft_reassigned_pow2_kernel<LOGN, false>), dim3(stream_column_grid(a.n_streams, chunks)), dim3(G::WG), lds, stream, a);
}

// fft_size = 1024, 2048, 4096 or 8192 (`a.tw4096` = exp(-2 pi i k / N), `a.tw8192` = exp(-2 pi i k / 2N), N entries each)

// ================================================================================================
// K2p-tri (round 4): the three-workgroups-per-CU form of stft4096_tri_kernels.hip for W = F = 1024 / 2048 — the reference's default
// spectrogram shape is 2048 / hop 64.  A frame slot (T = N/16 threads) carries two consecutive columns through ONE padded LDS buffer
// time-shared by the two chains of every dual transform (staggered by half a pass), column b's analytic slice waits in LDS as its
// imaginary half, FFT(t w s) is windowed in the time domain (twindow) and every window of the reference (a cosine sum of TERMS
// terms) is applied on the bins of Z: 51 KiB of LDS and <= 168 VGPR per 256-thread workgroup -> three workgroups per CU where the
// pair kernel above (two buffers per slot, 246 VGPR) fits two.
// ================================================================================================
// the Hilbert spectrum of element jf + T t, t = TT ... 15, with its 2N-point twiddle formed on the spot from the thread's table value
// (a resident run of sixteen would cost 32 registers across the phase)
template <int TT, int PS>
__device__ __forceinline__ void hilbert_steps_impl(v2f (&y)[16], const v2f (&v)[16], const v2f* Xp, v2f w8_base, int jf) {
    if constexpr (TT < 16) {
        const v2f w8 = rotate128<4 * TT>(w8_base);
        const v2f z = v[TT], zr = Xp[PS * (15 - TT)];
        const v2f sum{z.x + zr.x, z.y - zr.y}, dif{z.x - zr.x, z.y + zr.y};
        y[TT] = cmulc(sum, w8) - cmul(dif, w8);
        if (TT == 0 && jf == 0) y[TT] = v2f{0.0f, 0.0f};
        hilbert_steps_impl<TT + 1, PS>(y, v, Xp, w8_base, jf);
    }
}

template <bool INV, int LOGN>
__device__ __forceinline__ void tri_last_pass(v2f (&v)[16], const TwiddlesPow2<LOGN>& tw) {  // fftp_pass3's arithmetic, inputs already in registers
    using G = FftGeom<LOGN>;
#pragma unroll
    for (int u = G::M; u < 16; ++u) v[u] = twmul<INV>(v[u], tw.tw3[u - 1]);
    if constexpr (G::R3 == 8) {
        dft8<INV>(v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14]);
        dft8<INV>(v[1], v[3], v[5], v[7], v[9], v[11], v[13], v[15]);
    } else {
        static_assert(G::R3 == 4, "1024 / 2048 points");
#pragma unroll
        for (int m = 0; m < 4; ++m) dft4<INV>(v[m], v[m + 4], v[m + 8], v[m + 12]);
    }
}
// two N-point transforms through ONE buffer: x[jf + T t] in a[t] / b[t] on entry, X[jf + T t] on return; the caller has a sync
// between its last use of X and this call, and on return other threads may still be reading X (chain b's last-pass inputs)
template <bool INV, int LOGN>
__device__ __forceinline__ void tri_dual_pow2(v2f (&a)[16], v2f (&b)[16], v2f* X, int jf, const TwiddlesPow2<LOGN>& tw) {
    using G = FftGeom<LOGN>;
    constexpr int T = G::T, PS = T + T / 16;
    const unsigned k = (unsigned)jf & 15u;
    auto write1 = [&](const v2f (&v)[16]) {
        const int base = 17 * jf;
#pragma unroll
        for (int t = 0; t < 16; ++t) X[base + t] = v[DFT16_OUT(t)];
    };
    auto write2 = [&](const v2f (&v)[16]) {
        const int base = (jf >> 4) * 272 + (int)k;
#pragma unroll
        for (int t = 0; t < 16; ++t) X[base + 17 * t] = v[DFT16_OUT(t)];
    };
    auto read = [&](v2f (&v)[16]) {
        const int base = pad16(jf);
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = X[base + PS * t];
    };
    fftp_dft16<INV>(a);
    write1(a);
    frame_sync<LOGN>();
    read(a);
    fftp_dft16<INV>(b);
    frame_sync<LOGN>();  // every pass-2 input of chain a is in registers
    write1(b);
    fftp_tw_dft16<INV>(a, [&](int t) { return tw.tw2[k * (unsigned)t]; });
    frame_sync<LOGN>();
    read(b);
    frame_sync<LOGN>();
    write2(a);
    fftp_tw_dft16<INV>(b, [&](int t) { return tw.tw2[k * (unsigned)t]; });
    frame_sync<LOGN>();
    read(a);
    frame_sync<LOGN>();
    write2(b);
    tri_last_pass<INV, LOGN>(a, tw);
    frame_sync<LOGN>();
    read(b);
    tri_last_pass<INV, LOGN>(b, tw);
}

template <int LOGN, int TERMS>
__global__ __launch_bounds__(256, 3) void stft_reassigned_pow2_tri_kernel(StftFastArgs a) {
    using G = FftGeom<LOGN>;
    constexpr int N = G::N, T = G::T, F = G::FRAMES, WPF = T / 64, PS = T + T / 16;
    static_assert(LOGN == 10 || LOGN == 11, "1024 / 2048 points");
    constexpr int SLOT_FLOATS = 2 * G::LDS + N;  // the transform buffer, then Im analytic[N/2 + i] of column b
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    v2f* tw2_lds = reinterpret_cast<v2f*>(lds + F * SLOT_FLOATS);      // [256]
    uint32_t* scan_all = reinterpret_cast<uint32_t*>(tw2_lds + 256);   // [F][9][WPF]
    float* hil_all = reinterpret_cast<float*>(scan_all + F * 9 * WPF); // [F][4]

    const uint32_t pairs = (a.n_cols + 1u) / 2u, chunks = (pairs + F - 1) / F;
    const uint32_t blk = blockIdx.x, xcd = blk & 7u, q = blk >> 3;
    const uint32_t s = (q / chunks) * 8u + xcd, chunk = q % chunks;
    if (s >= a.n_streams) return;
    // the frame slot is wave-uniform (T = 64 or 128 threads per slot): taken through readfirstlane so that everything derived from it —
    // column indices, ring offsets, the buffer descriptors of the loads below — lives in scalar registers
    const int fs = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / T)), jf = threadIdx.x % T;
    const unsigned ju = (unsigned)jf;
    const int lane = threadIdx.x & 63, wf = jf >> 6;
    const int wf_u = __builtin_amdgcn_readfirstlane(wf);
    v2f* X = reinterpret_cast<v2f*>(lds + fs * SLOT_FLOATS);
    float* imb = lds + fs * SLOT_FLOATS + 2 * G::LDS;
    uint32_t* scan = scan_all + fs * 9 * WPF;
    float* hil = hil_all + fs * 4;
    const uint32_t n_cols_s = stft_cols(a, s), pairs_s = (n_cols_s + 1u) / 2u;
    if (chunk * F >= pairs_s) return;  // (whole workgroup: every slot is past this stream's columns)
    const uint32_t pair_raw = chunk * F + (uint32_t)fs;
    const bool in_range = pair_raw < pairs_s;
    const uint32_t pair = in_range ? pair_raw : pairs_s - 1u;  // idle slots shadow the last pair (syncs stay uniform)
    const uint32_t col0 = 2u * pair;
    const bool have1 = col0 + 1u < n_cols_s;
    const uint32_t col1 = have1 ? col0 + 1u : col0;  // an odd tail computes column 0 twice and stores it once

    const float* ring = a.ring + (uint64_t)s * a.cap;
    const uint32_t mask32 = (uint32_t)(a.cap - 1);
    const long long last_nonzero = a.last_nonzero[s];
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};
    const uint64_t tail_s = stft_tail(a, s);
    const uint64_t p0a = tail_s + (uint64_t)col0 * a.hop, p0b = tail_s + (uint64_t)col1 * a.hop;
    // silent fast path (:307-316).  The first slot's first column has the smallest p0: if it is silent, every column here is.
    const uint64_t p0_first = tail_s + (uint64_t)(2u * chunk * F) * a.hop;
    if (last_nonzero < (long long)p0_first) {
        if (jf == 0 && in_range) {
            a.counts[(uint64_t)s * a.n_cols + col0] = 0;
            if (have1) a.counts[(uint64_t)s * a.n_cols + col1] = 0;
        }
        return;
    }
    const bool silent_a = last_nonzero < (long long)p0a, silent_b = last_nonzero < (long long)p0b;  // computed anyway, emitted empty

    TwiddlesPow2<LOGN> tw;
    tw.tw2 = tw2_lds;
    tw.load(a.tw4096, ju);  // `tw4096` carries exp(-2 pi i k / N) for this N
    if (threadIdx.x < 256) tw2_lds[threadIdx.x] = a.tw256[threadIdx.x];
    const v2f w8_base = a.tw8192[ju];  // exp(-2 pi i jf / 2N) / 2

    // ---- 1. packed real FFTs of the two 2N-sample windows ------------------------------------------------------------------------------
    const uint32_t pa32 = (uint32_t)p0a, pb32 = (uint32_t)p0b;
    // Buffer-addressed loads (buffer_device.hpp) while the two windows lie in one piece of the ring: one per-lane byte offset and scalar
    // steps instead of a 64-bit vector address per load — the addresses of the 96 ring / table loads are what spilled here (48 ... 76 B
    // of scratch per lane until round 5, profiles/r14_kernel_resources.txt)
    const uint32_t off_a = pa32 & mask32, hop_bytes = (pb32 - pa32) * 4u;
    const bool direct = (uint64_t)off_a + (uint64_t)(pb32 - pa32) + 2ull * N <= a.cap && ((p0a | p0b) & 1ull) == 0;
    const GlobalBuffer windowb = global_buffer(ring + off_a, hop_bytes + 2u * (uint32_t)N * 4u);
    const GlobalBuffer normb = global_buffer(a.bin_norm, (uint32_t)(N / 2 + 1) * 4u), twinb = global_buffer(a.twindow, (uint32_t)N * 4u);
    v2f va[16], vb[16];
    if (direct) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            va[t] = load_v2f(windowb, ju * 8u, 8u * (unsigned)T * (unsigned)t);
            vb[t] = load_v2f(windowb, ju * 8u, hop_bytes + 8u * (unsigned)T * (unsigned)t);
        }
    } else if (((p0a | p0b) & 1ull) == 0) {  // pairs are 8-byte aligned and never straddle the ring wrap
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            va[t] = *reinterpret_cast<const v2f*>(ring + ((pa32 + 2u * (ju + (unsigned)T * (unsigned)t)) & mask32));
            vb[t] = *reinterpret_cast<const v2f*>(ring + ((pb32 + 2u * (ju + (unsigned)T * (unsigned)t)) & mask32));
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t qa = pa32 + 2u * (ju + (unsigned)T * (unsigned)t), qb = pb32 + 2u * (ju + (unsigned)T * (unsigned)t);
            va[t] = v2f{ring[qa & mask32], ring[(qa + 1u) & mask32]};
            vb[t] = v2f{ring[qb & mask32], ring[(qb + 1u) & mask32]};
        }
    }
    lds_workgroup_barrier();  // tw2_lds (shared by every slot)
    tri_dual_pow2<false, LOGN>(va, vb, X, jf, tw);  // v[t] = Zf[jf + T t]

    // ---- 2. Hilbert transform with one half-length inverse per column, one column at a time through X ----------------------------------
    const int part = (jf ? pad16(N - jf) : N + N / 16) - PS * 15;  // (thread 0, t = 0: one slot past X, inside imb; not used)
    auto hilbert_spectrum = [&](v2f (&y)[16], const v2f (&v)[16]) {
        hilbert_steps_impl<0, PS>(y, v, X + part, w8_base, jf);
    };
    v2f ya[16], yb[16];
    frame_sync<LOGN>();  // the last pass of chain b still reads X
#pragma unroll
    for (int t = 0; t < 16; ++t) X[pad16(jf) + PS * t] = va[t];
    if (jf == 0) {
        hil[0] = (va[0].x + va[0].y) * 0.5f;  // X[0] / 2
        hil[1] = (va[0].x - va[0].y) * 0.5f;  // X[N] / 2
        hil[2] = (vb[0].x + vb[0].y) * 0.5f;
        hil[3] = (vb[0].x - vb[0].y) * 0.5f;
    }
    frame_sync<LOGN>();
    hilbert_spectrum(ya, va);
    frame_sync<LOGN>();
#pragma unroll
    for (int t = 0; t < 16; ++t) X[pad16(jf) + PS * t] = vb[t];
    frame_sync<LOGN>();
    hilbert_spectrum(yb, vb);
    const float half_x0a = hil[0], half_xna = hil[1], half_x0b = hil[2], half_xnb = hil[3];
    frame_sync<LOGN>();
    tri_dual_pow2<true, LOGN>(ya, yb, X, jf, tw);  // y[t] = (Im a[2m], Im a[2m+1]), m = jf + T t

    // ---- 3. analytic slices s[i] = analytic[N/2 + i], i = jf + T t -------------------------------------------------------------------------
    auto load_real_half = [&](float (&xr)[16], uint32_t col_bytes, uint32_t p32) {
        if (direct) {
#pragma unroll
            for (int t = 0; t < 16; ++t) xr[t] = load_f32(windowb, ju * 4u, col_bytes + 2u * (uint32_t)N + 4u * (unsigned)T * (unsigned)t);
        } else {
            const uint32_t q0 = p32 + (uint32_t)(N / 2) + ju;
#pragma unroll
            for (int t = 0; t < 16; ++t) xr[t] = ring[(q0 + (unsigned)T * (unsigned)t) & mask32];
        }
    };
    auto load_twindow = [&](float (&twin)[16]) {
#pragma unroll
        for (int t = 0; t < 16; ++t) twin[t] = load_f32(twinb, ju * 4u, 4u * (unsigned)T * (unsigned)t);
    };
    float xra[16], twina[16];
    load_real_half(xra, 0u, pa32);
    load_twindow(twina);
    frame_sync<LOGN>();
    float* imag_a = reinterpret_cast<float*>(X);
#pragma unroll
    for (int t = 4; t < 12; ++t) {
        *reinterpret_cast<v2f*>(imag_a + 2 * (jf + T * t - N / 4)) = ya[t];
        *reinterpret_cast<v2f*>(imb + 2 * (jf + T * t - N / 4)) = yb[t];
    }
    frame_sync<LOGN>();
    constexpr int REACH = TERMS > 1 ? TERMS - 1 : 1;
    const float c0 = a.cos_c[0];
    float half_c[REACH], dscale[REACH];
#pragma unroll
    for (int m = 1; m <= REACH; ++m) {
        half_c[m - 1] = TERMS > 1 ? 0.5f * a.cos_c[m] : 0.0f;
        dscale[m - 1] = TERMS > 1 ? a.cos_c[m] * ((float)m * 3.14159265358979323846f / (float)N) : 0.0f;
    }
    v2f* lin_z = X;  // natural-order bins -3 ... N/2 + T of Z (slot 3 + k)

    // ---- 4. per column: Z = FFT(s), FFT(t w s) as one dual transform; w and w' applied on the bins of Z ------------------------------------
    auto column = [&](const float (&xr)[16], const float (&twin)[16], const float* imag, float half_x0, float half_xn, bool silent, bool store,
                      uint32_t col) {
        v2f z[16], z2[16];
        {
            const float par = (jf & 1) ? -half_xn : half_xn;  // (-1)^n: n = N/2 + i has jf's parity
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                z[t] = v2f{(float)N * xr[t] - half_x0 + par, imag[jf + T * t]};
                z2[t] = v2f{z[t].x * twin[t], z[t].y * twin[t]};
            }
        }
        frame_sync<LOGN>();  // the slice reads above / the previous column's neighbour reads still use X
        tri_dual_pow2<false, LOGN>(z, z2, X, jf, tw);
        float pn[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) pn[t] = load_f32(normb, ju * 4u, 4u * (unsigned)T * (unsigned)t);  // (t = 8, jf > 0: past the table, reads 0, not used)
        frame_sync<LOGN>();  // the last pass still reads X
#pragma unroll
        for (int t = 0; t < 8; ++t) lin_z[3 + jf + T * t] = z[t];
        if (wf_u == 0) lin_z[3 + jf + 8 * T] = z[8];  // bins N/2 ... N/2 + 63: the Nyquist bin and its upper neighbours
        if (jf >= T - 3) lin_z[jf - (T - 3)] = z[15];  // bins -3 ... -1 = bins N - 3 ... N - 1
        frame_sync<LOGN>();
        omx_spectrogram_point pts[9];
        unsigned long long masks[9];
        auto bins = [&](auto first, auto last) {
            constexpr int T0 = decltype(first)::value, T1 = decltype(last)::value;
            v2f nzm[T1 - T0][REACH], nzp[T1 - T0][REACH];
            if constexpr (TERMS > 1) {
#pragma unroll
                for (int t = T0; t < T1; ++t)
#pragma unroll
                    for (int m = 1; m <= REACH; ++m) {
                        nzm[t - T0][m - 1] = lin_z[3 + jf + T * t - m];
                        nzp[t - T0][m - 1] = lin_z[3 + jf + T * t + m];
                    }
            }
#pragma unroll
            for (int t = T0; t < T1; ++t) {
                const uint32_t bin = ju + (unsigned)T * (unsigned)t;
                v2f bb{c0 * z[t].x, c0 * z[t].y}, bd{0.0f, 0.0f};
                if constexpr (TERMS == 2) {
                    const v2f zm = nzm[t - T0][0], zp = nzp[t - T0][0];
                    const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y};
                    bb = v2f{c0 * z[t].x + half_c[0] * zs.x, c0 * z[t].y + half_c[0] * zs.y};
                    bd = v2f{-dscale[0] * zd.y, dscale[0] * zd.x};  // i c1 (pi / N) (Z[k-1] - Z[k+1])
                } else if constexpr (TERMS > 2) {
#pragma unroll
                    for (int m = 1; m <= REACH; ++m) {
                        const v2f zm = nzm[t - T0][m - 1], zp = nzp[t - T0][m - 1];
                        const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y};
                        bb = v2f{bb.x + half_c[m - 1] * zs.x, bb.y + half_c[m - 1] * zs.y};
                        bd = v2f{bd.x - dscale[m - 1] * zd.y, bd.y + dscale[m - 1] * zd.x};
                    }
                }
                const bool keep = reassign_flat(bin, bb, bd, z2[t], pn[t], rc, pts[t]) && (t < 8 || jf == 0) && !silent;
                masks[t] = __ballot(keep);
                if (lane == 0) scan[t * WPF + wf] = (uint32_t)__popcll(masks[t]);
            }
        };
        if constexpr (TERMS <= 2) {
            bins(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
            bins(std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});
        } else {
            bins(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            bins(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            bins(std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{});
            bins(std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{});
        }
        if (wf_u == 0) {
            bins(std::integral_constant<int, 8>{}, std::integral_constant<int, 9>{});
        } else {
            masks[8] = 0ull;
            if (lane == 0) scan[8 * WPF + wf] = 0u;
        }
        frame_sync<LOGN>();
        omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
        const uint32_t running = store_ordered<WPF>(masks, pts, scan, lane, wf, store, out);
        if (jf == 0 && store) a.counts[(uint64_t)s * a.n_cols + col] = running;
    };
    column(xra, twina, imag_a, half_x0a, half_xna, silent_a, in_range, col0);
    {
        float xrb[16], twinb[16];
        load_real_half(xrb, hop_bytes, pb32);
        load_twindow(twinb);
        column(xrb, twinb, imb, half_x0b, half_xnb, silent_b, in_range && have1, col1);
    }
}

template <int LOGN, int TERMS>
static void launch_pow2_tri_terms(const StftFastArgs& a, hipStream_t stream) {
    using G = FftGeom<LOGN>;
    constexpr int F = G::FRAMES, WPF = G::T / 64;
    const size_t lds = (size_t)F * (2 * G::LDS + G::N) * sizeof(float) + 256 * sizeof(v2f) + (size_t)F * 9 * WPF * sizeof(uint32_t) +
                       (size_t)F * 4 * sizeof(float);
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_pow2_tri_kernel<LOGN, TERMS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    const uint32_t pairs = (a.n_cols + 1u) / 2u;
    hipLaunchKernelGGL((stft_reassigned_pow2_tri_kernel<LOGN, TERMS>), dim3(stream_column_grid(a.n_streams, (pairs + F - 1) / F)), dim3(256), lds,
                       stream, a);
}
template <int LOGN>
static void launch_pow2_tri(const StftFastArgs& a, hipStream_t stream) {
    switch (a.cos_terms) {
        case 1: launch_pow2_tri_terms<LOGN, 1>(a, stream); break;
        case 2: launch_pow2_tri_terms<LOGN, 2>(a, stream); break;
        case 3: launch_pow2_tri_terms<LOGN, 3>(a, stream); break;
        default: launch_pow2_tri_terms<LOGN, 4>(a, stream); break;
    }
}

void launch_stft_reassigned_pow2(const StftFastArgs& a, uint32_t fft_size, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return;
    static const bool one_column_slots = [] { const char* e = tuning_env("OMX_POW2_SINGLE"); return e && atoi(e) == 1; }();  // A/B: the one-column-per-slot kernel
#ifdef POW2_FORCE_PAIR  // A/B build (tools/build_ab.sh): the round-2 pair kernel
    static const bool pair_form = true;
#else
    static const bool pair_form = [] { const char* e = tuning_env("OMX_POW2_PAIR"); return e && atoi(e) == 1; }();  // tuning build
#endif
    if (a.cos_terms >= 1 && a.cos_terms <= 4 && !one_column_slots && !pair_form && (fft_size == 1024 || fft_size == 2048)) {
        // round 4: three workgroups per CU, every window of the reference on the bins
        if (fft_size == 1024) launch_pow2_tri<10>(a, stream);
        else launch_pow2_tri<11>(a, stream);
        return;
    }
#if defined(OMX_TUNING) || defined(POW2_FORCE_PAIR)
    if (a.win_terms == 2 && !one_column_slots && (fft_size == 1024 || fft_size == 2048)) {  // Hann / Hamming: two columns per slot
        if (fft_size == 1024) launch_pow2_pair<10>(a, stream);
        else launch_pow2_pair<11>(a, stream);
        return;
    }
#endif
    switch (fft_size) {
        case 1024: launch_pow2<10>(a, stream); break;
        case 2048: launch_pow2<11>(a, stream); break;
        case 4096: launch_pow2<12>(a, stream); break;
        case 8192: launch_pow2<13>(a, stream); break;
        default: break;
    }
}

}  // namespace omx
