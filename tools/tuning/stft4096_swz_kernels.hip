// TUNING BUILD ONLY (make TUNING=1; OMX_K2_VARIANT = 50 / 51): two structural alternatives to the shipped pair kernel
// (stft4096_pair_kernels.hip), kept as measured negative results.  Both compute the same columns bit for bit (the GPU parity suite
// passes with either selected) and both were slower on MI355X, 65 536 frames per launch, same box:
//   shipped pair kernel                         1.71 ms   25 barriers per pair, LDS index-active 4.07e8 cycles, conflicts 4 %
//   50: swizzled pair kernel (below)            1.76 ms   13 barriers per pair, LDS index-active 3.60e8 cycles, conflicts 7 %
//   51: one column per workgroup, 3 per CU      2.61 ms   13 barriers per column, 12 wavefronts per CU
// What they change: the thread <-> index-digit assignment of the radix-16 passes, so that one of the two LDS exchanges of a
// transform stays inside a 16-lane row (no workgroup barrier), pass 2 runs in place, the natural-order copy of a spectrum is
// written to slots private to the thread that owns them (no barrier), and every read of the transform is a conflict-free
// single ds_read_b64 (layout 16 x + y + 257 z against the gfx950 banking rules).  The conclusion they bought: the pair kernel
// is bound neither by barriers nor by LDS cycles — with ONE workgroup per CU it takes 3.02 ms against 1.76 with two, which is
// what independent wavefronts that can each issue VALU work 32 % of the time predict (1 - 0.68^2 = 54 % busy) — and a
// single-transform-per-wavefront form at 12 wavefronts per CU loses more per wavefront (no second dependency chain, a global
// twiddle read per pass-3) than the third wavefront per SIMD returns.
#ifdef OMX_TUNING
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

// ---- the two dual transforms of this kernel -----------------------------------------------------------------------------
// Index digits of a 4096-point transform: n = n0 + 16 n1 + 256 n2, k = k0 + 16 k1 + 256 k2; pass 1 is the DFT over n2 (-> k0),
// pass 2 over n1 (-> k1, twiddle w256^(n1 k0)), pass 3 over n0 (-> k2, twiddle w4096^(n0 (k0 + 16 k1))).  Same butterflies, same
// twiddles, same operation order as fft4096t_dual (fft_device.hpp) — what changes is WHICH thread owns which digits, chosen so
// that one of the two LDS exchanges of a transform stays inside a 16-lane row of a wavefront and pass 2 overwrites exactly the
// slots it read: one workgroup barrier per transform instead of three (the LDS operations of one wavefront execute in issue
// order, so a wave-private exchange needs a compiler fence only).  swz(j) swaps the two hex digits of a thread index.
//   natural -> swizzled (forward, windowed):  thread j holds x[j + 256 t] on entry and X[swz(j) + 256 t] on return.
//       pass 1 -> 2 crosses wavefronts (barrier); pass 2 -> 3 is row-private.  LDS slot of digits (x, y, z) = 16 x + y + 257 z
//       (x = k0; y = n1, then k1; z = n0, then k2).  Conflict-free under the gfx950 rules for every access of the transform
//       (ds_read_b64: 32-lane halves, 8-byte slot index mod 32; ds_write_b64: 16-lane groups, mod 16): pass-1 writes have lanes
//       over z (257 z = z mod 16), pass-2 reads lanes (z, x pair) -> z + 16 h, pass-3 reads lanes (y, x pair) -> y + 16 h.  The
//       slots a thread reads in pass 3 are written by nobody else until the next barrier, so the natural-order copy the next
//       step needs is written back to them without a barrier either.
//   swizzled -> natural (inverse):  thread j holds Y[swz(j) + 256 t] on entry and y[j + 256 t] on return.
//       pass 1 -> 2 is row-private; pass 2 -> 3 crosses wavefronts (barrier).  Slots: the padded layout of fft_device.hpp.
// Reads walk their sixteen slots through a laundered base register: off one base hipcc fuses neighbouring ds_read_b64 into
// ds_read2_b64, which moves half the bytes per LDS cycle and is banked by the 16-lane rule.
// The caller keeps a barrier between the last reads of a buffer and the next transform's pass-1 writes.
constexpr int SWZ_LDS = 4360;  // complex slots per buffer: 16 * 257 (natural -> swizzled) and 4352 (inverse) fit; not a multiple of 64,
                               // so reads of the same slot of A and B are not fused into ds_read2st64_b64 either
__device__ __forceinline__ unsigned swz(unsigned j) { return (j >> 4) | ((j & 15u) << 4); }
__device__ __forceinline__ void row_sync() {  // orders the LDS accesses of ONE wavefront (see frame_sync in fft_pow2_device.hpp)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void launder(unsigned& byte_offset) {  // (an offset, not a pointer: the pointer keeps its LDS address space)
    asm("" : "+v"(byte_offset));
}
__device__ __forceinline__ v2f lds_at(const v2f* buf, unsigned byte_offset) {
    return *reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(buf) + byte_offset);
}

// natural in -> swizzled out; tw3[t - 1] = exp(-+2 pi i swz(j) t / 4096)
template <bool INV>
__device__ __forceinline__ void dual_nat_to_swz(v2f (&v0)[16], v2f (&v1)[16], v2f* A, v2f* B, int j, const v2f* tw2, const v2f (&tw3)[15]) {
    const int lo = j & 15, hi = j >> 4;
    dft16<INV>(v0);
    dft16<INV>(v1);
    {
        const int base = hi + 257 * lo;  // (x = k0 = t, y = n1 = hi, z = n0 = lo)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            A[base + 16 * t] = v0[DFT16_OUT(t)];
            B[base + 16 * t] = v1[DFT16_OUT(t)];
        }
    }
    __syncthreads();
    {
        v2f a[16], b[16];
        const int base = 16 * hi + 257 * lo;  // thread = (z = n0 = lo, x = k0 = hi), y = n1 = t
        unsigned rb = 8u * (unsigned)base;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            a[t] = lds_at(A, rb + 8u * (unsigned)t);
            b[t] = lds_at(B, rb + 8u * (unsigned)t);
            launder(rb);
        }
#pragma unroll
        for (int t = 1; t < 16; ++t) {
            const v2f w = tw2[16 * t + hi];
            a[t] = twmul<INV>(a[t], w);
            b[t] = twmul<INV>(b[t], w);
        }
        dft16<INV>(a);
        dft16<INV>(b);
#pragma unroll
        for (int t = 0; t < 16; ++t) {  // in place: y = k1 = t
            A[base + t] = a[DFT16_OUT(t)];
            B[base + t] = b[DFT16_OUT(t)];
        }
    }
    row_sync();
    {
        v2f a[16], b[16];
        const int base = 16 * hi + lo;  // thread = (y = k1 = lo, x = k0 = hi), z = n0 = t
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            a[t] = A[base + 257 * t];
            b[t] = B[base + 257 * t];
        }
#pragma unroll
        for (int t = 1; t < 16; ++t) {
            a[t] = twmul<INV>(a[t], tw3[t - 1]);
            b[t] = twmul<INV>(b[t], tw3[t - 1]);
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

// swizzled in -> natural out; pass-3 twiddles exp(-+2 pi i j t / 4096) are read from the global table at use (one inverse per pair)
template <bool INV>
__device__ __forceinline__ void dual_swz_to_nat(v2f (&v0)[16], v2f (&v1)[16], v2f* A, v2f* B, int j, const v2f* tw2, const v2f* tw4096) {
    const int lo = j & 15, hi = j >> 4;
    fft4096_pass1<INV>(v0, A, j);  // thread = (n1 = lo, n0 = hi): slots 17 j + k0
    fft4096_pass1<INV>(v1, B, j);
    row_sync();
    {
        v2f a[16], b[16];
        const int base = lo + 272 * hi;  // thread = (k0 = lo, n0 = hi), n1 = t: written by threads t + 16 hi of the same row
        unsigned rb = 8u * (unsigned)base;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            a[t] = lds_at(A, rb + 136u * (unsigned)t);
            b[t] = lds_at(B, rb + 136u * (unsigned)t);
            launder(rb);
        }
#pragma unroll
        for (int t = 1; t < 16; ++t) {
            const v2f w = tw2[16 * t + lo];
            a[t] = twmul<INV>(a[t], w);
            b[t] = twmul<INV>(b[t], w);
        }
        dft16<INV>(a);
        dft16<INV>(b);
#pragma unroll
        for (int t = 0; t < 16; ++t) {  // in place: k1 = t
            A[base + 17 * t] = a[DFT16_OUT(t)];
            B[base + 17 * t] = b[DFT16_OUT(t)];
        }
    }
    __syncthreads();
    {
        v2f a[16], b[16], w[16];
#pragma unroll
        for (int t = 1; t < 16; ++t) w[t] = tw4096[(unsigned)j * (unsigned)t];
#pragma unroll
        for (int t = 0; t < 16; ++t) {  // thread = k0 + 16 k1 = j, n0 = t
            a[t] = A[pad16(j + 256 * t)];
            b[t] = B[pad16(j + 256 * t)];
        }
#pragma unroll
        for (int t = 1; t < 16; ++t) {
            a[t] = twmul<INV>(a[t], w[t]);
            b[t] = twmul<INV>(b[t], w[t]);
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


// ---- single-transform forms (one column per workgroup, ONE LDS buffer: stft_reassigned_4096_col_kernel) -------------------------
template <bool INV>
__device__ __forceinline__ void nat_to_swz(v2f (&v)[16], v2f* A, int j, const v2f* tw2, const v2f* tw4096, unsigned js) {
    const int lo = j & 15, hi = j >> 4;
    dft16<INV>(v);
    {
        const int base = hi + 257 * lo;
#pragma unroll
        for (int t = 0; t < 16; ++t) A[base + 16 * t] = v[DFT16_OUT(t)];
    }
    __syncthreads();
    {
        const int base = 16 * hi + 257 * lo;
        unsigned rb = 8u * (unsigned)base;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            v[t] = lds_at(A, rb + 8u * (unsigned)t);
            launder(rb);
        }
#pragma unroll
        for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], tw2[16 * t + hi]);
        dft16<INV>(v);
        v2f o[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) o[t] = v[DFT16_OUT(t)];
#pragma unroll
        for (int t = 0; t < 16; ++t) A[base + t] = o[t];
    }
    row_sync();
    {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = A[j + 257 * t];
#pragma unroll
        for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], tw4096[js * (unsigned)t]);
        dft16<INV>(v);
        v2f o[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) o[t] = v[DFT16_OUT(t)];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = o[t];
    }
}
template <bool INV>
__device__ __forceinline__ void swz_to_nat(v2f (&v)[16], v2f* A, int j, const v2f* tw2, const v2f* tw4096) {
    const int lo = j & 15, hi = j >> 4;
    fft4096_pass1<INV>(v, A, j);
    row_sync();
    {
        const int base = lo + 272 * hi;
        unsigned rb = 8u * (unsigned)base;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            v[t] = lds_at(A, rb + 136u * (unsigned)t);
            launder(rb);
        }
#pragma unroll
        for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], tw2[16 * t + lo]);
        dft16<INV>(v);
        v2f o[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) o[t] = v[DFT16_OUT(t)];
#pragma unroll
        for (int t = 0; t < 16; ++t) A[base + 17 * t] = o[t];
    }
    __syncthreads();
    {
        v2f w[16];
#pragma unroll
        for (int t = 1; t < 16; ++t) w[t] = tw4096[(unsigned)j * (unsigned)t];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = A[pad16(j + 256 * t)];
#pragma unroll
        for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], w[t]);
        dft16<INV>(v);
        v2f o[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) o[t] = v[DFT16_OUT(t)];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = o[t];
    }
}

}  // namespace

#ifndef OMX_COL_WGS
#define OMX_COL_WGS 3
#endif
// One column per workgroup, one LDS buffer (37 KiB): three or four workgroups = 12 or 16 wavefronts per CU.  The pair kernel
// below runs two wavefronts per SIMD and its VALU pipe is busy 56 % of the time; with ONE workgroup per CU it takes 3.02 ms
// against 1.76 with two (measured), which is what independent wavefronts that can each issue 32 % of the time predict
// (1 - 0.68^2 = 54 %) — so the lever is wavefronts per SIMD, not barriers (25 -> 13 per pair changed nothing).
__global__ __launch_bounds__(256, OMX_COL_WGS) void stft_reassigned_4096_col_kernel(StftFastArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // LDS: 53 136 B, so that three workgroups fit a CU.  The wave counts of the compaction live in the transform buffer's tail
    // (slots the natural -> swizzled layout never touches; the inverse, which does, is long done by then).
    constexpr int COL_LDS = 4352;  // slots: the inverse's padded layout ends at 4350, the 16 x 257 layout at 4112
    v2f* A = reinterpret_cast<v2f*>(smem_raw);
    v2f* tw2_lds = A + COL_LDS - 16;                               // pass-2 twiddles exp(-2 pi i a t / 256) at [t][a], t >= 1 (row 0 is never read)
    v2f* stash = tw2_lds + 256;                                    // [2048] FFT(t w s) of this thread's own bins, parked across the last transform
    float* hil = reinterpret_cast<float*>(stash + 2048);           // X[0]/2, X[4096]/2
    uint32_t* scan = reinterpret_cast<uint32_t*>(A + 4200);        // [9][4] wave counts

    // XCD-aware map (same as block_to_stream_column): block b runs on XCD b % 8; stream s is pinned to XCD s % 8
    const uint32_t blk = blockIdx.x, xcd = blk & 7u, bq = blk >> 3;
    const uint32_t s = (bq / a.n_cols) * 8u + xcd, col = bq % a.n_cols;
    if (s >= a.n_streams) return;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const int lane = j & 63, wave = j >> 6;
    const unsigned js = swz(ju);
    const float* ring = a.ring + (uint64_t)s * a.cap;
    const uint32_t mask32 = (uint32_t)(a.cap - 1);
    const uint32_t bytemask = mask32 << 2;
    const char* ring_bytes = reinterpret_cast<const char*>(ring);
    const PairConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};
    if (col >= stft_cols(a, s)) return;  // ragged banks: this stream's own column count
    const uint64_t p0 = stft_tail(a, s) + (uint64_t)col * a.hop;
    uint32_t* count_out = a.counts + (uint64_t)s * a.n_cols + col;
    if (a.last_nonzero[s] < (long long)p0) {  // silent fast path (:307-316)
        if (j == 0) *count_out = 0;
        return;
    }
    if (j >= 16) tw2_lds[j] = a.tw256[(ju & 15u) * (ju >> 4)];

    // ---- 1. packed real FFT of the 8192-sample window ------------------------------------------------------------------------
    const uint32_t p32 = (uint32_t)p0;
    v2f v[16];
    if ((p0 & 1ull) == 0) {
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
    nat_to_swz<false>(v, A, j, tw2_lds, a.tw4096, js);  // v[t] = Zf[js + 256 t]

    // ---- 2. Hilbert transform with one half-length inverse ---------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 16; ++t) A[j + 257 * t] = v[t];
    if (j == 0) {
        hil[0] = (v[0].x + v[0].y) * 0.5f;  // X[0] / 2
        hil[1] = (v[0].x - v[0].y) * 0.5f;  // X[4096] / 2
    }
    __syncthreads();
    {
        const unsigned pm = 256u - js;
        const int part = (int)(16u * (pm & 15u) + ((pm >> 4) & 15u)) + (j == 0 ? 257 : 0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const v2f w8 = a.tw8192[js + 256u * (unsigned)t];
            const v2f z = v[t], zr = A[part + 257 * (15 - t)];
            const v2f sum{z.x + zr.x, z.y - zr.y}, dif{z.x - zr.x, z.y + zr.y};
            v[t] = cmulc(sum, w8) - cmul(dif, w8);
            if (t == 0 && j == 0) v[t] = v2f{0.0f, 0.0f};
        }
    }
    const float half_x0 = hil[0], half_xn = hil[1];
    __syncthreads();  // partners are read from the buffer the inverse is about to overwrite
    swz_to_nat<true>(v, A, j, tw2_lds, a.tw4096);  // v[t] = (Im a[2m], Im a[2m+1]), m = j + 256 t

    // ---- 3. analytic slice s[i] = analytic[2048 + i], i = j + 256 t -----------------------------------------------------------
    __syncthreads();  // pass 3 of the inverse reads all over the buffer
    float* imag = reinterpret_cast<float*>(A);
#pragma unroll
    for (int t = 4; t < 12; ++t) *reinterpret_cast<v2f*>(imag + 2 * (j + 256 * t - 1024)) = v[t];
    v2f sv[16];
    {
        const uint32_t qx = p32 + 2048u + ju;
#pragma unroll
        for (int t = 0; t < 16; ++t) sv[t].x = *reinterpret_cast<const float*>(ring_bytes + (((qx + 256u * (unsigned)t) << 2) & bytemask));
    }
    __syncthreads();
    {
        const float par = (j & 1) ? -half_xn : half_xn;
#pragma unroll
        for (int t = 0; t < 16; ++t) sv[t] = v2f{4096.0f * sv[t].x - half_x0 + par, imag[j + 256 * t]};
    }
    const float c0 = a.win_c0, half_c1 = 0.5f * a.win_c1, dscale = a.win_c1 * (3.14159265358979323846f / 4096.0f);
    struct NatRead {
        const v2f* lo;
        int q;
    };
    auto nat_read = [&](int d) {
        const int m = j + d;
        const int x = m & 15, y = (m >> 4) & 15, carry = m >> 8;
        const int q = ((x & 14) + carry) & 15;
        return NatRead{A + 16 * x + y + 257 * q, q};
    };
    const NatRead rd_m = nat_read(-1), rd_c = nat_read(0), rd_p = nat_read(1);
    const int wq = (j >> 4) & 14;
    auto natural_copy = [&](const v2f (&z)[16]) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t > 8 && t != 15) continue;
            v2f* w = (wq >= 16 - t ? A + j - 4112 : A + j) + 257 * (t + wq);
            w[0] = z[t];
        }
    };
    __syncthreads();  // the gather above still reads the buffer

    // ---- 4. Z2 = FFT((n - 2047.5) s) first: t = FFT(t w s) from its bins (18 registers to carry instead of 36) ---------------------
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float nc = (float)(j + 256 * t) - 2047.5f;  // compute_time_weighted's ramp (:601-608)
        v[t] = v2f{sv[t].x * nc, sv[t].y * nc};
    }
    nat_to_swz<false>(v, A, j, tw2_lds, a.tw4096, js);
    natural_copy(v);
    __syncthreads();
    v2f bt8{0.0f, 0.0f};  // bin 2048 (thread 0)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t == 8 && j != 0) break;
        const v2f* pm = (rd_m.q >= 16 - t ? rd_m.lo - 4112 : rd_m.lo) + 257 * t;
        const v2f* pc = (rd_c.q >= 16 - t ? rd_c.lo - 4112 : rd_c.lo) + 257 * t;
        const v2f* pp = (rd_p.q >= 16 - t ? rd_p.lo - 4112 : rd_p.lo) + 257 * t;
        const v2f z2c = pc[0], z2m = pm[0], z2p = pp[0];
        const v2f z2s{z2m.x + z2p.x, z2m.y + z2p.y};
        const v2f btv{c0 * z2c.x + half_c1 * z2s.x, c0 * z2c.y + half_c1 * z2s.y};
        if (t < 8) stash[j + 256 * t] = btv;  // read back by this thread only
        else bt8 = btv;
    }
    __syncthreads();  // the neighbour reads are done before the next transform's pass 1 writes

    // ---- 5. Z = FFT(s): b = FFT(w s), d = FFT(w' s) from its bins; reassignment + ordered compaction ------------------------------
    nat_to_swz<false>(sv, A, j, tw2_lds, a.tw4096, js);
    natural_copy(sv);
    __syncthreads();
    omx_spectrogram_point pts[9];
    unsigned long long masks[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint32_t bin = (uint32_t)(j + 256 * t);
        bool keep = false;
        if (t < 8 || j == 0) {
            const v2f* pm = (rd_m.q >= 16 - t ? rd_m.lo - 4112 : rd_m.lo) + 257 * t;
            const v2f* pc = (rd_c.q >= 16 - t ? rd_c.lo - 4112 : rd_c.lo) + 257 * t;
            const v2f* pp = (rd_p.q >= 16 - t ? rd_p.lo - 4112 : rd_p.lo) + 257 * t;
            const v2f zc = pc[0], zm = pm[0], zp = pp[0];
            const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y};
            const v2f bb{c0 * zc.x + half_c1 * zs.x, c0 * zc.y + half_c1 * zs.y};
            const v2f bd{-dscale * zd.y, dscale * zd.x};  // i c1 (pi / W) (Z[k-1] - Z[k+1])
            keep = reassign_one(bin, bb, bd, t < 8 ? stash[bin] : bt8, a.bin_norm[bin], rc, pts[t]);
        }
        masks[t] = __ballot(keep);
        if (lane == 0) scan[t * 4 + wave] = (uint32_t)__popcll(masks[t]);
    }
    __syncthreads();
    omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
    uint32_t running = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint4 c4 = *reinterpret_cast<const uint4*>(scan + t * 4);
        const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
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
}

void launch_stft_reassigned_4096_col(const StftFastArgs& a, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return;
    const size_t lds = (size_t)(4352 - 16 + 256 + 2048) * sizeof(v2f) + 4 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_4096_col_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(stft_reassigned_4096_col_kernel, dim3(stream_column_grid(a.n_streams, a.n_cols)), dim3(256), lds, stream, a);
}

__global__ __launch_bounds__(256, 2) void stft_reassigned_4096_swz_pair_kernel(StftFastArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* A = reinterpret_cast<v2f*>(smem_raw);
    v2f* B = A + SWZ_LDS;
    v2f* tw2_lds = B + SWZ_LDS;                                    // [16][16] pass-2 twiddles exp(-2 pi i a t / 256) at [t][a]
    uint32_t* scan = reinterpret_cast<uint32_t*>(tw2_lds + 256);   // [9][4] wave counts
    float* hil = reinterpret_cast<float*>(scan + 36);              // X[0]/2, X[4096]/2 of both columns

    const uint32_t chunks = (a.n_cols + 1u) / 2u;
    uint32_t s, chunk;
    if (!pair_block_to_stream_chunk(a.n_streams, chunks, s, chunk)) return;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const int lane = j & 63, wave = j >> 6;
    const unsigned js = swz(ju);  // this thread's OUTPUT index of a natural -> swizzled transform: X[js + 256 t]
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

    v2f tw3[15];  // pass-3 twiddles of the natural -> swizzled transforms (three of the four duals), resident
#pragma unroll
    for (int t = 1; t < 16; ++t) tw3[t - 1] = a.tw4096[js * (unsigned)t];
    tw2_lds[j] = a.tw256[(ju & 15u) * (ju >> 4)];  // [t][a] = exp(-2 pi i a t / 256); first read behind the first transform's pass-1 barrier

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
    dual_nat_to_swz<false>(va, vb, A, B, j, tw2_lds, tw3);  // v[t] = Zf[js + 256 t]

    // ---- 2. Hilbert transform with ONE half-length inverse per column (derivation: stft_kernels.hip step 2) -----------
    // natural-order copy of both spectra, written to the slots this thread read in pass 3: bin js + 256 t -> slot j + 257 t
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        A[j + 257 * t] = va[t];
        B[j + 257 * t] = vb[t];
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
        v2f w8[16];  // exp(-2 pi i k / 8192) / 2, k = js + 256 t: one table read serves both columns
#pragma unroll
        for (int t = 0; t < 16; ++t) w8[t] = a.tw8192[js + 256u * (unsigned)t];
        // partner Zf[(4096 - k) & 4095] of k = js + 256 t: 4096 - k = (256 - js) + 256 (15 - t), so its slot is part + 257 (15 - t)
        // (thread 0: 256 (16 - t); its t = 0 read lands one slot past the 16 x 257 block, inside the buffer, and is not used)
        const unsigned pm = 256u - js;  // 1 ... 256
        const int part = (int)(16u * (pm & 15u) + ((pm >> 4) & 15u)) + (j == 0 ? 257 : 0);
        auto hilbert_spectrum = [&](v2f (&y)[16], const v2f (&v)[16], const v2f* X) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const v2f z = v[t], zr = X[part + 257 * (15 - t)];
                const v2f sum{z.x + zr.x, z.y - zr.y}, dif{z.x - zr.x, z.y + zr.y};
                y[t] = cmulc(sum, w8[t]) - cmul(dif, w8[t]);
                if (t == 0 && j == 0) y[t] = v2f{0.0f, 0.0f};
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
    dual_swz_to_nat<true>(ya, yb, A, B, j, tw2_lds, a.tw4096);  // y[t] = (Im a[2m], Im a[2m+1]), m = j + 256 t

    // ---- 3. gather s[i] = analytic[2048 + i], i = j + 256 t, for both columns -------------------------------------------
    __syncthreads();  // pass 3 of the inverse reads all over A and B
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
    // Natural-order copy of the windowed spectra.  Bin k = x + 16 y + 256 t lives in the slot set of thread (y, x), at
    // 16 x + y + 257 zeta with zeta = (t + (x & 14)) & 15: rotating the sixteen slots of a thread by its x digit makes the reads
    // below — lanes over consecutive bins, i.e. over x — conflict-free for the centre bin and two-way for the two neighbours
    // (slot index mod 32 = 16 (x & 1) + y + zeta; without the rotation the sixteen x of a half-wave share two values).
    // A read is `lo + 257 t` while t + q < 16 and `lo - 4112 + 257 t` beyond, q = (x & 14) + carry into t.
    struct NatRead {
        const v2f* lo;  // A + 16 x + y + 257 q
        int q;
    };
    auto nat_read = [&](int d) {
        const int m = j + d;                 // -1 ... 256
        const int x = m & 15, y = (m >> 4) & 15, carry = m >> 8;  // arithmetic shift: -1 for m = -1 (bin 256 t - 1), 1 for m = 256
        const int q = ((x & 14) + carry) & 15;
        return NatRead{A + 16 * x + y + 257 * q, q};
    };
    const NatRead rd_m = nat_read(-1), rd_c = nat_read(0), rd_p = nat_read(1);
    const int wq = (j >> 4) & 14;  // writer side: this thread's own rotation
    __syncthreads();  // the gather above still reads A and B

    // ---- 4. per column: Z = FFT(s), Z2 = FFT((n - 2047.5) s) as one dual transform; windows applied on the bins --------
    auto column = [&](const v2f (&sv)[16], bool silent, uint32_t col, uint32_t* count_out) {
        v2f z[16], z2[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float nc = (float)(j + 256 * t) - 2047.5f;  // compute_time_weighted's ramp (:601-608)
            z[t] = sv[t];
            z2[t] = v2f{sv[t].x * nc, sv[t].y * nc};
        }
        dual_nat_to_swz<false>(z, z2, A, B, j, tw2_lds, tw3);  // z[t] = Z[js + 256 t]
        // natural-order copy, written to the slots this thread just read: bins 0 ... 2303 (t <= 8) and 3840 ... 4095 (bin -1)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t > 8 && t != 15) continue;
            v2f* w = (wq >= 16 - t ? A + j - 4112 : A + j) + 257 * (t + wq);
            w[0] = z[t];
            w[SWZ_LDS] = z2[t];
        }
        float pn[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) pn[t] = a.bin_norm[(t < 8 || j == 0) ? ju + 256u * (unsigned)t : 0u];
        __syncthreads();

        omx_spectrogram_point pts[9];
        unsigned long long masks[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint32_t bin = (uint32_t)(j + 256 * t);
            bool keep = false;
            if ((t < 8 || j == 0) && !silent) {
                const v2f* pm = (rd_m.q >= 16 - t ? rd_m.lo - 4112 : rd_m.lo) + 257 * t;
                const v2f* pc = (rd_c.q >= 16 - t ? rd_c.lo - 4112 : rd_c.lo) + 257 * t;
                const v2f* pp = (rd_p.q >= 16 - t ? rd_p.lo - 4112 : rd_p.lo) + 257 * t;
                const v2f zc = pc[0], zm = pm[0], zp = pp[0];
                const v2f z2c = pc[SWZ_LDS], z2m = pm[SWZ_LDS], z2p = pp[SWZ_LDS];
                const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y}, z2s{z2m.x + z2p.x, z2m.y + z2p.y};
                const v2f bb{c0 * zc.x + half_c1 * zs.x, c0 * zc.y + half_c1 * zs.y};
                const v2f bd{-dscale * zd.y, dscale * zd.x};  // i c1 (pi / W) (Z[k-1] - Z[k+1])
                const v2f bt{c0 * z2c.x + half_c1 * z2s.x, c0 * z2c.y + half_c1 * z2s.y};
                keep = reassign_one(bin, bb, bd, bt, pn[t], rc, pts[t]);
            }
            masks[t] = __ballot(keep);
            if (lane == 0) scan[t * 4 + wave] = (uint32_t)__popcll(masks[t]);
        }
        __syncthreads();  // wave counts; also: every neighbour read of A and B is done (the next column's pass 1 may write)
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

void launch_stft_reassigned_4096_swz_pair(const StftFastArgs& a, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return;
    const size_t lds = (size_t)(2 * SWZ_LDS + 256) * sizeof(v2f) + 9 * 4 * sizeof(uint32_t) + 4 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_4096_swz_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    const uint32_t chunks = (a.n_cols + 1u) / 2u;
    hipLaunchKernelGGL(stft_reassigned_4096_swz_pair_kernel, dim3(stream_column_grid(a.n_streams, chunks)), dim3(256), lds, stream, a);
}

}  // namespace omx
#endif  // OMX_TUNING
