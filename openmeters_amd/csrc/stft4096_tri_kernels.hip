// K2 (round 4 form): the pair kernel's arithmetic at THREE workgroups per CU (reference spectrogram/processor.rs:318-348,
// :439-488, :546-567).  Same four 4096-point transforms per column as stft4096_pair_kernels.hip; what changes is what a
// workgroup holds while it runs them:
//   * ONE padded 4096-complex LDS buffer, time-shared by the two chains of every dual transform.  The chains are staggered by
//     half a pass: while chain a's values cross the buffer, chain b's butterflies run in registers, and the other way round — every
//     barrier-to-barrier segment carries the LDS traffic of one chain and the arithmetic of the other.
//   * <= 168 VGPR: nothing is prefetched across a transform (bin normalisation, the real-part samples and the time-weighted
//     window are requested in the phase that consumes them — three wavefronts per SIMD cover that latency), column b's analytic
//     slice waits in LDS (its imaginary half; the real half is re-read from the ring) instead of in 32 registers, and
//     FFT(t w s) is windowed in the time domain (the reference's own twindow table, processor.rs:601-608) so that only ONE
//     spectrum needs its neighbour bins (18 KiB of natural-order bins instead of 37).
// LDS: 34 816 (buffer) + 16 384 (column b's imaginary half) + 2 048 (pass-2 twiddles) + 160 = 53 408 B -> three workgroups
// (twelve wavefronts) per CU, where the pair kernel's 70 KiB / 254 VGPR allow two.
#include "stft_kernels.hpp"

#include <mutex>

#include "buffer_device.hpp"
#include "fft_device.hpp"
#include "fft_fused_device.hpp"
#include "reassign_device.hpp"
#include "twiddle_run_device.hpp"

namespace omx {

namespace {

// TRI_KNOCK (pricing builds of the TUNING library only — `make TUNING=1 EXTRA=-DTRI_KNOCK=n` — WRONG columns): 1 no LDS traffic in the transforms, 2 no butterflies, 3 no barriers in the transforms, 4 no point stores, 5 no Hilbert-spectrum arithmetic, 6 no reassignment arithmetic, 7 no window / table loads, 8 no Hilbert LDS round trip
#if !defined(TRI_KNOCK) || !defined(OMX_TUNING)  // (product objects cannot be built to emit wrong columns: the knock-outs exist under OMX_TUNING only)
#undef TRI_KNOCK
#define TRI_KNOCK 0
#endif
#ifndef TRI_LDS_SYNC  // 1: the kernel's barriers wait for LDS traffic only (the workgroup exchanges nothing through global memory): __syncthreads()
#define TRI_LDS_SYNC 1  // also drains vmcnt — column a's point stores — at column b's first barrier.  Same-box A/B: 1.337 -> 1.329 ms (K2-ldssync)
#endif
#if TRI_KNOCK == 3
#define TRI_SYNC() __builtin_amdgcn_sched_barrier(0)
#elif TRI_LDS_SYNC
#define TRI_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define TRI_SYNC() __syncthreads()
#endif

__device__ __forceinline__ bool tri_block_to_stream_chunk(uint32_t n_streams, uint32_t chunks, uint32_t& s, uint32_t& chunk) {
    const uint32_t b = blockIdx.x;  // block b runs on XCD b % 8; stream s is pinned to XCD s % 8 (same map as the pair kernel)
    const uint32_t xcd = b & 7u, q = b >> 3;
    s = (q / chunks) * 8u + xcd;
    chunk = q % chunks;
    return s < n_streams;
}

template <int TT>
__device__ __forceinline__ v2f w8_at(v2f base) {  // exp(-2 pi i (j + 256 TT) / 8192) / 2 from thread j's table value
    return rotate128<4 * TT>(base);
}

// Buffer layout.  TRI_SWZ = 1 (tuning build only; it LOST its A/B on the extra v_xor per address): slot(n) = n ^ ((n >> 4) & 15), no padding — every access pattern of the three passes is
// conflict-free under the gfx950 rules for the instructions hipcc picks here (single ds_read_b64: 32-lane groups over 64 banks;
// ds_write_b64: 16-lane groups over 32 banks), and every address is one base register, an immediate offset and at most one v_xor.
// The +1/16 padding of fft_device.hpp is conflict-free for 16-lane groups only: a 32-lane read spans 33 slots and pays a second cycle.
#ifndef TRI_SWZ
#define TRI_SWZ 0
#endif
#if TRI_SWZ
constexpr int kTriSlots = 4096;
__device__ __forceinline__ int xslot(int j) { return j ^ ((j >> 4) & 15); }  // slot of element j + 256 t = xslot(j) + 256 t
// the 16 xor-ed write addresses are formed per call (one v_xor each): left to itself hipcc keeps all 32 of them in registers for the
// whole kernel, which is what spills at 168
__device__ __forceinline__ int per_call(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ void x_write1(const v2f (&v)[16], v2f* X, int j) {  // pass-1 outputs: y[16 j + t] -> slot 16 j + (t ^ (j & 15))
    const int base = per_call(16 * j + (j & 15));
#pragma unroll
    for (int t = 0; t < 16; ++t) X[base ^ t] = v[DFT16_OUT(t)];
}
__device__ __forceinline__ void x_write2(const v2f (&v)[16], v2f* X, int j) {  // pass-2 outputs: z[(j / 16) 256 + j % 16 + 16 t] -> ... + 16 t + ((j % 16) ^ t)
    const int base = per_call((j >> 4) * 256 + (j & 15));
#pragma unroll
    for (int t = 0; t < 16; ++t) X[(base ^ t) + 16 * t] = v[DFT16_OUT(t)];
}
#elif defined(TRI_PAD16) && TRI_PAD16
// the round-4 layout (A/B builds only: tools/build_ab.sh pad16 stft4096_tri_kernels.hip -DTRI_PAD16=1)
constexpr int kTriSlots = FFT4096_LDS;
constexpr int kTriStep = 272;
__device__ __forceinline__ int xslot(int j) { return pad16(j); }  // pad16 is linear across multiples of 256: + 272 t
__device__ __forceinline__ void x_write1(const v2f (&v)[16], v2f* X, int j) {  // pass-1 outputs: y[16 j + t]
    const int base = 17 * j;
#pragma unroll
    for (int t = 0; t < 16; ++t) X[base + t] = v[DFT16_OUT(t)];
}
__device__ __forceinline__ void x_write2(const v2f (&v)[16], v2f* X, int j) {  // pass-2 outputs: z[(j / 16) 256 + j % 16 + 16 t]
    const int base = (j >> 4) * 272 + (j & 15);
#pragma unroll
    for (int t = 0; t < 16; ++t) X[base + 17 * t] = v[DFT16_OUT(t)];
}
__device__ __forceinline__ void x_read2(v2f (&v)[16], const v2f* X, int j) {
    const int base = xslot(j);
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = X[base + kTriStep * t];
}
#else
// Round 5 layout: pass-1 outputs are stored TRANSPOSED — output k0 of thread j1 at element k0 * 256 + j1 — and the only padding is ONE slot
// per 256 elements (slot(e) = e + (e >> 8)).  Banking is per instruction (MI355X_MICROARCH.md, LDS), and hipcc picks the instruction:
//   x_write1   ds_write_b64 (thread stride 257 slots: not mergeable), 16-lane groups over 32 banks: 16 consecutive lanes on 16 consecutive slots
//   x_read2    thread stride 16 slots = 128 B, which hipcc ALWAYS merges into ds_read2_b64 — per access 16-lane groups over 32 banks:
//              lanes k = 0 ... 15 of a group sit at slots 257 k + const = k (mod 16): conflict-free BECAUSE the pad is odd
//   x_write2   ds_write2_b64, 16 consecutive lanes on 16 consecutive slots
//   x_read     (pass 3, natural order) ds_read_b64 (stride 257: not mergeable), 32-lane groups over 64 banks: 32 consecutive slots
// A/B on one box (tools/ab_bench.sh, tools/debug/ab_lds_pmc.sh): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 21.8 % -> 3.3 %,
// LDS-active cycles -10 %, kernel 1.339 -> 1.318 ms.  The round-4 layout (y[16 j + t], one pad slot per 16) was conflict-free for the
// 16-lane stores only: a 32-lane ds_read_b64 spanned 33 slots and paid a second cycle.  TRI_STEP = 258 (pad 2: conflict-free for plain
// 32-lane ds_read_b64 in x_read2) LOST — 1.358 ms, conflicts 18.7 % — because the merged ds_read2_b64 is what actually runs there.
#ifndef TRI_STEP
#define TRI_STEP 257
#endif
constexpr int kTriStep = TRI_STEP;               // slots between elements j + 256 t and j + 256 (t + 1)
constexpr int kTriSlots = 16 * kTriStep;    // 4112
__device__ __forceinline__ int xslot(int j) { return j; }  // slot of element j + 256 t = j + 258 t
__device__ __forceinline__ void x_write1(const v2f (&v)[16], v2f* X, int j) {  // pass-1 output t of thread j -> element 256 t + j
    if (TRI_KNOCK == 1) return;
#pragma unroll
    for (int t = 0; t < 16; ++t) X[j + kTriStep * t] = v[DFT16_OUT(t)];
}
__device__ __forceinline__ void x_read2(v2f (&v)[16], const v2f* X, int j) {  // pass-2 inputs: output j % 16 of the threads j / 16 + 16 t
    const int base = kTriStep * (j & 15) + (j >> 4);
    if (TRI_KNOCK == 1) return;
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = X[base + 16 * t];
}
__device__ __forceinline__ void x_write2(const v2f (&v)[16], v2f* X, int j) {  // pass-2 outputs: z[(j / 16) 256 + j % 16 + 16 t]
    const int base = (j >> 4) * kTriStep + (j & 15);
    if (TRI_KNOCK == 1) return;
#pragma unroll
    for (int t = 0; t < 16; ++t) X[base + 16 * t] = v[DFT16_OUT(t)];
}
#endif
#if TRI_SWZ
constexpr int kTriStep = 256;
__device__ __forceinline__ void x_read2(v2f (&v)[16], const v2f* X, int j) {
    const int base = xslot(j);
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = X[base + kTriStep * t];
}
#endif
__device__ __forceinline__ void x_read(v2f (&v)[16], const v2f* X, int j) {  // inputs of pass 3 / natural order: element j + 256 t
    const int base = xslot(j);
    if (TRI_KNOCK == 1) return;
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = X[base + kTriStep * t];
}
// TRI_FUSED (default): the twiddle products ride the butterflies as fused multiply-adds (fft_fused_device.hpp: 269 packed
// operations per 4096-point transform instead of 300)
#ifndef TRI_FUSED
#define TRI_FUSED 1
#endif
template <bool INV>
__device__ __forceinline__ void plain16(v2f (&v)[16]) {
    if (TRI_KNOCK == 2) return;
#if TRI_FUSED
    dft16_fused<INV>(v);
#else
    dft16<INV>(v);
#endif
}
template <bool INV>
__device__ __forceinline__ void pass2_16(v2f (&v)[16], const v2f* tw2, int j) {  // exp(-+2 pi i (j % 16) t / 256) v[t], then the DFT
    const unsigned k = (unsigned)j & 15u;
    if (TRI_KNOCK == 2) return;
#if TRI_FUSED
    v2f w[16];
#pragma unroll
    for (int t = 1; t < 16; ++t) w[t] = tw2[k * (unsigned)t];
    w[0] = w[1];
    dft16_fused_tw<INV>(v, w);
#else
#pragma unroll
    for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], tw2[k * (unsigned)t]);
    dft16<INV>(v);
#endif
}
template <bool INV>
__device__ __forceinline__ void pass3_16(v2f (&v)[16], const v2f (&tw3)[15]) {  // exp(-+2 pi i j t / 4096) v[t], then the DFT
    if (TRI_KNOCK == 2) return;
#if TRI_FUSED
    v2f w[16];
#pragma unroll
    for (int t = 1; t < 16; ++t) w[t] = tw3[t - 1];
    w[0] = w[1];
    dft16_fused_tw<INV>(v, w);
#else
#pragma unroll
    for (int t = 1; t < 16; ++t) v[t] = twmul<INV>(v[t], tw3[t - 1]);
    dft16<INV>(v);
#endif
}
__device__ __forceinline__ void natural(v2f (&v)[16]) {  // X[k] sits in v[DFT16_OUT(k)] after dft16: rename (constant indices, no moves)
    v2f r[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) r[t] = v[DFT16_OUT(t)];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = r[t];
}

// Two 4096-point transforms through ONE buffer: x[j + 256 t] in a[t] / b[t] on entry, X[j + 256 t] on return.  The caller has a
// barrier between its last use of X and this call; on return other wavefronts may still be reading X (chain b's pass-3 inputs).
template <bool INV>
__device__ __forceinline__ void tri_dual(v2f (&a)[16], v2f (&b)[16], v2f* X, const v2f* tw2, const v2f (&tw3)[15], int j) {
    plain16<INV>(a);
    x_write1(a, X, j);
    TRI_SYNC();
    x_read2(a, X, j);
    plain16<INV>(b);
    TRI_SYNC();  // every pass-2 input of chain a is in registers
    x_write1(b, X, j);
    pass2_16<INV>(a, tw2, j);
    TRI_SYNC();
    x_read2(b, X, j);
    TRI_SYNC();
    x_write2(a, X, j);
    pass2_16<INV>(b, tw2, j);
    TRI_SYNC();
    x_read(a, X, j);
    TRI_SYNC();
    x_write2(b, X, j);
    pass3_16<INV>(a, tw3);
    natural(a);
    TRI_SYNC();
    x_read(b, X, j);
    pass3_16<INV>(b, tw3);
    natural(b);
}

// one 12-byte point at slot `pos` of a column (a wavefront's kept points are contiguous: 768 B per store instruction).  Non-temporal
// (`nt`) and write-through (`sc0 sc1`) stores measured 1.304 / 1.348 ms against 1.305 for the plain store: not used.
__device__ __forceinline__ void store_point(omx_spectrogram_point* out, uint32_t pos, const omx_spectrogram_point& p) {
    *reinterpret_cast<omx_spectrogram_point*>(reinterpret_cast<char*>(out) + __umul24(pos, 12u)) = p;
}

}  // namespace

constexpr size_t kTriLds = (size_t)kTriSlots * sizeof(v2f) + 4096 * sizeof(float) + 256 * sizeof(v2f) + 36 * sizeof(uint32_t) + 4 * sizeof(float);

// TERMS = cosine terms of the window (1 rectangular, 2 Hann / Hamming, 3 Blackman, 4 Blackman-Harris): with Z = FFT(s)
//   FFT(w s)[k]  = c0 Z[k] + sum_m c_m / 2 (Z[k-m] + Z[k+m]),      FFT(w' s)[k] = i sum_m c_m (pi m / W) (Z[k-m] - Z[k+m])
// (w' = the spectral derivative of w, processor.rs:569-599 = - sum_m c_m (2 pi m / W) sin(2 pi m n / W) exactly), m = 1 ... TERMS - 1
template <int TERMS>
__global__ __launch_bounds__(256, 3) void stft_reassigned_4096_tri_kernel(StftFastArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* X = reinterpret_cast<v2f*>(smem_raw);
    float* imb = reinterpret_cast<float*>(X + kTriSlots);        // [4096] Im analytic[2048 + i] of column b
    v2f* tw2_lds = reinterpret_cast<v2f*>(imb + 4096);             // [256] exp(-2 pi i k / 256)
    uint32_t* scan = reinterpret_cast<uint32_t*>(tw2_lds + 256);   // [9][4] wave counts
    float* hil = reinterpret_cast<float*>(scan + 36);              // X[0]/2, X[4096]/2 of both columns

    const uint32_t chunks = (a.n_cols + 1u) / 2u;
    uint32_t s, chunk;
    if (!tri_block_to_stream_chunk(a.n_streams, chunks, s, chunk)) return;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const int lane = j & 63, wave = j >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const float* ring = a.ring + (uint64_t)s * a.cap;
    const uint32_t mask32 = (uint32_t)(a.cap - 1);
    const uint32_t bytemask = mask32 << 2;
    const char* ring_bytes = reinterpret_cast<const char*>(ring);
    const long long last_nonzero = a.last_nonzero[s];
    const uint32_t* cols_p = a.cols;
    const uint64_t* tails_p = a.tails;
    const uint32_t n_cols_s = cols_p ? cols_p[s] : a.n_cols;
    const uint64_t tail_s = tails_p ? tails_p[s] : a.tail;
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};

    const uint32_t col0 = chunk * 2u;
    if (col0 >= n_cols_s) return;
    const bool have1 = col0 + 1u < n_cols_s;
    const uint32_t col1 = have1 ? col0 + 1u : col0;  // an odd tail computes column 0 twice and stores it once
    const uint64_t p0a = tail_s + (uint64_t)col0 * a.hop, p0b = tail_s + (uint64_t)col1 * a.hop;
    uint32_t* count_a = a.counts + (uint64_t)s * a.n_cols + col0;
    uint32_t* count_b = a.counts + (uint64_t)s * a.n_cols + col1;

    const uint32_t pa32 = (uint32_t)p0a, pb32 = (uint32_t)p0b;
    const uint32_t off_a = pa32 & mask32, hop_bytes = (pb32 - pa32) * 4u;
    const bool direct = (uint64_t)off_a + (uint64_t)(pb32 - pa32) + 8192ull <= a.cap && ((p0a | p0b) & 1ull) == 0;
    const GlobalBuffer windowb = global_buffer(ring + off_a, hop_bytes + 8192u * 4u);
    const GlobalBuffer tw4096b = global_buffer(a.tw4096, 4096u * 8u), tw8192b = global_buffer(a.tw8192, 4096u * 8u),
                       normb = global_buffer(a.bin_norm, 2049u * 4u), twinb = global_buffer(a.twindow, 4096u * 4u);

    // ---- 1. packed real FFTs of the two 8192-sample windows ---------------------------------------------------------------
    v2f va[16], vb[16];
    if (TRI_KNOCK == 7) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            va[t] = v2f{(float)(j + t) * 1e-4f, (float)(j - t) * 1e-4f};
            vb[t] = v2f{(float)(j + 2 * t) * 1e-4f, (float)(j - 3 * t) * 1e-4f};
        }
    } else if (direct) {
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
    v2f tw3[15];
#pragma unroll
    for (int t = 1; t < 16; ++t) tw3[t - 1] = load_v2f(tw4096b, ju * 8u * (unsigned)t, 0);
    tw2_lds[j] = a.tw256[ju];
    const v2f w8_base = load_v2f(tw8192b, ju * 8u, 0);
    const bool silent_a = last_nonzero < (long long)p0a, silent_b = last_nonzero < (long long)p0b;
    if (silent_a && (silent_b || !have1)) {  // silent fast path (:307-316)
        if (j == 0) {
            *count_a = 0;
            if (have1) *count_b = 0;
        }
        return;
    }
    tri_dual<false>(va, vb, X, tw2_lds, tw3, j);  // v[t] = Zf[j + 256 t]

    // ---- 2. Hilbert transform with ONE half-length inverse per column, one column at a time through X ----------------------
    // partner Zf[(4096 - k) & 4095] of k = j + 256 t: element (256 - j) + 256 (15 - t); thread 0's partners 4096 - 256 t sit one block
    // higher, and its t = 0 read (one slot past the buffer, inside the allocation) is not used
    const int part = j ? xslot(256 - j) : kTriStep;
    auto hilbert_spectrum = [&](v2f (&y)[16], const v2f (&v)[16]) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const v2f z = v[t], zr = TRI_KNOCK == 8 ? v[15 - t] : X[part + kTriStep * (15 - t)];
            const v2f sum{z.x + zr.x, z.y - zr.y}, dif{z.x - zr.x, z.y + zr.y};
            v2f w8;
            switch (t) {  // compile-time after unrolling
                case 0: w8 = w8_at<0>(w8_base); break;
                case 1: w8 = w8_at<1>(w8_base); break;
                case 2: w8 = w8_at<2>(w8_base); break;
                case 3: w8 = w8_at<3>(w8_base); break;
                case 4: w8 = w8_at<4>(w8_base); break;
                case 5: w8 = w8_at<5>(w8_base); break;
                case 6: w8 = w8_at<6>(w8_base); break;
                case 7: w8 = w8_at<7>(w8_base); break;
                case 8: w8 = w8_at<8>(w8_base); break;
                case 9: w8 = w8_at<9>(w8_base); break;
                case 10: w8 = w8_at<10>(w8_base); break;
                case 11: w8 = w8_at<11>(w8_base); break;
                case 12: w8 = w8_at<12>(w8_base); break;
                case 13: w8 = w8_at<13>(w8_base); break;
                case 14: w8 = w8_at<14>(w8_base); break;
                default: w8 = w8_at<15>(w8_base); break;
            }
            y[t] = TRI_KNOCK == 5 ? sum + dif : cmulc(sum, w8) - cmul(dif, w8);
            if (t == 0 && j == 0) y[t] = v2f{0.0f, 0.0f};
        }
    };
    v2f ya[16], yb[16];
    TRI_SYNC();  // pass 3 of chain b still reads X
#pragma unroll
    for (int t = 0; t < 16; ++t) X[xslot(j) + kTriStep * t] = va[t];
    if (j == 0) {
        hil[0] = (va[0].x + va[0].y) * 0.5f;  // X[0] / 2
        hil[1] = (va[0].x - va[0].y) * 0.5f;  // X[4096] / 2
        hil[2] = (vb[0].x + vb[0].y) * 0.5f;
        hil[3] = (vb[0].x - vb[0].y) * 0.5f;
    }
    TRI_SYNC();
    hilbert_spectrum(ya, va);
    TRI_SYNC();
#pragma unroll
    for (int t = 0; t < 16; ++t) X[xslot(j) + kTriStep * t] = vb[t];
    TRI_SYNC();
    hilbert_spectrum(yb, vb);
    const float half_x0a = hil[0], half_xna = hil[1], half_x0b = hil[2], half_xnb = hil[3];
    TRI_SYNC();
    tri_dual<true>(ya, yb, X, tw2_lds, tw3, j);  // y[t] = (Im a[2m], Im a[2m+1]), m = j + 256 t

    // ---- 3. the analytic slices s[i] = analytic[2048 + i], i = j + 256 t ----------------------------------------------------
    // real half of column a and the time-weighted window: requested here, consumed behind the two barriers of the gather
    auto load_real_half = [&](float (&xr)[16], uint32_t col_bytes, uint32_t p32) {
        if (direct) {
#pragma unroll
            for (int t = 0; t < 16; ++t) xr[t] = load_f32(windowb, ju * 4u, col_bytes + 8192u + 1024u * (unsigned)t);
        } else {
            const uint32_t q = p32 + 2048u + ju;
#pragma unroll
            for (int t = 0; t < 16; ++t) xr[t] = *reinterpret_cast<const float*>(ring_bytes + (((q + 256u * (unsigned)t) << 2) & bytemask));
        }
    };
    auto load_twindow = [&](float (&twin)[16]) {
#pragma unroll
        for (int t = 0; t < 16; ++t) twin[t] = load_f32(twinb, ju * 4u, 1024u * (unsigned)t);
    };
    float xra[16], twina[16];
    load_real_half(xra, 0u, pa32);
    load_twindow(twina);
    TRI_SYNC();
    float* imag_a = reinterpret_cast<float*>(X);
#pragma unroll
    for (int t = 4; t < 12; ++t) {
        *reinterpret_cast<v2f*>(imag_a + 2 * (j + 256 * t - 1024)) = ya[t];
        *reinterpret_cast<v2f*>(imb + 2 * (j + 256 * t - 1024)) = yb[t];
    }
    TRI_SYNC();
    constexpr int REACH = TERMS > 1 ? TERMS - 1 : 1;  // neighbour bins on either side (the natural-order copy always keeps 3)
    const float c0 = a.cos_c[0];
    float half_c[REACH], dscale[REACH];
#pragma unroll
    for (int m = 1; m <= REACH; ++m) {
        half_c[m - 1] = TERMS > 1 ? 0.5f * a.cos_c[m] : 0.0f;
        dscale[m - 1] = TERMS > 1 ? a.cos_c[m] * ((float)m * 3.14159265358979323846f / 4096.0f) : 0.0f;
    }
    v2f* lin_z = X;  // natural-order bins -3 ... 2111 of Z (slot 3 + k)

    // ---- 4. per column: Z = FFT(s), T = FFT(t w s) as one dual transform; w and w' applied on the bins of Z -----------------
    // A column in three steps — build (the analytic slice and its time-weighted copy from the loaded halves), analyse (dual transform, window on
    // the bins, reassignment, the kept points' positions), store.  (Issuing column a's stores behind column b's build, so that they drain
    // behind its transforms, spilled 32 B and measured +1.5 %: ledger K2-latestore.)
    struct ColumnPoints {
        omx_spectrogram_point pts[9];
        unsigned long long masks[9];
        uint32_t exc, running;
    };
    auto build = [&](v2f (&z)[16], v2f (&z2)[16], const float (&xr)[16], const float (&twin)[16], const float* imag, float half_x0, float half_xn) {
        const float par = (j & 1) ? -half_xn : half_xn;  // (-1)^n: n = 2048 + i has j's parity
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            z[t] = v2f{4096.0f * xr[t] - half_x0 + par, imag[j + 256 * t]};
            z2[t] = v2f{z[t].x * twin[t], z[t].y * twin[t]};
        }
    };
    auto analyse = [&](v2f (&z)[16], v2f (&z2)[16], bool silent, ColumnPoints& o) {
        omx_spectrogram_point (&pts)[9] = o.pts;
        unsigned long long (&masks)[9] = o.masks;
        TRI_SYNC();  // the slice reads of the build / the previous column's neighbour reads still use X
        tri_dual<false>(z, z2, X, tw2_lds, tw3, j);
        float pn[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) pn[t] = load_f32(normb, ju * 4u, 1024u * (unsigned)t);  // (t = 8, j > 0: past the table, reads 0, not used)
        TRI_SYNC();  // pass 3 still reads X
#pragma unroll
        for (int t = 0; t < 8; ++t) lin_z[3 + j + 256 * t] = z[t];
        if (wave_u == 0) lin_z[3 + j + 2048] = z[8];  // bins 2048 ... 2111: the Nyquist bin and its upper neighbours
        if (j >= 253) lin_z[j - 253] = z[15];         // bins -3 ... -1 = bins 4093 ... 4095
        TRI_SYNC();
        auto bins = [&](auto first, auto last) {
            constexpr int T0 = decltype(first)::value, T1 = decltype(last)::value;
            v2f nzm[T1 - T0][REACH], nzp[T1 - T0][REACH];
            if constexpr (TERMS > 1) {
#pragma unroll
                for (int t = T0; t < T1; ++t)
#pragma unroll
                    for (int m = 1; m <= REACH; ++m) {
                        nzm[t - T0][m - 1] = lin_z[3 + j + 256 * t - m];
                        nzp[t - T0][m - 1] = lin_z[3 + j + 256 * t + m];
                    }
            }
#pragma unroll
            for (int t = T0; t < T1; ++t) {
                const uint32_t bin = (uint32_t)(j + 256 * t);
                v2f bb{c0 * z[t].x, c0 * z[t].y}, bd{0.0f, 0.0f};
                if constexpr (TERMS == 2) {  // (the statement order of the two-term form this kernel started with: bit-equal columns)
                    const v2f zm = nzm[t - T0][0], zp = nzp[t - T0][0];
                    const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y};
                    bb = v2f{c0 * z[t].x + half_c[0] * zs.x, c0 * z[t].y + half_c[0] * zs.y};
                    bd = v2f{-dscale[0] * zd.y, dscale[0] * zd.x};  // i c1 (pi / W) (Z[k-1] - Z[k+1])
                } else if constexpr (TERMS > 2) {
#pragma unroll
                    for (int m = 1; m <= REACH; ++m) {
                        const v2f zm = nzm[t - T0][m - 1], zp = nzp[t - T0][m - 1];
                        const v2f zs{zm.x + zp.x, zm.y + zp.y}, zd{zm.x - zp.x, zm.y - zp.y};
                        bb = v2f{bb.x + half_c[m - 1] * zs.x, bb.y + half_c[m - 1] * zs.y};
                        bd = v2f{bd.x - dscale[m - 1] * zd.y, bd.y + dscale[m - 1] * zd.x};
                    }
                }
                bool keep;
                if (TRI_KNOCK == 6) {
                    pts[t].time_offset = bb.x;
                    pts[t].freq_hz = bd.y + z2[t].x;
                    pts[t].power = pn[t];
                    keep = (t < 8 || j == 0) && !silent;
                } else {
                    keep = reassign_flat(bin, bb, bd, z2[t], pn[t], rc, pts[t]) && (t < 8 || j == 0) && !silent;
                }
                masks[t] = __ballot(keep);
                if (lane == 0) scan[t * 4 + wave] = (uint32_t)__popcll(masks[t]);
            }
        };
        if constexpr (TERMS <= 2) {  // four bins at a time: 16 registers of neighbours in flight
            bins(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
            bins(std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});
        } else {                     // 2 (TERMS - 1) neighbours per bin: one bin at a time stays inside 168 registers
            bins(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            bins(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
            bins(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
            bins(std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{});
            bins(std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
            bins(std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{});
            bins(std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{});
            bins(std::integral_constant<int, 7>{}, std::integral_constant<int, 8>{});
        }
        if (wave_u == 0) {
            bins(std::integral_constant<int, 8>{}, std::integral_constant<int, 9>{});
        } else {
            masks[8] = 0ull;
            if (lane == 0) scan[32 + wave] = 0u;
        }
        TRI_SYNC();
        const uint32_t cnt = lane < 36 ? scan[lane] : 0u;
        const uint32_t inc = wave_inclusive_sum(cnt);
        o.exc = inc - cnt;
        o.running = (uint32_t)__builtin_amdgcn_readlane((int)inc, 35);
    };
    auto store = [&](const ColumnPoints& o, uint32_t col, uint32_t* count_out) {
        omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)o.exc, 4 * t + wave_u);
            if (TRI_KNOCK != 4 && ((o.masks[t] >> lane) & 1ull)) {
                const uint32_t pos = before + lanes_below(o.masks[t]);
                store_point(out, pos, o.pts[t]);
            }
        }
        if (j == 0) *count_out = o.running;
    };
    v2f z[16], z2[16];
    ColumnPoints pts;
    build(z, z2, xra, twina, imag_a, half_x0a, half_xna);
    analyse(z, z2, silent_a, pts);
    store(pts, col0, count_a);
    if (have1) {
        float xrb[16], twinb_v[16];
        load_real_half(xrb, hop_bytes, pb32);
        load_twindow(twinb_v);
        build(z, z2, xrb, twinb_v, imb, half_x0b, half_xnb);
        analyse(z, z2, silent_b, pts);
        store(pts, col1, count_b);
    }
}

template <int TERMS>
static void launch_tri(const StftFastArgs& a, hipStream_t stream) {
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_4096_tri_kernel<TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)kTriLds);
    });
    const uint32_t chunks = (a.n_cols + 1u) / 2u;
    hipLaunchKernelGGL(stft_reassigned_4096_tri_kernel<TERMS>, dim3(stream_column_grid(a.n_streams, chunks)), dim3(256), kTriLds, stream, a);
}

void launch_stft_reassigned_4096_tri(const StftFastArgs& a, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return;
    switch (a.cos_terms) {
        case 1: launch_tri<1>(a, stream); break;
        case 2: launch_tri<2>(a, stream); break;
        case 3: launch_tri<3>(a, stream); break;
        default: launch_tri<4>(a, stream); break;
    }
}

}  // namespace omx
