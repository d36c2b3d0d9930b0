// HIP kernels of the spectrum path (gfx950 only).
//   K3a spectrum_power_*   DC-removed windowed real FFT -> |X|^2 * norm per (stream, trace, hop)
//                          (reference src/util/audio/window.rs:66-88, src/visuals/spectrum/processor.rs:215-244)
//   K3b spectrum_levels    per-bin None / Exponential / PeakHold recurrence over hops + raw and
//                          A-weighted dB (reference spectrum/processor.rs:349-402)
#include <algorithm>
#include <cstdlib>

#include "buffer_device.hpp"
#define OMX_FRAME_SYNC_LDS_ONLY 1  // this file's pow2 transforms exchange data through LDS only (fft_pow2_device.hpp: frame_sync)
#include "fft_pow2_device.hpp"
#include "stft_kernels.hpp"
#include "wave_device.hpp"

namespace omx {

// ---- K3a fast: N = 1024 / 2048 / 4096 (size-templated FFT, 4 / 2 / 1 transforms per workgroup).  Two consecutive hops of one (stream, trace) share one complex FFT: hop 2p is
// the real part, hop 2p+1 the imaginary part, X_a = (Z[k] + conj Z[N-k])/2, X_b = (Z[k] - conj Z[N-k])/(2i).
// With AveragingMode::None the dB conversion (:391-401) is fused here and the traces are written directly;
// otherwise the per-hop power goes to the scratch buffer for spectrum_levels_kernel.
// ln(p) * LN_TO_DB for p at or above the state floor (>= f32::MIN_POSITIVE, :332-336, so never a denormal): one v_log_f32
// and one multiply instead of libm's 14-instruction denormal-safe logf — the two differ by < 1e-5 dB
__device__ __forceinline__ float fast_power_db(float p) { return __builtin_amdgcn_logf(p) * 3.0102999566f; }

__device__ __forceinline__ void spectrum_store(const SpectrumPowerArgs& a, uint32_t s, uint32_t tr, uint32_t h, uint32_t k,
                                               float power) {
    if (a.fused_db) {
        if (!(a.emit_all || a.hops == nullptr || h + 1 == a.hops[s])) return;  // ragged, newest hop only: the stream's last hop writes
        float raw = a.floor_db, weighted = a.floor_db;
        if (!(power < a.state_floor)) {
            const float db = logf(power) * 4.3429448f;
            raw = fmaxf(db, a.floor_db);
            weighted = fmaxf(db + a.a_weighting_db[k], a.floor_db);
        }
        const uint32_t ho = a.emit_all ? h : 0;
        float* out = a.traces + (((uint64_t)s * a.n_hops_out + ho) * 2 + a.trace_slot[tr]) * 2 * a.bins + k;
        out[0] = weighted;
        out[a.bins] = raw;
    } else {
        a.power[(((uint64_t)s * a.n_traces + tr) * a.n_hops + h) * a.bins + k] = power;
    }
}

// SPEC_KNOCK: pricing builds of the TUNING library only (`make TUNING=1 EXTRA=-DSPEC_KNOCK=n`; WRONG rows): 1 no stores, 2 no transform,
// 4 no ring loads.  Product objects ignore the macro.
#if !defined(SPEC_KNOCK) || !defined(OMX_TUNING)
#undef SPEC_KNOCK
#define SPEC_KNOCK 0
#endif
#ifndef SPEC_NT  // 1 = non-temporal row stores (A/B: -3 % of the kernel)
#define SPEC_NT 1
#endif
// One workgroup = F hop pairs of one (stream, trace), one complex transform each.
// Tried and dropped (round 5, ledger SP-pipe): a persistent workgroup over 4 ... 32 consecutive pairs, software-pipelined across the pair
// boundary (next pair's samples and this pair's tables loaded ahead of this pair's row stores, no conditional store so that hipcc's
// static s_waitcnt placement keeps the stores in flight, hop == T sliding window of 17 registers) — 1-2 % at equal occupancy: the
// kernel is not waiting at its workgroup edges.
template <int LOGN, bool FUSED>
__device__ __forceinline__ void spectrum_power_pow2_body(const SpectrumPowerArgs& a, uint32_t s, uint32_t tr, uint32_t chunk, v2f* lds) {
    using G = FftGeom<LOGN>;
    constexpr int N = G::N, T = G::T, F = G::FRAMES, WPF = T / 64;  // a transform = T threads; F transforms per workgroup
    v2f* tw2_lds = lds + F * G::LDS;                                  // [256]
    float (*wave_red)[4][WPF] = reinterpret_cast<float (*)[4][WPF]>(tw2_lds + 256);  // [F][max a, max b, min a, min b][WPF]
    // F == 1 (4096 points and up): the frame slot is the workgroup — spelled out so that everything derived from it (hop indices, store
    // bases, the has_b / in_range predicates) is wave-uniform for the compiler: scalar branches and SGPR-base stores instead of
    // exec-mask regions and per-lane 64-bit addresses
    const int fs = F == 1 ? 0 : (int)(threadIdx.x / T), jf = F == 1 ? (int)threadIdx.x : (int)(threadIdx.x % T);
    const unsigned ju = (unsigned)jf;
    v2f* A = lds + fs * G::LDS;
    const uint32_t n_hops_s = spectrum_hops(a, s), pairs_s = (n_hops_s + 1) / 2;  // ragged banks: this stream's own hop count
    if (chunk * F >= pairs_s) return;  // (whole workgroup)
    const char* ring = reinterpret_cast<const char*>(a.ring[tr] + (uint64_t)s * a.cap);
    const uint32_t bytemask = (uint32_t)(a.cap - 1) << 2;  // cap <= 2^30 (host-checked)
    const uint32_t ring_bytes = (uint32_t)min((uint64_t)a.cap * 4u, (uint64_t)0xFFFFFFFCu);
    const uint64_t tail = spectrum_tail(a, s);
    const uint32_t hop_bytes = (uint32_t)a.hop * 4u;
    v2f x[16];  // (.x: hop 2p, .y: hop 2p + 1) at element j + T t — the pair the complex transform carries, and the operand layout of the packed f32 instructions
    auto pair_of = [&](uint32_t chunk, uint32_t& h0, bool& has_b, bool& in_range) {
        const uint32_t pair_raw = chunk * F + (uint32_t)fs;
        in_range = pair_raw < pairs_s;
        h0 = 2 * (in_range ? pair_raw : pairs_s - 1u);  // idle slots shadow the last pair (barriers stay uniform)
        has_b = h0 + 1 < n_hops_s;
    };
    // every global load of a pair is issued in one go.  Buffer-addressed (buffer_device.hpp): one per-lane byte offset and scalar steps
    // while both hops' windows lie in one piece of the ring, per-element wrapped offsets into a descriptor of the whole ring when the
    // window wraps around its end.  Hop b through a descriptor of its own: 0 bytes long when the pair has no second hop — every load
    // then returns 0, no branch.
    auto load_pair = [&](uint32_t chunk, unsigned jl) {
        uint32_t h0;
        bool has_b, in_range;
        pair_of(chunk, h0, has_b, in_range);
        const uint32_t p32 = (uint32_t)(tail + (uint64_t)(a.first_hop + h0) * a.hop);
        const uint32_t off0 = (uint32_t)(((uint64_t)p32 << 2) & bytemask) >> 2;
        const bool direct = F == 1 && (uint64_t)off0 + (has_b ? (uint64_t)a.hop : 0ull) + (uint64_t)N <= a.cap;
        if (SPEC_KNOCK == 4) {  // pricing build: no ring loads
#pragma unroll
            for (int t = 0; t < 16; ++t) x[t] = v2f{(float)(jl + 3u * (unsigned)t) * 1e-4f, (float)(jl ^ (unsigned)t) * 1e-4f};
        } else if (direct) {
            const GlobalBuffer ra = global_buffer(ring + (uint64_t)off0 * 4u, (uint32_t)N * 4u);
            const GlobalBuffer rb = global_buffer(ring + (uint64_t)off0 * 4u + hop_bytes, has_b ? (uint32_t)N * 4u : 0u);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                x[t].x = load_f32(ra, jl * 4u, 4u * (unsigned)T * (unsigned)t);
                x[t].y = load_f32(rb, jl * 4u, 4u * (unsigned)T * (unsigned)t);
            }
        } else {
            const GlobalBuffer ra = global_buffer(ring, ring_bytes), rb = global_buffer(ring, has_b ? ring_bytes : 0u);
            const uint32_t q0 = (p32 + jl) << 2, q1 = q0 + hop_bytes;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                x[t].x = load_f32(ra, (q0 + 4u * (unsigned)T * (unsigned)t) & bytemask, 0u);
                x[t].y = load_f32(rb, (q1 + 4u * (unsigned)T * (unsigned)t) & bytemask, 0u);
            }
        }
    };
    load_pair(chunk, ju);
    float w[16];
    if (F == 1) {
        const GlobalBuffer winb = global_buffer(a.window, (uint32_t)N * 4u);
#pragma unroll
        for (int t = 0; t < 16; ++t) w[t] = load_f32(winb, ju * 4u, 4u * (unsigned)T * (unsigned)t);
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) w[t] = a.window[ju + (unsigned)T * (unsigned)t];
    }
    if (threadIdx.x < 256) tw2_lds[threadIdx.x] = a.tw256[threadIdx.x];
    // the Nyquist bin's table entries, resident in SGPRs (a load in the epilogue would wait for the row stores)
    const float nrm_ny = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bin_norm[N / 2])));
    const float aw_ny = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.a_weighting_db[N / 2])));
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // rows start at multiples of `bins` floats: 4-byte alignment only
    const float floor_v = a.floor_db;

    {
        const unsigned jl = ju;
        const int jfl = jf;
        uint32_t h0;
        bool has_b, in_range;
        pair_of(chunk, h0, has_b, in_range);
        TwiddlesPow2<LOGN> tw;
        tw.tw2 = tw2_lds;
        tw.load(a.tw4096, jl);  // exp(-2 pi i k / N) for this N
        // window.rs:76-79: the hop's mean is the reference's SEQUENTIAL f32 sum / N — taken by window_sums_seq_kernel ahead of this launch
        // (a.hop_sums; round 6: as a tree sum here, a hop with a large constant offset sat 8e-5 of the trace maximum from the reference
        // in bins 0 ... 2).  One reduction round for what is left: the largest / smallest sample of each hop (level equalisation, below).
        const float* hs = a.hop_sums + ((uint64_t)s * a.n_traces + tr) * a.n_hops + h0;
        float sum_a, sum_b;
        if (F == 1) {  // the pair is the workgroup's: scalar loads
            sum_a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hs[0])));
            sum_b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hs[has_b ? 1 : 0])));
        } else {
            sum_a = hs[0];
            sum_b = hs[has_b ? 1 : 0];
        }
        float hi_a = wave::vmax(x[0].x, x[1].x), lo_a = wave::vmin(x[0].x, x[1].x), hi_b = wave::vmax(x[0].y, x[1].y), lo_b = wave::vmin(x[0].y, x[1].y);
#pragma unroll
        for (int t = 2; t < 16; t += 2) {
            hi_a = wave::vmax3(hi_a, x[t].x, x[t + 1].x);
            lo_a = wave::vmin3(lo_a, x[t].x, x[t + 1].x);
            hi_b = wave::vmax3(hi_b, x[t].y, x[t + 1].y);
            lo_b = wave::vmin3(lo_b, x[t].y, x[t + 1].y);
        }
        wave::scan_max2_min2(hi_a, hi_b, lo_a, lo_b);
        const float red[4] = {hi_a, hi_b, lo_a, lo_b};
        if ((jfl & 63) == 63) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wave_red[fs][q][jfl >> 6] = red[q];
        }
        if (F == 1) frame_sync<LOGN>();  // wave partials (and, for the first pair, tw2_lds)
        else lds_workgroup_barrier();            // (tw2_lds is shared by every frame slot)
        hi_a = hi_b = -INFINITY;
        lo_a = lo_b = INFINITY;
#pragma unroll
        for (int i = 0; i < WPF; ++i) {
            hi_a = wave::vmax(hi_a, wave_red[fs][0][i]);
            hi_b = wave::vmax(hi_b, wave_red[fs][1][i]);
            lo_a = wave::vmin(lo_a, wave_red[fs][2][i]);
            lo_b = wave::vmin(lo_b, wave_red[fs][3][i]);
        }
        const v2f mean{sum_a / (float)N, has_b ? sum_b / (float)N : 0.0f};
        // Level equalisation (round 5; see stft_classic_pow2_kernel): the two hops ride one complex transform and the split cancels the
        // partner's spectrum only to ~4e-7 of ITS largest bin — a hop 60 dB under its partner came out 3e-5 of the trace maximum off
        // (tests/test_gpu_parity.py::test_spectrum_quiet_hop_paired_with_a_loud_one).  Each hop is scaled by an exact power of two that
        // brings it to unit level — 2^-e, e = the exponent of its sample RANGE (max - min: within a factor 2 of the largest |x - mean|,
        // and known before the mean is subtracted) — and its powers are scaled back by the exact inverse.  The scale rides the
        // DC removal: fma(x, 2^-e, -mean 2^-e) = (x - mean) 2^-e bit for bit (both products are exact), so the conditioning of a sample
        // is one packed FMA and one packed multiply by the window for both hops.  |e| <= 60 keeps 4^e a normal f32: levels beyond
        // 2^+-60 are not audio, and are equalised as far as that.
        const float range_a = hi_a - lo_a, range_b = hi_b - lo_b;
        int ea = (range_a > 0.0f && range_a < INFINITY) ? __builtin_amdgcn_frexp_expf(range_a) : 0;
        int eb = (range_b > 0.0f && range_b < INFINITY) ? __builtin_amdgcn_frexp_expf(range_b) : 0;
        ea = min(max(ea, -60), 60);
        eb = min(max(eb, -60), 60);
        const v2f scale{wave::pow2f(-ea), wave::pow2f(-eb)};
        const v2f shift = -(mean * scale);
        v2f v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const v2f centred{__builtin_fmaf(x[t].x, scale.x, shift.x), __builtin_fmaf(x[t].y, scale.y, shift.y)};
            v[t] = centred * v2f{w[t], w[t]};
        }
        if (SPEC_KNOCK != 2) fftp_inplace<false, LOGN>(v, A, jfl, tw);
        frame_sync<LOGN>();
        const int own_base = pad16(jfl);  // pad16(j + T t) = pad16(j) + (T + T / 16) t
#pragma unroll
        for (int t = 0; t < 16; ++t) A[own_base + (T + T / 16) * t] = v[t];
        // the per-bin tables of both epilogue rounds: issued here, ahead of the row stores (a load between the rounds would wait for the first round's stores)
        float4 nrm4[2], aw4[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t k0 = (unsigned)r * 4u * (unsigned)T + 4u * jl;
            nrm4[r] = *reinterpret_cast<const float4*>(a.bin_norm + k0);
            aw4[r] = *reinterpret_cast<const float4*>(a.a_weighting_db + k0);  // (the table exists in every averaging mode)
        }
        frame_sync<LOGN>();
        if (in_range) {
        // Epilogue, FOUR consecutive bins per thread and round (bins 4 j ... 4 j + 3, then 4 T + 4 j ...; thread 0 adds the Nyquist bin): every
        // row is written by 16-byte stores — a wavefront moves 1 KiB per store instruction where the bin-per-lane order moved 256 B.
        // Both Z[k] and its partner come from the natural-order copy.  The split keeps the (hop a, hop b) pair in one register pair:
        //   U = Z + Zr = 2 (Re Xa, Re Xb),   V = (Z.y - Zr.y, Zr.x - Z.x) = 2 (Im Xa, Im Xb),   P = U U + V V = 4 (|Xa|^2, |Xb|^2)
        // and the factors 1/4, 4^ea / 4^eb (the level equalisation undone) and the bin's normalisation are exact powers of two times
        // one rounding: P * norm * (4^e / 4) rounds where (|X|^2 norm) rounded before.
        float* out0 = nullptr;
        if (FUSED)
            out0 = a.traces + (((uint64_t)s * a.n_hops_out + (a.emit_all ? h0 : 0)) * 2 + a.trace_slot[tr]) * 2 * a.bins;
        const uint32_t hop_stride = a.emit_all ? 4u * a.bins : 0u;  // floats between consecutive hops of one stream
        float* pw = FUSED ? nullptr : a.power + (((uint64_t)s * a.n_traces + tr) * a.n_hops + h0) * a.bins;
        const bool write_a = a.emit_all || h0 + 1 == n_hops_s, write_b = has_b && (a.emit_all || h0 + 2 == n_hops_s);
        const v2f unscale{wave::pow2f(2 * ea - 2), wave::pow2f(2 * eb - 2)};
        auto split_power = [&](v2f z, v2f zr, float nrm) -> v2f {
            const v2f u = z + zr;
            const v2f q{z.y - zr.y, zr.x - z.x};
            const v2f p = u * u + q * q;
            return (p * v2f{nrm, nrm}) * unscale;
        };
        // update_outputs with AveragingMode::None (:391-401): a power under the state floor shows the floor in both rows — taken as
        // log(0) = -inf here, which the two max() turn into the floor (and -inf + A-weighting stays -inf)
        auto levels = [&](v2f p, float awk, float& wt_a, float& raw_a, float& wt_b, float& raw_b) {
            const v2f kept{p.x < a.state_floor ? 0.0f : p.x, p.y < a.state_floor ? 0.0f : p.y};
            const v2f db = v2f{__builtin_amdgcn_logf(kept.x), __builtin_amdgcn_logf(kept.y)} * v2f{3.0102999566f, 3.0102999566f};
            const v2f dbw = db + v2f{awk, awk};
            raw_a = wave::vmax(db.x, floor_v);
            raw_b = wave::vmax(db.y, floor_v);
            wt_a = wave::vmax(dbw.x, floor_v);
            wt_b = wave::vmax(dbw.y, floor_v);
        };
        auto put = [&](float* p, f4u v4) {  // (rows are written once and read by another kernel: non-temporal, -3 % of the kernel)
            if (SPEC_NT) __builtin_nontemporal_store(v4, reinterpret_cast<f4u*>(p));
            else *reinterpret_cast<f4u*>(p) = v4;
        };
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t k0 = (unsigned)r * 4u * (unsigned)T + 4u * jl;
            const int zb = pad16((int)k0);                           // bins k0 ... k0 + 3 share a 16-group: consecutive slots
            const int pb3 = pad16(N - (int)k0 - 4);                  // partners N - k0 - 1 ... N - k0 - 3 at pb3 + 3 ... pb3 + 1
            const int p0 = (r == 0 && jfl == 0) ? 0 : pad16(N - (int)k0);  // partner of k0 itself (bin 0 pairs with itself)
            v2f z[4], zr[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) z[c] = A[zb + c];
            zr[0] = A[p0];
#pragma unroll
            for (int c = 1; c < 4; ++c) zr[c] = A[pb3 + 4 - c];
            const float4 nrm = nrm4[r];
            const float nrmv[4] = {nrm.x, nrm.y, nrm.z, nrm.w};
            v2f p[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = split_power(z[c], zr[c], nrmv[c]);
            if (FUSED) {
                const float4 aw = aw4[r];
                const float awv[4] = {aw.x, aw.y, aw.z, aw.w};
                f4u wt_a, raw_a, wt_b, raw_b;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float w0, r0, w1, r1;
                    levels(p[c], awv[c], w0, r0, w1, r1);
                    wt_a[c] = w0;
                    raw_a[c] = r0;
                    wt_b[c] = w1;
                    raw_b[c] = r1;
                }
                if (SPEC_KNOCK == 1) {
                    if (wt_a[0] + raw_a[1] + wt_b[2] + raw_b[3] == 1.2345f) out0[k0] = wt_a[0];
                    continue;
                }
                // emit_all == 0: only the newest hop is materialised (slot 0).  A lock-step call launches that hop alone (n_hops == 1);
                // a ragged call launches every hop and the stream's last one writes
                if (write_a) {
                    put(out0 + k0, wt_a);
                    put(out0 + a.bins + k0, raw_a);
                }
                if (write_b) {
                    put(out0 + hop_stride + k0, wt_b);
                    put(out0 + hop_stride + a.bins + k0, raw_b);
                }
            } else {
                *reinterpret_cast<f4u*>(pw + k0) = f4u{p[0].x, p[1].x, p[2].x, p[3].x};
                if (has_b) *reinterpret_cast<f4u*>(pw + a.bins + k0) = f4u{p[0].y, p[1].y, p[2].y, p[3].y};
            }
        }
        if (jfl == 0) {  // Nyquist bin N / 2 pairs with itself
            const uint32_t k = (unsigned)N / 2u;
            const v2f z = A[pad16(N / 2)];
            const v2f pn = split_power(z, z, nrm_ny);
            if (FUSED) {
                float w0, r0, w1, r1;
                levels(pn, aw_ny, w0, r0, w1, r1);
                if (write_a && SPEC_KNOCK != 1) {
                    out0[k] = w0;
                    out0[a.bins + k] = r0;
                }
                if (write_b && SPEC_KNOCK != 1) {
                    out0[hop_stride + k] = w1;
                    out0[hop_stride + a.bins + k] = r1;
                }
            } else {
                pw[k] = pn.x;
                if (has_b) pw[a.bins + k] = pn.y;
            }
        }
        }
    }
}

// FUSED: AveragingMode::None — the dB rows are written here; otherwise the per-hop powers go to the scratch buffer for spectrum_levels_kernel
// (a template parameter, not a branch: with both forms in one body their row stores were tail-merged behind one `s_waitcnt vmcnt(0)`).
template <int LOGN, bool FUSED>
__global__ __launch_bounds__(FftGeom<LOGN>::WG, LOGN == 12 ? 4 : 1) void spectrum_power_pow2_kernel(SpectrumPowerArgs a) {
    using G = FftGeom<LOGN>;
    constexpr int F = G::FRAMES;
    extern __shared__ __attribute__((aligned(16))) unsigned char spectrum_smem[];
    v2f* lds = reinterpret_cast<v2f*>(spectrum_smem);                 // [F][G::LDS] + tw2 + wave partials
    const uint32_t pairs = (a.n_hops + 1) / 2, chunks = (pairs + F - 1) / F;
    // XCD-aware map (as block_to_stream_column in stft_kernels.hip): block b runs on XCD b % 8 and every (stream, trace) is pinned to
    // one XCD, so the 16 hops that share a sample find it in that XCD's L2 (chunk-fastest over all XCDs fetched the rings 8 times:
    // 524 MB per launch against 67 MB of new samples)
    const uint32_t xcd = blockIdx.x & 7u, bq = blockIdx.x >> 3;
    const uint32_t chunk = bq % chunks, st = (bq / chunks) * 8u + xcd;
    if (st >= a.n_streams * a.n_traces) return;
    const uint32_t tr = st % a.n_traces, s = st / a.n_traces;
    spectrum_power_pow2_body<LOGN, FUSED>(a, s, tr, chunk, lds);
}

// ---- K3a split: N = 16384 (the reference's default spectrum size) as ONE packed-real transform per hop on the tuned 4096-point dual
// transform (fft_device.hpp): z[m] = (x[2m], x[2m+1]), M = 8192 complex points = the dual transform of the even / odd interleaved
// halves + a radix-2 step in registers (cf. stft8192_kernels.hip), then the real-input split
//   X[k] = (Z[k] + conj Z[M-k]) / 2 - i w^k (Z[k] - conj Z[M-k]) / 2,   w = exp(-2 pi i / N)
// with the partners read from a natural-order copy in both LDS buffers.  256 threads per hop, two workgroups per CU; the
// size-templated kernel above runs this size with 1024 threads, four in-place passes and one workgroup per CU (19.4 M hops/s;
// this one 23.0 M).  The same form for N = 8192 (two hops per workgroup as one dual transform) measured the same as the
// size-templated kernel (46 M hops/s) and was not kept.
__global__ __launch_bounds__(256, 2) void spectrum_power_16384_kernel(SpectrumPowerArgs a) {
    constexpr int N = 16384, M = N / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char spectrum_smem[];
    v2f* A = reinterpret_cast<v2f*>(spectrum_smem);
    v2f* B = A + FFT4096_LDS;        // contiguous with A: one 8704-slot buffer for the partner exchange
    v2f* tw2_lds = B + FFT4096_LDS;  // [256]
    const uint32_t xcd = blockIdx.x & 7u, bq = blockIdx.x >> 3;
    const uint32_t h0 = bq % a.n_hops, st = (bq / a.n_hops) * 8u + xcd;  // XCD-aware: a (stream, trace) stays on one XCD
    if (st >= a.n_streams * a.n_traces) return;
    const uint32_t tr = st % a.n_traces, s = st / a.n_traces;
    const uint32_t n_hops_s = spectrum_hops(a, s);  // ragged banks: this stream's own hop count
    if (h0 >= n_hops_s) return;
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;
    const float* ring = a.ring[tr] + (uint64_t)s * a.cap;
    const uint32_t mask32 = (uint32_t)(a.cap - 1);  // cap <= 2^30 (host-checked)
    const uint32_t p32 = (uint32_t)(spectrum_tail(a, s) + (uint64_t)(a.first_hop + h0) * a.hop);
    const v2f* T = a.tw4096;  // exp(-2 pi i k / N), k < N
    struct TW {
        const v2f* tw2;
        const v2f* table;
        unsigned step;  // 4 j
        __device__ __forceinline__ v2f w2(unsigned k, int t) const { return tw2[k * (unsigned)t]; }
        __device__ __forceinline__ v2f w3(int t) const { return table[step * (unsigned)t]; }  // exp(-2 pi i j t / 4096)
    };
    const TW tw{tw2_lds, T, ju * 4u};
    tw2_lds[j] = a.tw256[ju];

    // ---- load, remove the mean, window (window.rs:66-88) ----------------------------------------------------------------------------------
    // v0 / v1 = even / odd packed elements 2 (j + 256 t) + r -> samples 4 (j + 256 t) + 2r, + 1
    v2f v0[16], v1[16];
    // the samples of the hop lie in one piece of the ring and pairs are 8-byte aligned: buffer loads (buffer_device.hpp)
    const uint32_t off0 = p32 & mask32;
    const bool direct = (uint64_t)off0 + (uint64_t)N <= a.cap && (p32 & 1u) == 0;
    const GlobalBuffer window = global_buffer(ring + off0, (uint32_t)N * 4u);
    const GlobalBuffer winb = global_buffer(a.window, (uint32_t)N * 4u), normb = global_buffer(a.bin_norm, (uint32_t)(M + 1) * 4u),
                       awb = global_buffer(a.a_weighting_db, (uint32_t)(M + 1) * 4u), Tb = global_buffer(T, (uint32_t)N * 8u);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if (direct) {
            v0[t] = load_v2f(window, ju * 16u, 4096u * (unsigned)t);
            v1[t] = load_v2f(window, ju * 16u + 8u, 4096u * (unsigned)t);
        } else {
            const uint32_t q = p32 + 4u * (ju + 256u * (unsigned)t);
            v0[t] = v2f{ring[q & mask32], ring[(q + 1u) & mask32]};
            v1[t] = v2f{ring[(q + 2u) & mask32], ring[(q + 3u) & mask32]};
        }
    }
    lds_workgroup_barrier();  // tw2_lds
    // window.rs:76-79: the reference's sequential f32 sum of the hop, from window_sums_seq_kernel (round 6; a tree sum before)
    const float sum0 = __builtin_bit_cast(
        float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.hop_sums[((uint64_t)s * a.n_traces + tr) * a.n_hops + h0])));
    const float mean0 = sum0 / (float)N;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const v2f wl = load_v2f(winb, ju * 16u, 4096u * (unsigned)t), wh = load_v2f(winb, ju * 16u + 8u, 4096u * (unsigned)t);
        v0[t] = v2f{(v0[t].x - mean0) * wl.x, (v0[t].y - mean0) * wl.y};
        v1[t] = v2f{(v1[t].x - mean0) * wh.x, (v1[t].y - mean0) * wh.y};
    }
    fft4096t_dual<false>(v0, v1, A, B, j, tw);

    // ---- radix-2 step and the natural-order copy for the partner reads -------------------------------------------------------------------------
    lds_workgroup_barrier();  // pass 3 still reads A and B
#pragma unroll
    for (int t = 0; t < 16; ++t) {  // Z[k] = F0[k] + w2^k F1[k], Z[k + 4096] = F0[k] - w2^k F1[k], w2 = exp(-2 pi i / 8192)
        const v2f m = cmul(v1[t], load_v2f(Tb, ju * 16u, 4096u * (unsigned)t));
        const v2f lo = v0[t] + m, hi = v0[t] - m;
        v0[t] = lo;
        v1[t] = hi;
        A[pad16(j + 256 * t)] = lo;
        A[pad16(j + 256 * t + 4096)] = hi;
    }
    lds_workgroup_barrier();

    // ---- real-input split, power, dB; bins k = j + 256 t' (t' < 32) and bin M (thread 0) ---------------------------------------------------------
    constexpr int PER = M / 256;
    const int part = (j ? pad16(M - j) : M + M / 16) - 272 * (PER - 1);  // partner Z[M - k] of k = j + 256 t': pad16(M - j) - 272 t'
    float* out0 = nullptr;
    if (a.fused_db) out0 = a.traces + (((uint64_t)s * a.n_hops_out + (a.emit_all ? h0 : 0)) * 2 + a.trace_slot[tr]) * 2 * a.bins;
    auto emit = [&](uint32_t k, v2f x, float norm, float aw) {
        const float p = (x.x * x.x + x.y * x.y) * norm;
        if (a.fused_db) {  // update_outputs with AveragingMode::None (:391-401), branch-free
            if (!(a.emit_all || h0 + 1 == n_hops_s)) return;  // emit_all == 0: the stream's last hop is the one materialised
            const float db = fast_power_db(p);
            const bool low = p < a.state_floor;
            // (rows are written once and read by another kernel: non-temporal, 2.815 -> 2.74 ms per 65 536 hops)
            __builtin_nontemporal_store(low ? a.floor_db : fmaxf(db + aw, a.floor_db), out0 + k);
            __builtin_nontemporal_store(low ? a.floor_db : fmaxf(db, a.floor_db), out0 + a.bins + k);
        } else {
            a.power[(((uint64_t)s * a.n_traces + tr) * a.n_hops + h0) * a.bins + k] = p;
        }
    };
    auto split = [&](v2f z, v2f zr, v2f w) {  // (Z + conj Zr)/2 - i w (Z - conj Zr)/2
        const v2f e{(z.x + zr.x) * 0.5f, (z.y - zr.y) * 0.5f}, o{(z.x - zr.x) * 0.5f, (z.y + zr.y) * 0.5f};
        const v2f wo = cmul(o, w);
        return v2f{e.x + wo.y, e.y - wo.x};
    };
#pragma unroll
    for (int g = 0; g < PER; g += 8) {  // eight bins at a time: their table and partner reads are issued together
        v2f w[8], zr[8];
        float norm[8], aw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tp = g + u;
            w[u] = load_v2f(Tb, ju * 8u, 2048u * (unsigned)tp);
            norm[u] = load_f32(normb, ju * 4u, 1024u * (unsigned)tp);
            aw[u] = a.fused_db ? load_f32(awb, ju * 4u, 1024u * (unsigned)tp) : 0.0f;
            zr[u] = A[part + 272 * (PER - 1 - tp)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tp = g + u;
            const v2f z = tp < 16 ? v0[tp & 15] : v1[tp & 15];
            emit(ju + 256u * (unsigned)tp, split(z, (tp == 0 && j == 0) ? v0[0] : zr[u], w[u]), norm[u], aw[u]);  // k = 0: Z[M] is Z[0]
        }
    }
    if (j == 0) emit((uint32_t)M, v2f{v0[0].x - v0[0].y, 0.0f}, a.bin_norm[M], a.fused_db ? a.a_weighting_db[M] : 0.0f);  // X[M] = Re Z[0] - Im Z[0]
}

static void launch_spectrum_16384(const SpectrumPowerArgs& a, uint32_t stream_traces, hipStream_t stream) {
    const size_t lds = (size_t)(2 * FFT4096_LDS + 256) * sizeof(v2f);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spectrum_power_16384_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(spectrum_power_16384_kernel, dim3(stream_column_grid(stream_traces, a.n_hops)), dim3(256), lds, stream, a);
}

// ---- K3a generic: any power-of-two N, radix-2 in a global workspace, reference operation order ------
__global__ __launch_bounds__(256) void spectrum_power_generic_kernel(SpectrumPowerArgs a) {
    __shared__ float mean_sh;
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    const uint64_t total = (uint64_t)a.n_streams * a.n_traces * a.n_hops;
    v2f* ws = a.workspace + (uint64_t)blockIdx.x * ((uint64_t)a.fft_size + a.blu.m);
    for (uint64_t item = blockIdx.x; item < total; item += gridDim.x) {
        const uint32_t h = (uint32_t)(item % a.n_hops), st = (uint32_t)(item / a.n_hops);
        const uint32_t tr = st % a.n_traces, s = st / a.n_traces;
        if (h >= spectrum_hops(a, s)) continue;  // ragged banks: past this stream's own hop count (workgroup-uniform)
        const float* ring = a.ring[tr] + (uint64_t)s * a.cap;
        const uint64_t mask = a.cap - 1;
        const uint64_t p0 = spectrum_tail(a, s) + (uint64_t)(a.first_hop + h) * a.hop;
        lds_workgroup_barrier();
        if (tid == 0) {
            float sum = -0.0f;
            for (uint32_t i = 0; i < a.fft_size; ++i) sum = sum + ring[(p0 + i) & mask];
            mean_sh = sum / (float)a.fft_size;
        }
        lds_workgroup_barrier();
        const float mean = mean_sh;
        for (uint32_t i = tid; i < a.fft_size; i += nt) ws[i] = v2f{(ring[(p0 + i) & mask] - mean) * a.window[i], 0.0f};
        fft_forward_any(ws, a.fft_size, a.log_fft, a.tw_fft, ws + a.fft_size, a.blu, tid, nt);
        for (uint32_t i = tid; i < a.bins; i += nt) {
            const v2f c = ws[i];
            spectrum_store(a, s, tr, h, i, (c.x * c.x + c.y * c.y) * a.bin_norm[i]);
        }
    }
}

template <int LOGN, bool FUSED>
static void launch_spectrum_pow2_form(const SpectrumPowerArgs& a, uint32_t stream_traces, uint32_t hop_pairs, hipStream_t stream) {
    using G = FftGeom<LOGN>;
    constexpr int F = G::FRAMES, WPF = G::T / 64;
    const size_t lds = (size_t)(F * G::LDS + 256) * sizeof(v2f) + (size_t)F * 4 * WPF * sizeof(float);  // + the wave partials (max, min of both hops)
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spectrum_power_pow2_kernel<LOGN, FUSED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const uint32_t chunks = (hop_pairs + F - 1) / F;
    hipLaunchKernelGGL((spectrum_power_pow2_kernel<LOGN, FUSED>), dim3(stream_column_grid(stream_traces, chunks)), dim3(G::WG), lds, stream, a);
}
template <int LOGN>
static void launch_spectrum_pow2(const SpectrumPowerArgs& a, uint32_t stream_traces, uint32_t hop_pairs, hipStream_t stream) {
    if (a.fused_db) launch_spectrum_pow2_form<LOGN, true>(a, stream_traces, hop_pairs, stream);
    else launch_spectrum_pow2_form<LOGN, false>(a, stream_traces, hop_pairs, stream);
}

void launch_spectrum_power(const SpectrumPowerArgs& a, bool fast4096, uint32_t generic_wgs, hipStream_t stream) {
    const uint64_t total = (uint64_t)a.n_streams * a.n_traces * a.n_hops;
    if (total == 0) return;
    const uint64_t pairs = (uint64_t)a.n_streams * a.n_traces * ((a.n_hops + 1) / 2);
    const uint32_t hop_pairs = (a.n_hops + 1) / 2, st = a.n_streams * a.n_traces;
    (void)pairs;
    static const bool templated = [] { const char* e = tuning_env("OMX_SPECTRUM_TEMPLATED"); return e && atoi(e) == 1; }();  // A/B: the size-templated kernel
    if (fast4096 && a.fft_size == 16384 && !templated) launch_spectrum_16384(a, st, stream);
    else if (fast4096 && a.fft_size == 16384) launch_spectrum_pow2<14>(a, st, hop_pairs, stream);
    else if (fast4096 && a.fft_size == 8192) launch_spectrum_pow2<13>(a, st, hop_pairs, stream);
    else if (fast4096 && a.fft_size == 4096) launch_spectrum_pow2<12>(a, st, hop_pairs, stream);
    else if (fast4096 && a.fft_size == 2048) launch_spectrum_pow2<11>(a, st, hop_pairs, stream);
    else if (fast4096 && a.fft_size == 1024) launch_spectrum_pow2<10>(a, st, hop_pairs, stream);
    else hipLaunchKernelGGL(spectrum_power_generic_kernel, dim3(generic_wgs), dim3(256), 0, stream, a);
}

// ---- K3b: per-bin recurrences + dB ------------------------------------------------------------------
__global__ __launch_bounds__(256) void spectrum_levels_kernel(SpectrumLevelsArgs a) {
    const uint32_t bin = blockIdx.x * 256 + threadIdx.x;
    const uint32_t tr = blockIdx.y, s = blockIdx.z;
    if (bin >= a.bins) return;
    const uint32_t slot = a.trace_slot[tr];  // 0/1: which output trace this active trace fills
    float* state = a.smoothed ? a.smoothed + ((uint64_t)s * 2 + slot) * a.bins + bin : nullptr;
    float st = state ? *state : 0.0f;
    const float aw = a.a_weighting_db[bin];
    const float* pw = a.power + ((uint64_t)s * a.n_traces + tr) * a.n_hops * a.bins + bin;
    const uint32_t n_hops_s = a.hops ? a.hops[s] : a.n_hops;  // ragged banks: this stream's own hop count
    for (uint32_t h = 0; h < n_hops_s; ++h) {
        const float power = pw[(uint64_t)h * a.bins];
        float p = power;
        if (a.mode == OMX_AVERAGING_EXPONENTIAL) {  // :366-379
            st = (st <= 0.0f) ? power : st * a.alpha + power * (1.0f - a.alpha);
            if (st < a.state_floor) st = 0.0f;
            p = st;
        } else if (a.mode == OMX_AVERAGING_PEAK_HOLD) {  // :380-389
            st = fmaxf(st * a.decay, power);
            if (st < a.state_floor) st = 0.0f;
            p = st;
        }
        const bool last = h + 1 == n_hops_s;
        if (a.emit_all || last) {
            const uint32_t ho = a.emit_all ? h : 0;
            float* out = a.traces + (((uint64_t)s * a.n_hops_out + ho) * 2 + slot) * 2 * a.bins + bin;
            float raw = a.floor_db, weighted = a.floor_db;  // :392-401
            if (!(p < a.state_floor)) {
                const float db = logf(p) * 4.3429448f;
                raw = fmaxf(db, a.floor_db);
                weighted = fmaxf(db + aw, a.floor_db);
            }
            out[0] = weighted;
            out[a.bins] = raw;
        }
    }
    if (state) *state = st;
}

void launch_spectrum_levels(const SpectrumLevelsArgs& a, hipStream_t stream) {
    if (a.n_hops == 0 || a.n_traces == 0) return;
    dim3 grid((a.bins + 255) / 256, a.n_traces, a.n_streams);
    hipLaunchKernelGGL(spectrum_levels_kernel, grid, dim3(256), 0, stream, a);
}

__global__ void fill_kernel(float* p, uint64_t n, float v) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill(float* p, uint64_t n, float v, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, n, v);
}

// ---- ragged bank: per-stream state machine ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void spectrum_plan_kernel(SpectrumPlanArgs a) {
    const uint32_t s = blockIdx.x * 64u + threadIdx.x;
    if (s >= a.n_streams) return;
    uint64_t head = a.head[s], tail = a.tail[s], pending_skip = a.pending_skip[s];
    if (a.reset_mask && a.reset_mask[s]) {  // reset_audio (:112-118): the pending audio is dropped (the level state: spectrum_reset_streams)
        tail = head;
        pending_skip = 0;
    }
    const uint64_t frames = a.frames[s];
    uint32_t n_hops = 0, skip32 = 0, count32 = 0;
    const uint64_t head_before = head;
    uint64_t tail0 = tail;
    if (frames != 0) {  // block.is_empty() -> None (:256)
        const uint64_t skip = min(pending_skip, frames);  // push_sources (:271-298)
        pending_skip -= skip;
        const uint64_t count = frames - skip;
        skip32 = (uint32_t)skip;
        count32 = (uint32_t)count;
        head += count;
        while (head - tail >= a.fft_size && n_hops < a.max_hops) {  // process_ready_windows (:179-213)
            const uint64_t len = head - tail, d = min(a.hop, len);
            tail += d;
            pending_skip += a.hop - d;
            ++n_hops;
        }
    }
    // the window folds of this call (SpectrumBank::launch_window_sums_for has the lock-step form of the same rule): carried while the
    // stream's positions move by pushes alone and the call brings few samples, walked otherwise.  The bookkeeping moves here, ahead of
    // the fold kernels, which only read it.
    if (a.fold_mode) {
        uint32_t mode = kFoldNone, valid = a.carry_valid[s];
        if (a.reset_mask && a.reset_mask[s]) valid = 0u;
        if (n_hops != 0u) {
            const uint64_t from = valid ? a.carry_pos[s] : tail0;
            if (a.fold_slots != 0u && head - from <= a.fft_size + 3u * a.hop) {
                mode = kFoldCarry;
                const uint32_t slot0 = valid ? a.carry_slot0[s] : 0u;
                a.fold_from[s] = from;
                a.fold_slot0[s] = slot0;
                a.carry_slot0[s] = (uint32_t)(((uint64_t)slot0 + n_hops) % a.fold_slots);
                a.carry_pos[s] = head;
                valid = 1u;
            } else {
                mode = kFoldWalk;
                valid = 0u;
            }
        }
        a.carry_valid[s] = valid;
        a.fold_mode[s] = mode;
    }
    a.ing_skip[s] = skip32;
    a.ing_count[s] = count32;
    a.ing_head[s] = head_before;
    a.hop_tail[s] = tail0;
    a.n_hops[s] = n_hops;
    a.head[s] = head;
    a.tail[s] = tail;
    a.pending_skip[s] = pending_skip;
}
void launch_spectrum_plan(const SpectrumPlanArgs& a, hipStream_t stream) {
    if (a.n_streams == 0) return;
    hipLaunchKernelGGL(spectrum_plan_kernel, dim3((a.n_streams + 63u) / 64u), dim3(64), 0, stream, a);
}

__global__ __launch_bounds__(256) void spectrum_reset_streams_kernel(const uint8_t* reset_mask, float* smoothed, uint64_t smoothed_per_stream,
                                                                     float* traces, uint64_t traces_per_stream, float floor_db) {
    const uint32_t s = blockIdx.y;
    if (!reset_mask[s]) return;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (smoothed && i < smoothed_per_stream) smoothed[(uint64_t)s * smoothed_per_stream + i] = 0.0f;
    if (traces && i < traces_per_stream) traces[(uint64_t)s * traces_per_stream + i] = floor_db;
}
void launch_spectrum_reset_streams(const uint8_t* reset_mask, uint32_t n_streams, float* smoothed, uint64_t smoothed_per_stream, float* traces,
                                   uint64_t traces_per_stream, float floor_db, hipStream_t stream) {
    const uint64_t most = std::max(smoothed ? smoothed_per_stream : 0, traces ? traces_per_stream : 0);
    if (n_streams == 0 || most == 0) return;
    hipLaunchKernelGGL(spectrum_reset_streams_kernel, dim3((uint32_t)((most + 255) / 256), n_streams), dim3(256), 0, stream, reset_mask, smoothed,
                       smoothed_per_stream, traces, traces_per_stream, floor_db);
}

}  // namespace omx
