// K5c: the stereometer recurrences evaluated chunk-parallel in time (one chunk = one block of a bank call).
//
// The sequential kernels (stereometer_kernels.hip) give one lane to a stream and walk its frames in order: bit-identical to the
// reference, but a 256-stream bank occupies 16-20 wavefronts of a 1024-SIMD part (0.5 % of HBM).  Every
// recurrence on this path is LINEAR in its state — Biquad::process (reference src/dsp.rs:422-432) while its output stays finite,
// Correlator::update (src/visuals/stereometer/processor.rs:40-46) always — so a call of n_blocks blocks splits into
//   pass A   per (stream, block): the band filters from a ZERO state over the block -> zero-state end state e_c         (parallel)
//   scan 1   per (stream, band, channel): true start state of every block, s_{c+1} = flush(T s_c + e_c), T = the L-frame
//            zero-input transition of the cascade (host, f64) — a wave-parallel scan over the blocks           (6 shuffle steps)
//   pass B   per (stream, block): the filters again from the TRUE start state -> band outputs; the correlator moments from a
//            zero state (f64); the newest `hist_frames` pairs of the call go to the history rings                     (parallel)
//   scan 2   per (stream, band): m_{c+1} = flush((1 - alpha)^L m_c + zs_c) -> the per-block correlations       (6 shuffle steps)
// Filter outputs differ from the sequential evaluation only through the rounding of the block-boundary states (f32, ~1e-7
// relative); the parity bars are rho <= 1e-6 and points <= 1e-6 (tests/test_gpu_parity_meters.py).  What is NOT linear — the
// non-finite reset of Biquad::process — is detected (poison accumulator, non-finite input) and sends the WHOLE call through the
// sequential kernel again from the saved state, so such input keeps the reference's behaviour exactly.
// Chunks may be shorter than blocks (cpb = chunks per block: 2 or 4 when a call has too few (stream, block) items to fill the SIMDs,
// e.g. 256 streams x 64 blocks = 256 workgroups of three wavefronts): every kernel then works on chunks, the correlations of a
// block are those of its last chunk, and the once-per-block denormal flush is applied at block ends only.
// Single-stream handles and short calls stay on the sequential kernels (bit-identical form, the A/B reference).
#include <type_traits>
#include "stereometer.hpp"

namespace omx {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));  // (left, right)

constexpr int STEP = 16;         // frames per staged tile
constexpr int ROW_FLOATS = 34;   // 16 frames x 2 + 2 pad floats: lanes read their own row with conflict-free ds_read_b64

__device__ __forceinline__ v2f biquad_lr(const BiquadCoef& c, v2f& z0, v2f& z1, v2f x, v2f& poison) {  // dsp.rs:422-432, L and R at once
    // Fused multiply-adds (5 packed operations per section where the reference's unfused statement order takes 9).  This form differs
    // from the sequential order through the block-boundary states anyway, and both sit on the f32 noise floor of these sections
    // (1e-5 ... 2e-5 of full scale, tests/test_kat_stereometer.py): against the oracle the fused form measures points 2.5e-5 (unfused
    // 2.9e-5), rho 1.8e-7 on bands within 16 dB of the full level (the same).  The sequential kernels keep the reference's order.
#ifdef STEREO_CHUNK_UNFUSED   // A/B build: the reference's statement order (dsp.rs:422-432), no fused multiply-add
    const v2f out = c.b[0] * x + z0;
    z0 = c.b[1] * x - c.a[0] * out + z1;
    z1 = c.b[2] * x - c.a[1] * out;
#else
    const v2f out = __builtin_elementwise_fma(v2f{c.b[0], c.b[0]}, x, z0);
    z0 = __builtin_elementwise_fma(v2f{-c.a[0], -c.a[0]}, out, __builtin_elementwise_fma(v2f{c.b[1], c.b[1]}, x, z1));
    z1 = __builtin_elementwise_fma(v2f{-c.a[1], -c.a[1]}, out, c.b[2] * x);
#endif
    poison = __builtin_elementwise_fma(out, v2f{0.0f, 0.0f}, poison);  // NaN as soon as an output was inf / NaN
    return out;
}

// The LOW band's zero-state pass runs in f64 (round 4).  Its block-boundary states then follow the exact trajectory of the recurrence, and
// pass B — f32, the reference's precision — restarts every block from a state that carries no rounding history of its own.  With f32
// boundary states the low band's correlation sat 4 ... 16x further from exact arithmetic than the reference's own f32 evaluation
// (tests/test_exact_f64.py, exact-f64 third leg: two independent f32 error processes — the zero-state pass's and pass B's — add up
// in a band whose signal is 35 dB below the full level); with exact boundary states it is as close as the oracle.  Role 0 has two
// sections where roles 1 and 2 have four, so the f64 sections (half rate) fill what was idle time of that wavefront in pass A.
__device__ __forceinline__ void biquad_lr_f64(const BiquadCoef& c, double (&z0)[2], double (&z1)[2], double (&x)[2], double& poison) {
    const double b0 = (double)c.b[0], b1 = (double)c.b[1], b2 = (double)c.b[2], a0 = (double)c.a[0], a1 = (double)c.a[1];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const double out = __builtin_fma(b0, x[ch], z0[ch]);
        z0[ch] = __builtin_fma(-a0, out, __builtin_fma(b1, x[ch], z1[ch]));
        z1[ch] = __builtin_fma(-a1, out, b2 * x[ch]);
        poison = __builtin_fma(out, 0.0, poison);
        x[ch] = out;
    }
}

struct Moments {
    double cross = 0.0, ll = 0.0, rr = 0.0;
    __device__ __forceinline__ void update(v2f y, double alpha) {  // Correlator::update (:40-46)
        const double l = (double)y.x, r = (double)y.y;
        cross += alpha * (l * r - cross);
        ll += alpha * (l * l - ll);
        rr += alpha * (r * r - rr);
    }
};

}  // namespace

// role 0: full-band correlator + low band (LP_low cascade); role 1: mid (HP_low, LP_high); role 2: high (HP_low, HP_high)
// RAGGED: the per-stream values of a ragged call are compiled in only there (as per-lane values they cost the lock-step kernel 20 %)
template <bool PASS_B, bool RAGGED>
__global__ __launch_bounds__(192) void stereo_chunk_kernel(StereoChunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [2][64][ROW_FLOATS]
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint64_t items = (uint64_t)a.n_streams * a.n_blocks;
    const uint64_t item0 = (uint64_t)blockIdx.x * 64u;
    const uint32_t L = a.block_frames, steps = L / STEP;

    // ---- loader: (row, part) pairs of a tile, 16 bytes (two frames) each; 512 pairs over blockDim.x threads
    const float* src[3];
    uint32_t dst[3];
    bool live[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const uint32_t q = tid + (uint32_t)n * blockDim.x;
        const uint32_t row = q >> 3, part = q & 7u;
        const uint64_t item = item0 + row;
        live[n] = q < 512u && item < items;
        const uint64_t s = live[n] ? item / a.n_blocks : 0, c = live[n] ? item % a.n_blocks : 0;
        if constexpr (RAGGED) live[n] = live[n] && c < a.blocks_v[s] * a.cpb;  // a stream's unused block slots hold anything
        src[n] = a.pcm + (s * a.frames_total + c * L) * 2u + part * 4u;
        dst[n] = row * ROW_FLOATS + part * 4u;
    }
    float4 pre[3];
    auto issue = [&](uint32_t step) {
#pragma unroll
        for (int n = 0; n < 3; ++n)
            pre[n] = live[n] ? *reinterpret_cast<const float4*>(src[n] + (uint64_t)step * (STEP * 2)) : float4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    uint32_t bad = 0;
    auto stage = [&](uint32_t buf) {
        float* t = tile + buf * (64 * ROW_FLOATS);
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            if ((uint32_t)n * blockDim.x + tid >= 512u) continue;
            const float4 p = pre[n];
            bad |= (!isfinite(p.x) || !isfinite(p.y) || !isfinite(p.z) || !isfinite(p.w)) ? 1u : 0u;
            // dsp.rs:232-239: left = (0.0 + s0 * w00) + s1 * w10 (two-channel fold, statement order kept)
            const v2f f0{0.0f + p.x * a.m00 + p.y * a.m10, 0.0f + p.x * a.m01 + p.y * a.m11};
            const v2f f1{0.0f + p.z * a.m00 + p.w * a.m10, 0.0f + p.z * a.m01 + p.w * a.m11};
            *reinterpret_cast<v2f*>(t + dst[n]) = f0;
            *reinterpret_cast<v2f*>(t + dst[n] + 2) = f1;
        }
    };

    // ---- per-lane recurrence state
    // ragged calls (omx_stereometer_bank_process_ragged): stream s runs blocks_v[s] of the call's n_blocks block slots; its history
    // positions come from the plan kernel (start_v), its reset flag acts in the scans
    const uint64_t item = item0 + lane;
    const bool in_call = item < items;
    const uint64_t s = in_call ? item / a.n_blocks : 0, c = in_call ? item % a.n_blocks : 0;
    uint32_t blocks_s = a.n_blocks;
    bool mine = in_call;
    if constexpr (RAGGED) {
        blocks_s = in_call ? a.blocks_v[s] * a.cpb : 0u;
        mine = c < blocks_s;
        if (__ballot(mine) == 0ull) return;  // (the three wavefronts of the workgroup see the same 64 items)
    }
    const bool bands = a.analyze_bands != 0;
    const BiquadCoef ca = role == 0 ? a.lp_lo : a.hp_lo, cb = role == 1 ? a.lp_hi : a.hp_hi;
    v2f z0[4], z1[4];  // role 0 uses elements 0, 1 (LP_low); roles 1, 2: 0, 1 = HP_low, 2, 3 = LP_high / HP_high
#pragma unroll
    for (int e = 0; e < 4; ++e) z0[e] = z1[e] = v2f{0.0f, 0.0f};
    float* cs = a.chunk_state + (item * 3u + role) * 16u;  // [8 states][2 channels] as 8 (L, R) pairs: z0[e], z1[e] interleaved
    // the low band's slot holds its 8 boundary states as f64 ([z0[e], z1[e]] x (L, R), the same order); the other bands' 16 as f32
    double* cs64 = reinterpret_cast<double*>(cs);
    if (PASS_B && mine && bands) {
        if (role == 0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                z0[e] = v2f{(float)cs64[4 * e], (float)cs64[4 * e + 1]};
                z1[e] = v2f{(float)cs64[4 * e + 2], (float)cs64[4 * e + 3]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z0[e] = *reinterpret_cast<const v2f*>(cs + 4 * e);
                z1[e] = *reinterpret_cast<const v2f*>(cs + 4 * e + 2);
            }
        }
    }
    double d0[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, d1[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, poison64 = 0.0;  // pass A, role 0: [section][channel]
    v2f poison{0.0f, 0.0f};
    Moments full, band;
    const double alpha = a.alpha;
    // history: the newest hist_frames pairs of the call (absolute position hist_pos + frame index, modulo the ring)
    const uint64_t total = (uint64_t)blocks_s * L;
    const uint64_t tail_from = total > a.hist_frames ? total - a.hist_frames : 0;
    const uint32_t band_id = role + 1u;
    float* hist_band = a.history + ((s * 4u + band_id) * (uint64_t)a.hist_frames) * 2u;
    float* hist_full = a.history + ((s * 4u) * (uint64_t)a.hist_frames) * 2u;
    const bool chunk_in_tail = PASS_B && mine && (c + 1u) * (uint64_t)L > tail_from;
    const bool wave_in_tail = __ballot(chunk_in_tail) != 0ull;
    uint64_t pos_full = a.hist_pos[0], pos_band = a.hist_pos[band_id];
    if constexpr (RAGGED) {
        pos_full = a.start_v[s * 4u];
        pos_band = a.start_v[s * 4u + band_id];
    }

    // The step loop once per band wavefront, the band a compile-time constant in each copy: tested at run time inside the loop, the
    // wave-uniform `role` cost scalar branches and register copies on every frame (waveform_chunked.hip has the census; round 6).
    auto run = [&](auto role_c) {
        constexpr uint32_t role = decltype(role_c)::value;
        issue(0);
        for (uint32_t step = 0; step < steps; ++step) {
            stage(step & 1u);
            if (step + 1u < steps) issue(step + 1u);
            __syncthreads();
            const float* row = tile + (step & 1u) * (64 * ROW_FLOATS) + lane * ROW_FLOATS;
            v2f x[STEP];
    #pragma unroll
            for (int f = 0; f < STEP; ++f) x[f] = *reinterpret_cast<const v2f*>(row + 2 * f);
            v2f y[STEP];
    #pragma unroll
            for (int f = 0; f < STEP; ++f) {
                v2f v = x[f];
                if (!PASS_B && role == 0) {   // (wave-uniform) the low band's zero-state pass in f64: only its end state is used
                    if (bands) {
                        double xd[2] = {(double)v.x, (double)v.y};
                        biquad_lr_f64(ca, d0[0], d1[0], xd, poison64);
                        biquad_lr_f64(ca, d0[1], d1[1], xd, poison64);
                    }
                } else if (bands) {
                    v = biquad_lr(ca, z0[0], z1[0], v, poison);
                    v = biquad_lr(ca, z0[1], z1[1], v, poison);
                    if (role != 0) {
                        v = biquad_lr(cb, z0[2], z1[2], v, poison);
                        v = biquad_lr(cb, z0[3], z1[3], v, poison);
                    }
                }
                y[f] = v;
                if constexpr (PASS_B) {
                    if (role == 0) full.update(x[f], alpha);
                    if (bands) band.update(v, alpha);
                }
            }
            if (PASS_B && wave_in_tail) {
    #pragma unroll
                for (int f = 0; f < STEP; ++f) {
                    const uint64_t g = c * (uint64_t)L + step * STEP + (uint32_t)f;
                    if (chunk_in_tail && g >= tail_from) {
                        if (role == 0) {
                            const uint64_t slot = (pos_full + g) % a.hist_frames;
                            *reinterpret_cast<v2f*>(hist_full + slot * 2u) = x[f];
                        }
                        if (bands && a.emit_band_points) {
                            const uint64_t slot = (pos_band + g) % a.hist_frames;
                            *reinterpret_cast<v2f*>(hist_band + slot * 2u) = y[f];
                        }
                    }
                }
            }
        }
    };
    if (role == 0u) run(std::integral_constant<uint32_t, 0>{});
    else if (role == 1u) run(std::integral_constant<uint32_t, 1>{});
    else run(std::integral_constant<uint32_t, 2>{});
    if (!(poison.x == 0.0f && poison.y == 0.0f) || !(poison64 == 0.0)) bad = 1u;
    if (__ballot(bad != 0u) != 0ull && lane == 0) atomicOr(a.bad, 1u);
    if (!mine) return;
    if constexpr (!PASS_B) {
        if (bands && role == 0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                cs64[4 * e] = d0[e][0];
                cs64[4 * e + 1] = d0[e][1];
                cs64[4 * e + 2] = d1[e][0];
                cs64[4 * e + 3] = d1[e][1];
            }
        } else if (bands) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<v2f*>(cs + 4 * e) = z0[e];
                *reinterpret_cast<v2f*>(cs + 4 * e + 2) = z1[e];
            }
        }
    } else {
        double* cm = a.chunk_moments + item * 12u;
        if (role == 0) {
            cm[0] = full.cross;
            cm[1] = full.ll;
            cm[2] = full.rr;
        }
        if (bands) {
            cm[3 * band_id + 0] = band.cross;
            cm[3 * band_id + 1] = band.ll;
            cm[3 * band_id + 2] = band.rr;
        }
    }
}

// ---- the two scans, wave-parallel: lane = block of the call, 64 blocks per sweep --------------------------------------------
// x_c <- x_c + T^d x_{c-d} for d = 1, 2, 4 ... 32 (Hillis-Steele over the affine maps s -> T s + e with a common T) leaves
// x_c = sum_{i <= c} T^(c-i) e_i; the state carried into the sweep enters as T carry added to the first lane.  The powers
// T^(2^k) come from the host (f64).  The once-per-block denormal flush of the reference (:134-140) is applied to the values
// that leave the scan: it only ever changes magnitudes below 1e-20 (f32 states) / 1e-30 (f64 moments).
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    const int lo = __shfl_up(__double2loint(v), d), hi = __shfl_up(__double2hiint(v), d);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    const int lo = __shfl(__double2loint(v), src), hi = __shfl(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// scan 1: wavefront = (stream, band 1..3, channel); states of one channel of one band: [z0[e], z1[e]] for e = 0..3 (band 1: e = 0, 1)
// F64IO (the low band): the zero-state end states arrive, and the true start states leave, as f64 in the same slots
template <int N, bool F64IO>
__device__ __forceinline__ void scan_states_wave(const StereoChunkArgs& a, const double* __restrict__ Tp /* [6][8][8] powers 1,2,..32 */,
                                                 uint32_t s, uint32_t r, uint32_t ch, uint32_t lane) {
    StereoLaneState& st = a.state[(uint64_t)s * 4u + r + 1u];
    const uint32_t nb = a.blocks_v ? a.blocks_v[s] * a.cpb : a.n_blocks;  // chunks of this stream
    const bool reset = a.reset_v != nullptr && a.reset_v[s] != 0;  // reset_audio (:92-97) of this stream before its blocks
    double carry[N];
#pragma unroll
    for (int k = 0; k < N; ++k) carry[k] = reset ? 0.0 : (double)st.z[k >> 2][(k >> 1) & 1][ch][k & 1];  // [stage][element][channel][z0 / z1]
    for (uint32_t c0 = 0; c0 < nb; c0 += 64u) {
        const uint32_t c = c0 + lane;
        const bool live = c < nb;
        float* cs = a.chunk_state + (((uint64_t)s * a.n_blocks + (live ? c : c0)) * 3u + r) * 16u;
        double* cs64 = reinterpret_cast<double*>(cs);
        double x[N];
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] = !live ? 0.0 : F64IO ? cs64[2 * k + ch] : (double)cs[2 * k + ch];
        if (lane == 0) {  // x_0 += T carry
#pragma unroll
            for (int k = 0; k < N; ++k) {
                double acc = x[k];
#pragma unroll
                for (int m = 0; m < N; ++m) acc += Tp[k * 8 + m] * carry[m];
                x[k] = acc;
            }
        }
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int d = 1 << step;
            const double* Td = Tp + step * 64;
            double up[N];
#pragma unroll
            for (int k = 0; k < N; ++k) up[k] = shfl_up_f64(x[k], d);
            if ((int)lane >= d) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    double acc = x[k];
#pragma unroll
                    for (int m = 0; m < N; ++m) acc += Td[k * 8 + m] * up[m];
                    x[k] = acc;
                }
            }
        }
        // x = state AFTER block c; the start state of block c is lane c - 1's (the carry for the first lane)
        const uint32_t last = min(nb - c0, 64u) - 1u;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double start = shfl_up_f64(x[k], 1);
            if (lane == 0) start = carry[k];
            const float v = (float)start;
            // (the flush belongs to block ends, :134-140: a chunk that starts inside a block takes its state unflushed)
            const bool at_block = c % a.cpb == 0u;
            const bool flush = at_block && fabsf(v) < 1.0e-20f;
            if (live) {  // the chunk's TRUE start state replaces its zero-state end state
                if constexpr (F64IO) cs64[2 * k + ch] = flush ? 0.0 : start;
                else cs[2 * k + ch] = flush ? 0.0f : v;
            }
            const double e64 = shfl_f64(x[k], (int)last);
            const float e = (float)e64;
            const bool flush_e = (c0 + last + 1u) % a.cpb == 0u && fabsf(e) < 1.0e-20f;
            // between the sweeps of one call the low band keeps its f64 trajectory; what leaves the call is the filters' f32 state
            carry[k] = flush_e ? 0.0 : (F64IO && c0 + 64u < nb) ? e64 : (double)e;
        }
    }
    if (lane == 0 && (nb != 0u || reset)) {
#pragma unroll
        for (int k = 0; k < N; ++k) st.z[k >> 2][(k >> 1) & 1][ch][k & 1] = (float)carry[k];
    }
}
__global__ __launch_bounds__(256) void stereo_scan_states_kernel(StereoChunkArgs a, const double* __restrict__ T /* [3][6][8][8] */) {
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (w >= a.n_streams * 6u) return;
    const uint32_t s = w / 6u, r = (w % 6u) >> 1, ch = w & 1u;  // r = band - 1
    if (r == 0) scan_states_wave<4, true>(a, T, s, r, ch, lane);
    else scan_states_wave<8, false>(a, T + r * 384, s, r, ch, lane);
}

// scan 2: wavefront = (stream, band 0..3); the three moments decay by (1 - alpha)^L per block
__global__ __launch_bounds__(256) void stereo_scan_moments_kernel(StereoChunkArgs a, double decay) {
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (w >= a.n_streams * 4u) return;
    const uint32_t s = w >> 2, b = w & 3u;
    const bool active = b == 0 || a.analyze_bands != 0;
    StereoLaneState& st = a.state[(uint64_t)s * 4u + b];
    const uint32_t nb = a.blocks_v ? a.blocks_v[s] * a.cpb : a.n_blocks;  // chunks of this stream
    const bool reset = a.reset_v != nullptr && a.reset_v[s] != 0;
    double carry[3] = {reset ? 0.0 : st.moments[0], reset ? 0.0 : st.moments[1], reset ? 0.0 : st.moments[2]};
    double dp[6];  // decay^(2^k)
    dp[0] = decay;
#pragma unroll
    for (int k = 1; k < 6; ++k) dp[k] = dp[k - 1] * dp[k - 1];
    for (uint32_t c0 = 0; c0 < nb; c0 += 64u) {
        const uint32_t c = c0 + lane;
        const bool live = c < nb;
        float value = 0.0f;
        if (active) {
            const double* cm = a.chunk_moments + ((uint64_t)s * a.n_blocks + (live ? c : c0)) * 12u + 3u * b;
            double x[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = live ? cm[k] : 0.0;
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) x[k] += decay * carry[k];
            }
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int d = 1 << step;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double up = shfl_up_f64(x[k], d);
                    if ((int)lane >= d) x[k] += dp[step] * up;
                }
            }
            const bool block_end = (c + 1u) % a.cpb == 0u;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (block_end && fabs(x[k]) < 1.0e-30) x[k] = 0.0;  // flush_denormals once per block (:134-136)
            const double denom = sqrt(x[1] * x[2]);  // Correlator::value (:48-56)
            if (denom > 1e-12) {
                const double v = x[0] / denom;
                if (isfinite(v)) value = (float)fmin(fmax(v, -1.0), 1.0);
            }
            const uint32_t last = min(nb - c0, 64u) - 1u;
#pragma unroll
            for (int k = 0; k < 3; ++k) carry[k] = shfl_f64(x[k], (int)last);
        }
        if (live && (c + 1u) % a.cpb == 0u) a.correlations[((uint64_t)s * (a.n_blocks / a.cpb) + c / a.cpb) * 4u + b] = value;
    }
    if (lane == 0 && (nb != 0u || reset)) {
        st.moments[0] = carry[0];
        st.moments[1] = carry[1];
        st.moments[2] = carry[2];
    }
}

void launch_stereometer_chunked(const StereoChunkArgs& a, const double* d_T, double decay, hipStream_t stream) {
    const uint64_t items = (uint64_t)a.n_streams * a.n_blocks;
    const uint32_t groups = (uint32_t)((items + 63) / 64);
    const uint32_t threads = 192u;  // without band analysis roles 1 and 2 only help staging the tiles
    const size_t lds = (size_t)2 * 64 * ROW_FLOATS * sizeof(float);
    const bool ragged = a.blocks_v != nullptr;
    if (a.analyze_bands) {
        if (ragged) hipLaunchKernelGGL((stereo_chunk_kernel<false, true>), dim3(groups), dim3(threads), lds, stream, a);
        else hipLaunchKernelGGL((stereo_chunk_kernel<false, false>), dim3(groups), dim3(threads), lds, stream, a);
        hipLaunchKernelGGL(stereo_scan_states_kernel, dim3((a.n_streams * 6u + 3u) / 4u), dim3(256), 0, stream, a, d_T);
    }
    if (ragged) hipLaunchKernelGGL((stereo_chunk_kernel<true, true>), dim3(groups), dim3(threads), lds, stream, a);
    else hipLaunchKernelGGL((stereo_chunk_kernel<true, false>), dim3(groups), dim3(threads), lds, stream, a);
    hipLaunchKernelGGL(stereo_scan_moments_kernel, dim3((a.n_streams * 4u + 3u) / 4u), dim3(256), 0, stream, a, decay);
}

}  // namespace omx
