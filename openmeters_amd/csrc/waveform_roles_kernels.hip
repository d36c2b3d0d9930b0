// Waveform kernel, role-per-wavefront form (reference src/visuals/waveform/processor.rs:92-121, :213-298,
// src/dsp.rs:264-371, :422-432, :489-495).  Same lanes as waveform_kernels.hip (16 per stream, lane = channel * 3 + band; four
// streams per workgroup), same operations on every value in the same order — the results are bit-identical — but the work of
// one lane is spread over the wavefronts of a workgroup, each with its own short instruction stream:
//
//   wavefront 0  memory     every global access of the batch loop: PCM -> LDS (kStage frames at a time), the expiring ring values
//                           of the next batch -> LDS, the previous batch's new ring values LDS -> rings
//   wavefront 1  front      stereo fold, band filters, channel value -> |v| gain and v^2 into an LDS batch; min/max column state
//                           machine, fractional column phase, min / max fields of the columns
//   wavefront 2  colour     the colour window of every lane: KBN pair, refresh, its field of the columns
//   wavefront 3, 4          the fast / slow RMS history windows (HISTORY only)
//
// Why: a lone wavefront that issues its own loads meets an `s_waitcnt vmcnt(0)` somewhere in every batch (the counter is in-order
// and the compiler's bookkeeping across the batch loop's control flow is conservative), i.e. one exposed memory round trip per
// eight frames on top of ~160 dependent VALU instructions per frame: 165 us for a 256-frame block.  Here the wavefronts that
// compute never wait for memory (their only global accesses are column stores), the memory wavefront's round trip overlaps the
// others' arithmetic, and the longest instruction stream per frame is the front's.  Rounds are separated by one LDS barrier.
#include <type_traits>

#include "waveform_device.hpp"

namespace omx {

using namespace wf;

namespace {
// Workgroup barrier for data handed over through LDS: waits for this wavefront's LDS traffic, not for its outstanding global
// stores / loads (__syncthreads would add vmcnt(0)).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr uint32_t kStage = 64;  // frames of PCM per LDS refill
}  // namespace

// B frames per straight-line batch, NSUB batches per round (R = B * NSUB frames between two barriers): the memory wavefront's round
// trip — about 2.5 us on an otherwise idle device — has to fit under the front's R frames
template <int B, int NSUB, bool HISTORY>
__global__ __launch_bounds__(HISTORY ? 320 : 192) void waveform_roles_kernel(WaveformArgs a) {
    constexpr int R = B * NSUB;
    constexpr int NW = HISTORY ? 3 : 1, NV = HISTORY ? 2 : 1;
    static_assert(kStage % R == 0, "a round never straddles two PCM refills");
    __shared__ float stage[2][4][kStage * OMX_MAX_CHANNELS];  // PCM, two refills in flight: [parity][stream of the group][frame][channel]
    __shared__ float olds[2][NW][R][64];                       // expiring values [round parity][window][frame][lane]
    __shared__ float vals[2][NV][R][64];                       // |v| gain, v^2 [round parity][which][frame][lane]
    __shared__ uint32_t flags[2][R];                           // 1 + kept column index when a column ends at the frame, else 0
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t gid = blockIdx.x * 64 + lane;  // stream * 16 + lane of the stream
    const uint32_t s = gid >> 4, ln = gid & 15;
    const bool in_bank = s < a.n_streams, live = in_bank && ln < 12;
    const uint32_t ch = ln / 3, band = ln % 3;
    const uint32_t row = a.n_streams * 16;
    const uint64_t n_rounds = (a.frames + R - 1) / R;
    const uint32_t tail = (uint32_t)(a.frames - (n_rounds - 1) * R);  // frames of the last round
    const uint64_t kept_cols = a.n_emit - a.first_kept;

    if (wave == 0) {
        // ---------------------------------------------------------------- memory wavefront
        const uint32_t column = in_bank ? gid : 0u;  // lanes past the last stream read column 0 (discarded) and store nothing
        float* cring = a.color_ring + column;
        float* hring = a.hist_ring + column;
        const float* pcm = a.pcm + (uint64_t)(in_bank ? s : 0) * a.frames * a.fmt.channels;
        uint32_t head_c = (uint32_t)(a.pushes % a.color_len), head_h = (uint32_t)(a.pushes % a.slow_len);  // slots of the batch being fetched
        uint32_t store_c = head_c, store_h = head_h;                                                        // ... and being stored
        auto refill = [&](uint64_t f) {  // kStage frames from f on -> stage[parity of the refill]
            float* dst = stage[(f / kStage) & 1][lane >> 4];
            const uint32_t n = (uint32_t)min((uint64_t)kStage, a.frames - f) * a.fmt.channels;
            const float* src = pcm + f * a.fmt.channels;
            const uint32_t l16 = lane & 15;
            for (uint32_t e0 = 0; e0 < n; e0 += 16 * 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = src[min(e0 + (uint32_t)j * 16 + l16, n - 1)];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (e0 + (uint32_t)j * 16 + l16 < n) dst[e0 + (uint32_t)j * 16 + l16] = v[j];
            }
        };
        if (a.frames) refill(0);
        lds_barrier();  // round -1
        for (uint64_t r = 0; r <= n_rounds; ++r) {
            if (r < n_rounds) {
                // the expiring values of round r, for the windows' round r + 1.  The slots are those of round r itself (colour,
                // slow: cap == ring length) or `color_len` pushes back (fast history): written at least two rounds ago, or —
                // a window that is not full yet — never used.  Every load is unconditional (always-valid slots).
                float oc[R], oh0[HISTORY ? R : 1], oh1[HISTORY ? R : 1];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    oc[k] = cring[(uint64_t)expiring_index(head_c, k, a.color_len, a.color_len) * row];
                    if constexpr (HISTORY) {
                        oh0[k] = hring[(uint64_t)expiring_index(head_h, k, a.slow_len, a.color_len) * row];
                        oh1[k] = hring[(uint64_t)expiring_index(head_h, k, a.slow_len, a.slow_len) * row];
                    }
                }
                head_c = (head_c + (uint32_t)R) % a.color_len;
                head_h = (head_h + (uint32_t)R) % a.slow_len;
                // PCM of the next refill, if round r + 1 is its first (the front reads the other parity meanwhile)
                const uint64_t f_next = (r + 1) * R;
                if (f_next % kStage == 0 && f_next < a.frames) refill(f_next);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    olds[r & 1][0][k][lane] = oc[k];
                    if constexpr (HISTORY) {
                        olds[r & 1][1][k][lane] = oh0[k];
                        olds[r & 1][2][k][lane] = oh1[k];
                    }
                }
            }
            if (r >= 1) {  // the ring values of round r - 1 (the front wrote them in round r - 1), after this round's loads
                const uint64_t b = r - 1;
                const uint32_t nr = b + 1 == n_rounds ? tail : (uint32_t)R;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    if ((uint32_t)k < nr && live) {
                        uint32_t slot = store_c + (uint32_t)k;
                        slot = slot >= a.color_len ? slot - a.color_len : slot;
                        cring[(uint64_t)slot * row] = vals[b & 1][0][k][lane];
                        if constexpr (HISTORY) {
                            uint32_t hs = store_h + (uint32_t)k;
                            hs = hs >= a.slow_len ? hs - a.slow_len : hs;
                            hring[(uint64_t)hs * row] = vals[b & 1][1][k][lane];
                        }
                    }
                }
                store_c = (store_c + nr) % a.color_len;
                store_h = (store_h + nr) % a.slow_len;
            }
            lds_barrier();
        }
        return;
    }

    if (wave == 1) {
        // ---------------------------------------------------------------- front wavefront
        float za[2][2] = {}, zb[2][2] = {};
        float cur_min = 0.0f, cur_max = 0.0f, cur_last = 0.0f, last_sample = 0.0f;
        uint32_t cur_some = 0, cur_has_last = 0, last_valid = 0;
        if (live) {
            const WaveLaneState& st = a.state[gid];
            za[0][0] = st.za[0][0]; za[0][1] = st.za[0][1]; za[1][0] = st.za[1][0]; za[1][1] = st.za[1][1];
            zb[0][0] = st.zb[0][0]; zb[0][1] = st.zb[0][1]; zb[1][0] = st.zb[1][0]; zb[1][1] = st.zb[1][1];
            cur_min = st.cur_min; cur_max = st.cur_max; cur_last = st.cur_last; last_sample = st.last_sample;
            cur_some = st.cur_some; cur_has_last = st.cur_has_last; last_valid = st.last_valid;
        }
        const BiquadCoef cb = band == 0 ? a.lp_lo : (band == 1 ? a.lp_hi : a.hp_hi);
        const bool use_a = band == 1;
        const float gain = band == 0 ? 1.0f : (band == 1 ? 0.7f : 2.0f);  // BAND_COLOR_GAINS (:22)
        const bool minmax_lane = live && band == 0;
        const bool two_channels = a.fmt.channels == 2;
        double phase = a.column_phase;
        uint64_t col = 0;
        auto write_minmax = [&](omx_wave_column* dst) {  // column_for (:213-235), the min / max fields
            float mn = 0.0f, mx = 0.0f;
            if (cur_some) {
                mn = cur_min;
                mx = cur_max;
                if (last_valid) {
                    mn = fminf(mn, last_sample);
                    mx = fmaxf(mx, last_sample);
                }
            }
            dst->min = mn;
            dst->max = mx;
        };
        lds_barrier();  // round -1: the first PCM refill
        for (uint64_t r = 0; r <= n_rounds; ++r) {
            const uint32_t nr = r < n_rounds ? (r + 1 == n_rounds ? tail : (uint32_t)R) : 0u;
#pragma unroll 1
            for (uint32_t k0 = 0; k0 < nr; k0 += B) {  // the round's batches
                const uint64_t f0 = r * R + k0;
                const uint32_t nb = min(nr - k0, (uint32_t)B);
                const float* chunk = stage[(f0 / kStage) & 1][lane >> 4] + (uint32_t)(f0 % kStage) * a.fmt.channels;
                float lr[B][2];
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const uint32_t kc = (uint32_t)k < nb ? (uint32_t)k : nb - 1u;
                    const float* frame = chunk + kc * a.fmt.channels;
                    float left = 0.0f, right = 0.0f;  // dsp.rs:223-249 stereo fold
                    if (two_channels) {  // uniform; the common shape without a runtime trip count
                        left = 0.0f + frame[0] * a.fmt.m[0][0] + frame[1] * a.fmt.m[1][0];
                        right = 0.0f + frame[0] * a.fmt.m[0][1] + frame[1] * a.fmt.m[1][1];
                    } else {
                        for (uint32_t c = 0; c < a.fmt.channels; ++c) {
                            const float v = frame[c];
                            left = left + v * a.fmt.m[c][0];
                            right = right + v * a.fmt.m[c][1];
                        }
                    }
                    lr[k][0] = left;
                    lr[k][1] = right;
                }
                auto samples = [&](auto tail_c, auto emit_c) {
                    constexpr bool TAIL = decltype(tail_c)::value, EMIT = decltype(emit_c)::value;
#pragma unroll
                    for (int k = 0; k < B; ++k) {
                        if constexpr (TAIL) {
                            if ((uint32_t)k >= nb) break;
                        }
                        const float left = lr[k][0], right = lr[k][1];
                        // derived_frame (:123-125): Left, Right, Mid, Side
                        const float derived = ch == 0 ? left : (ch == 1 ? right : (ch == 2 ? (left + right) * 0.5f : (left - right) * 0.5f));
                        const bool fin = isfinite(derived);
                        // :258-272 (non-live lanes compute on a neighbour's frames; nothing of theirs is stored)
                        float xl = isfinite(left) ? left : 0.0f, xr = isfinite(right) ? right : 0.0f;
                        // mid = LP_high(HP_low(x))  (CASCADE_HIGH = false: the high band takes the raw sample)
                        const float hl = biquad_step(a.hp_lo, za[0], xl), hr = biquad_step(a.hp_lo, za[1], xr);
                        xl = use_a ? hl : xl;
                        xr = use_a ? hr : xr;
                        const float bl = biquad_step(cb, zb[0], xl), br = biquad_step(cb, zb[1], xr);
                        float v = ch == 0 ? bl : (ch == 1 ? br : (ch == 2 ? (bl + br) * 0.5f : (bl - br) * 0.5f));
                        v = fin ? v : 0.0f;
                        // BandTracker::process (:108-121)
                        float cv = fabsf(v) * gain;
                        cv = isfinite(cv) ? cv : 0.0f;
                        vals[r & 1][0][k0 + k][lane] = cv;
                        if constexpr (HISTORY) {
                            float pw = v * v;
                            pw = isfinite(pw) ? pw : 0.0f;
                            vals[r & 1][1][k0 + k][lane] = pw;
                        }
                        // ingest_derived (:275-291), as selects
                        const bool some = cur_some != 0;
                        cur_min = fin ? (some ? fminf(cur_min, derived) : derived) : cur_min;
                        cur_max = fin ? (some ? fmaxf(cur_max, derived) : derived) : cur_max;
                        cur_last = fin ? derived : cur_last;
                        cur_has_last = fin ? 1 : (some ? 0 : cur_has_last);
                        cur_some = fin ? 1 : cur_some;
                        last_valid = fin ? last_valid : 0;
                        phase += a.step;
                        if constexpr (!EMIT) continue;  // the phase additions were replayed: no column ends in this batch
                        uint32_t flag = 0;
                        if (phase >= 1.0) {  // emit_column (:237-250); uniform over the workgroup
                            if (col >= a.first_kept) {
                                flag = (uint32_t)(col - a.first_kept) + 1u;
                                if (minmax_lane) write_minmax(a.columns + ((uint64_t)s * kept_cols + (col - a.first_kept)) * 4 + ch);
                            }
                            if (cur_some && cur_has_last) {
                                last_valid = 1;
                                last_sample = cur_last;
                            }
                            cur_some = 0;
                            cur_has_last = 0;
                            ++col;
                            phase -= 1.0;
                        }
                        if (lane == 0) flags[r & 1][k0 + k] = flag;
                    }
                };
                // does a column end inside this batch?  Replay the f64 phase additions (they are the reference's, bit for bit)
                bool emits = false;
                {
                    double ph = phase;
#pragma unroll
                    for (int k = 0; k < B; ++k) {
                        ph += a.step;
                        emits = emits || ph >= 1.0;
                    }
                }
                using T = std::true_type;
                using F = std::false_type;
                if (nb == (uint32_t)B && !emits) {
                    if (lane < (uint32_t)B) flags[r & 1][k0 + lane] = 0u;
                    samples(F{}, F{});  // the straight-line batch
                } else {
                    samples(T{}, T{});
                }
            }
            lds_barrier();
        }
        // BandFilter::flush_denormals once per block (:321-323)
        float* z[8] = {&za[0][0], &za[0][1], &za[1][0], &za[1][1], &zb[0][0], &zb[0][1], &zb[1][0], &zb[1][1]};
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (fabsf(*z[i]) < 1.0e-20f) *z[i] = 0.0f;
        if (minmax_lane && a.write_preview) write_minmax(a.preview + (uint64_t)s * 4 + ch);  // preview (:300-306)
        if (live) {
            WaveLaneState& st = a.state[gid];
            st.za[0][0] = za[0][0]; st.za[0][1] = za[0][1]; st.za[1][0] = za[1][0]; st.za[1][1] = za[1][1];
            st.zb[0][0] = zb[0][0]; st.zb[0][1] = zb[0][1]; st.zb[1][0] = zb[1][0]; st.zb[1][1] = zb[1][1];
            st.cur_min = cur_min; st.cur_max = cur_max; st.cur_last = cur_last; st.last_sample = last_sample;
            st.cur_some = cur_some; st.cur_has_last = cur_has_last; st.last_valid = last_valid;
        }
        return;
    }

    // -------------------------------------------------------------------- window wavefronts: 2 colour, 3 fast history, 4 slow history
    const uint32_t role = wave - 2;                                 // wave-uniform
    const uint32_t cap = role == 2 ? a.slow_len : a.color_len;      // colour, fast: color_len; slow: slow_len
    const uint32_t mean_len = role == 0 ? a.color_len : a.slow_len; // ring length of the reference's WindowedMeans (dsp.rs:367-370)
    const uint32_t which = role == 0 ? 0u : 1u;                     // |v| gain for the colour window, v^2 for the histories
    Window w;
    {
        double init[4] = {0.0, 0.0, 0.0, 0.0};
        if (live) {
            const double* src = role == 0 ? a.state[gid].color : a.state[gid].hist[role - 1];
            init[0] = src[0]; init[1] = src[1]; init[2] = src[2]; init[3] = src[3];
        }
        w.init(init, cap, a.pushes);
    }
    uint64_t pushes = a.pushes;
    auto write_field = [&](omx_wave_column* dst) {  // column_for (:213-235), this window's field
        const double m = fmax(w.mean(pushes, mean_len), 0.0);
        if (role == 0) {
            dst->color_bands[band] = (float)m;
            if constexpr (!HISTORY) {
                dst->rms_db[0][band] = -140.0f;
                dst->rms_db[1][band] = -140.0f;
            }
        } else {
            dst->rms_db[role - 1][band] = power_to_db_f((float)m, -140.0f);
        }
    };
    lds_barrier();  // round -1
    for (uint64_t r = 0; r <= n_rounds; ++r) {
        const uint64_t b = r - 1;  // the round being consumed
        const uint32_t nr = r >= 1 ? (b + 1 == n_rounds ? tail : (uint32_t)R) : 0u;
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < nr; k0 += B) {  // the round's batches
            const uint32_t nb = min(nr - k0, (uint32_t)B);
            float v[B], old[B];
            uint32_t any_flag = 0;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                v[k] = vals[b & 1][which][k0 + k][lane];
                old[k] = olds[b & 1][role][k0 + k][lane];
                any_flag |= flags[b & 1][k0 + k];
            }
            any_flag = (uint32_t)__builtin_amdgcn_readfirstlane((int)any_flag);
            const uint32_t unf = w.unfilled;  // samples of this batch that precede the window's first expiring value (dsp.rs:336-338)
            auto consume = [&](auto check_c, auto tail_c) {
                constexpr bool CHECK = decltype(check_c)::value, TAIL = decltype(tail_c)::value;
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    if constexpr (TAIL) {
                        if ((uint32_t)k >= nb) break;
                    }
                    w.template push<CHECK>((double)v[k], (uint32_t)k >= unf ? (double)old[k] : 0.0);
                    ++pushes;
                    if constexpr (TAIL) {  // also the batches in which a column ends
                        const uint32_t flag = (uint32_t)__builtin_amdgcn_readfirstlane((int)flags[b & 1][k0 + k]);
                        if (flag && live) write_field(a.columns + ((uint64_t)s * kept_cols + (flag - 1u)) * 4 + ch);
                    }
                }
            };
            using T = std::true_type;
            using F = std::false_type;
            const bool may_refresh = w.refresh + (uint32_t)B >= w.cap;
            if (nb == (uint32_t)B && !any_flag && !may_refresh) consume(F{}, F{});  // the straight-line batch
            else if (nb == (uint32_t)B && !any_flag) consume(T{}, F{});
            else consume(T{}, T{});
        }
        lds_barrier();
    }
    if (live && a.write_preview) write_field(a.preview + (uint64_t)s * 4 + ch);  // preview (:300-306)
    if (live) {
        double out[4];
        w.save(out);
        double* dst = role == 0 ? a.state[gid].color : a.state[gid].hist[role - 1];
        dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
    }
}

// true when the role form applies: band analysis on, windows long enough for the two-round distance between a ring store and the
// load of the same slot (see the memory wavefront)
bool waveform_roles_applicable(const WaveformArgs& a) { return a.analyze != 0 && a.color_len >= 64 && a.slow_len >= 64; }  // 2 R

void launch_waveform_roles(const WaveformArgs& a, hipStream_t stream) {
    const uint32_t threads = a.n_streams * 16;
    const dim3 grid((threads + 63) / 64);
    if (a.track_history) hipLaunchKernelGGL((waveform_roles_kernel<8, 4, true>), grid, dim3(320), 0, stream, a);
    else hipLaunchKernelGGL((waveform_roles_kernel<8, 4, false>), grid, dim3(192), 0, stream, a);
}

}  // namespace omx
