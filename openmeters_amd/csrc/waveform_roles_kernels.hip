// Waveform kernel, role-per-wavefront form (reference src/visuals/waveform/processor.rs:92-121, :213-298,
// src/dsp.rs:264-371, :422-432, :489-495).  Same lanes as waveform_kernels.hip (16 per stream, lane = channel * 3 + band; four
// streams per workgroup), same operations on every value in the same order — the results are bit-identical — but the work of
// one lane is spread over the wavefronts of a workgroup, each with its own short instruction stream:
//
//   wavefront 0     memory   every global access of the frame loop: PCM -> LDS (kStage frames at a time), the expiring ring values of a
//                           round -> LDS, the new ring values of an earlier round LDS -> rings
//   wavefront 1, 2  filters  the left / right half of the band split: stereo fold of its side, HP_low, band filter -> LDS
//   wavefront 3     mix      one round behind the filters: channel value from the two filtered sides -> |v| gain and v^2 into LDS;
//                           min/max column state machine, fractional column phase, min / max fields of the columns
//   wavefront 4     colour   two rounds behind: the colour window of every lane (KBN pair, refresh), its field of the columns
//   wavefront 5, 6           the fast / slow RMS history windows (HISTORY only)
//
// Why: a lone wavefront that issues its own loads meets an `s_waitcnt vmcnt(0)` somewhere in every batch (the counter is in-order
// and the compiler's bookkeeping across the batch loop's control flow is conservative), i.e. one exposed memory round trip per
// eight frames, on top of ~160 mostly dependent VALU instructions per frame: 165 us for a 256-frame block.  Here the wavefronts
// that compute never wait for memory (their only global accesses are column stores), the memory wavefront's round trip overlaps
// the others' arithmetic, and the longest instruction stream per frame is 35 ... 45 instructions.  Rounds (R frames) are
// separated by one LDS barrier; data of round d is filtered in round d, mixed in d + 1, windowed and stored in d + 2.
#include <type_traits>

#include "waveform_device.hpp"

namespace omx {

using namespace wf;

namespace {
// Workgroup barrier for data handed over through LDS: waits for this wavefront's LDS traffic, not for its outstanding global
// stores / loads (__syncthreads would add vmcnt(0)).
#ifdef OMX_WF_PHASES
// per-wavefront timing of the rounds (build this file with -DOMX_WF_PHASES; the latency harness then prints one line per wavefront)
#define WF_T0 const long long wf_t0 = clock64(); long long wf_wait = 0;
__device__ __forceinline__ void lds_barrier_timed(long long& acc) {
    const long long t = clock64();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    acc += clock64() - t;
}
#define lds_barrier() lds_barrier_timed(wf_wait)
#define WF_REPORT(name) if (lane == 0 && blockIdx.x == 0 && a.pushes == 256 * 300) printf("%s wave %u: total %lld barrier-wait %lld\n", name, wave, (long long)(clock64() - wf_t0), wf_wait);
#else
#define WF_T0
#define WF_REPORT(name)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
constexpr uint32_t kStage = 64;  // frames of PCM per LDS refill
}  // namespace

// B frames per straight-line batch, NSUB batches per round (R = B * NSUB frames between two barriers): the memory wavefront's round
// trip — about 2.5 us on an otherwise idle device — has to fit under the front's R frames
// RAGGED (omx_waveform_bank_process_ragged; registry.rs:396-418 feeds every capture on its own): the four streams of a workgroup bring
// their own frame count, push count, column phase and reset flag.  The round loop runs to the longest of the four; ring positions,
// the column phase and the column flags become per-lane / per-stream values (a conditional subtract where the lock-step kernel has
// a scalar modulo), and the batches in which a stream has already ended take the masked (tail) form of the batch body.  With equal
// frame counts every batch is the straight-line one, as in the lock-step kernel.
template <int B, int NSUB, bool HISTORY, bool RAGGED>
__global__ __launch_bounds__(HISTORY ? 448 : 320) void waveform_roles_kernel(WaveformArgs a) {
    if (a.run_if && *a.run_if == 0u) return;  // fallback launch of the chunk-parallel path: the PCM was finite (workgroup-uniform)
    constexpr int R = B * NSUB;
    constexpr int NW = HISTORY ? 3 : 1, NV = HISTORY ? 2 : 1;
    static_assert(kStage % R == 0, "a round never straddles two PCM refills");
    __shared__ float stage[2][4][kStage * OMX_MAX_CHANNELS];  // PCM, two refills in flight: [parity][stream of the group][frame][channel]
    __shared__ float olds[2][NW][R][64];                       // expiring values [round parity][window][frame][lane]
    __shared__ float vals[2][NV][R][64];                       // |v| gain, v^2 [round parity][which][frame][lane]
    __shared__ float sides[2][2][R][64];                       // band-filtered left, right [round parity][side][frame][lane]
    __shared__ float folded[2][2][R][64];                      // the folded left, right themselves (the mix derives the channel value)
    __shared__ uint32_t flags[2][R][RAGGED ? 4 : 1];           // 1 + kept column index when a column ends at the frame, else 0 (per stream when ragged)
    __shared__ uint32_t preview_on[4];                         // ragged: the stream's preview column exists (progress > 0)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    WF_T0
    const uint32_t gid = blockIdx.x * 64 + lane;  // stream * 16 + lane of the stream
    const uint32_t s = gid >> 4, ln = gid & 15;
    const bool in_bank = s < a.n_streams, live = in_bank && ln < 12;
    const uint32_t ch = ln / 3, band = ln % 3;
    const uint32_t row = a.n_streams * 16;
    const uint32_t grp = lane >> 4;  // stream of the workgroup
    // per-stream call values (ragged) or the bank's common ones
    bool reset_s = false;
    uint64_t frames_s = a.frames, pushes0 = a.pushes, frames_max = a.frames;
    double phase0 = a.column_phase;
    if constexpr (RAGGED) {
        reset_s = in_bank && a.reset_v != nullptr && a.reset_v[s] != 0;
        frames_s = in_bank ? a.frames_v[s] : 0u;
        pushes0 = (in_bank && !reset_s) ? a.pushes_v[s] : 0ull;
        phase0 = (in_bank && !reset_s) ? a.phase_v[s] : 0.0;
        const uint32_t f = (uint32_t)frames_s;
        const uint32_t m = max(max((uint32_t)__shfl((int)f, 0), (uint32_t)__shfl((int)f, 16)), max((uint32_t)__shfl((int)f, 32), (uint32_t)__shfl((int)f, 48)));
        frames_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
    }
    const uint64_t n_rounds = (frames_max + R - 1) / R;
    const uint32_t tail = (uint32_t)(frames_max - (n_rounds - 1) * R);  // frames of the last round
    const uint64_t kept_cols = RAGGED ? a.max_cols : a.n_emit - a.first_kept;
    const uint64_t last_iter = n_rounds + 1;  // rounds 0 ... n_rounds + 1: the windows run two behind the filters
    auto frames_of = [&](uint64_t d) { return d + 1 == n_rounds ? tail : (uint32_t)R; };
    // this lane's stream: frames of round d / of the batch starting at frame f0 (ragged: 0 once the stream has ended)
    auto frames_of_lane = [&](uint64_t d) -> uint32_t {
        if constexpr (!RAGGED) return frames_of(d);
        const uint64_t f0 = d * R;
        return frames_s > f0 ? (uint32_t)min(frames_s - f0, (uint64_t)R) : 0u;
    };
    auto batch_of_lane = [&](uint64_t f0, uint32_t nb) -> uint32_t {
        if constexpr (!RAGGED) return nb;
        return frames_s > f0 ? (uint32_t)min(frames_s - f0, (uint64_t)B) : 0u;
    };

    if (wave == 0) {
        // ---------------------------------------------------------------- memory wavefront
        const uint32_t column = in_bank ? gid : 0u;  // lanes past the last stream read column 0 (discarded) and store nothing
        float* cring = a.color_ring + column;
        float* hring = a.hist_ring + column;
        const float* pcm = a.pcm + (uint64_t)(in_bank ? s : 0) * a.frames * a.fmt.channels;  // (ragged: a.frames = the row stride)
        uint32_t head_c = (uint32_t)(pushes0 % a.color_len), head_h = (uint32_t)(pushes0 % a.slow_len);  // slots of the batch being fetched
        uint32_t store_c = head_c, store_h = head_h;                                                        // ... and being stored
        auto refill = [&](uint64_t f) {  // kStage frames from f on -> stage[parity of the refill]
            float* dst = stage[(f / kStage) & 1][lane >> 4];
            const uint32_t n = frames_s > f ? (uint32_t)min((uint64_t)kStage, frames_s - f) * a.fmt.channels : 0u;
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
        if (frames_max) refill(0);
        lds_barrier();  // round -1
        for (uint64_t r = 0; r <= last_iter; ++r) {
            if (r >= 1 && r <= n_rounds) {
                // the expiring values of data round d = r - 1, for the windows' round r + 1.  The slots are those of round d itself
                // (colour, slow: cap == ring length) or `color_len` pushes back (fast history): stored in an earlier round than
                // this one (cap >= 2 R), or — a window that is not full yet — never used.  Every load is unconditional.
                const uint64_t d = r - 1;
                float oc[R], oh0[HISTORY ? R : 1], oh1[HISTORY ? R : 1];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    oc[k] = cring[(uint64_t)expiring_index(head_c, k, a.color_len, a.color_len) * row];
                    if constexpr (HISTORY) {
                        oh0[k] = hring[(uint64_t)expiring_index(head_h, k, a.slow_len, a.color_len) * row];
                        oh1[k] = hring[(uint64_t)expiring_index(head_h, k, a.slow_len, a.slow_len) * row];
                    }
                }
                if constexpr (RAGGED) {  // per-lane positions: R <= len / 2
                    head_c += (uint32_t)R;
                    head_c = head_c >= a.color_len ? head_c - a.color_len : head_c;
                    head_h += (uint32_t)R;
                    head_h = head_h >= a.slow_len ? head_h - a.slow_len : head_h;
                } else {
                    head_c = (head_c + (uint32_t)R) % a.color_len;
                    head_h = (head_h + (uint32_t)R) % a.slow_len;
                }
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    olds[d & 1][0][k][lane] = oc[k];
                    if constexpr (HISTORY) {
                        olds[d & 1][1][k][lane] = oh0[k];
                        olds[d & 1][2][k][lane] = oh1[k];
                    }
                }
            }
            {  // PCM of the next refill, if round r + 1 is its first (filters and mix read the other parity meanwhile)
                const uint64_t f_next = (r + 1) * R;
                if (f_next % kStage == 0 && f_next < frames_max) refill(f_next);
            }
            if (r >= 2) {  // the ring values of data round r - 2 (the mix wrote them in round r - 1), after this round's loads
                const uint64_t b = r - 2;
                const uint32_t nr = frames_of_lane(b);
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
                if constexpr (RAGGED) {
                    store_c += nr;
                    store_c = store_c >= a.color_len ? store_c - a.color_len : store_c;
                    store_h += nr;
                    store_h = store_h >= a.slow_len ? store_h - a.slow_len : store_h;
                } else {
                    store_c = (store_c + nr) % a.color_len;
                    store_h = (store_h + nr) % a.slow_len;
                }
            }
            lds_barrier();
        }
        WF_REPORT("memory")
        return;
    }

    if (wave == 1 || wave == 2) {
        // ---------------------------------------------------------------- filter wavefronts: left (1), right (2)
        const uint32_t side = wave - 1;  // wave-uniform
        float za[2] = {0.0f, 0.0f}, zb[2] = {0.0f, 0.0f};
        if (live && !reset_s) {
            const WaveLaneState& st = a.state[gid];
            za[0] = st.za[side][0]; za[1] = st.za[side][1];
            zb[0] = st.zb[side][0]; zb[1] = st.zb[side][1];
        }
        const BiquadCoef cb = band == 0 ? a.lp_lo : (band == 1 ? a.lp_hi : a.hp_hi);
        const bool use_a = band == 1;
        const bool two_channels = a.fmt.channels == 2;
        const float m0 = a.fmt.m[0][side], m1 = a.fmt.m[1][side];
        lds_barrier();  // round -1: the first PCM refill
        for (uint64_t r = 0; r <= last_iter; ++r) {
            const uint32_t nr = r < n_rounds ? frames_of(r) : 0u;
#pragma unroll 1
            for (uint32_t k0 = 0; k0 < nr; k0 += B) {  // the round's batches
                const uint64_t f0 = r * R + k0;
                const uint32_t nb = min(nr - k0, (uint32_t)B);
                const uint32_t nb_lane = batch_of_lane(f0, nb);
                const float* chunk = stage[(f0 / kStage) & 1][lane >> 4] + (uint32_t)(f0 % kStage) * a.fmt.channels;
                float x[B];
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const uint32_t kc = (uint32_t)k < nb ? (uint32_t)k : nb - 1u;
                    const float* frame = chunk + kc * a.fmt.channels;
                    float folded = 0.0f;  // dsp.rs:223-249 stereo fold, this side's column of the matrix
                    if (two_channels) {  // uniform; the common shape without a runtime trip count
                        folded = 0.0f + frame[0] * m0 + frame[1] * m1;
                    } else {
                        for (uint32_t c = 0; c < a.fmt.channels; ++c) folded = folded + frame[c] * a.fmt.m[c][side];
                    }
                    x[k] = folded;
                }
                auto samples = [&](auto tail_c) {
                    constexpr bool TAIL = decltype(tail_c)::value;
#pragma unroll
                    for (int k = 0; k < B; ++k) {
                        if constexpr (TAIL) {
                            if ((uint32_t)k >= nb) break;
                            if (RAGGED && (uint32_t)k >= nb_lane) continue;  // this lane's stream has ended
                        }
                        // :258-272 (non-live lanes compute on a neighbour's frames; nothing of theirs is stored)
                        folded[r & 1][side][k0 + k][lane] = x[k];
                        float xs = isfinite(x[k]) ? x[k] : 0.0f;
                        // mid = LP_high(HP_low(x))  (CASCADE_HIGH = false: the high band takes the raw sample)
                        const float h = biquad_step(a.hp_lo, za, xs);
                        xs = use_a ? h : xs;
                        sides[r & 1][side][k0 + k][lane] = biquad_step(cb, zb, xs);
                    }
                };
                if (nb == (uint32_t)B && (!RAGGED || __all(nb_lane == (uint32_t)B))) samples(std::false_type{});
                else samples(std::true_type{});
            }
            lds_barrier();
        }
        WF_REPORT("filter")
        // BandFilter::flush_denormals once per block (:321-323)
        if (fabsf(za[0]) < 1.0e-20f) za[0] = 0.0f;
        if (fabsf(za[1]) < 1.0e-20f) za[1] = 0.0f;
        if (fabsf(zb[0]) < 1.0e-20f) zb[0] = 0.0f;
        if (fabsf(zb[1]) < 1.0e-20f) zb[1] = 0.0f;
        if (live) {
            WaveLaneState& st = a.state[gid];
            st.za[side][0] = za[0]; st.za[side][1] = za[1];
            st.zb[side][0] = zb[0]; st.zb[side][1] = zb[1];
        }
        return;
    }

    if (wave == 3) {
        // ---------------------------------------------------------------- mix wavefront (data round d = r - 1)
        float cur_min = 0.0f, cur_max = 0.0f, cur_last = 0.0f, last_sample = 0.0f;
        uint32_t cur_some = 0, cur_has_last = 0, last_valid = 0;
        if (live && !reset_s) {
            const WaveLaneState& st = a.state[gid];
            cur_min = st.cur_min; cur_max = st.cur_max; cur_last = st.cur_last; last_sample = st.last_sample;
            cur_some = st.cur_some; cur_has_last = st.cur_has_last; last_valid = st.last_valid;
        }
        const float gain = band == 0 ? 1.0f : (band == 1 ? 0.7f : 2.0f);  // BAND_COLOR_GAINS (:22)
        const ChannelPick pick(ch);
        const bool minmax_lane = live && band == 0;
        double phase = phase0;
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
        lds_barrier();  // round -1
        for (uint64_t r = 0; r <= last_iter; ++r) {
            const uint64_t d = r - 1;
            const uint32_t nr = (r >= 1 && d < n_rounds) ? frames_of(d) : 0u;
#pragma unroll 1
            for (uint32_t k0 = 0; k0 < nr; k0 += B) {  // the round's batches
                const uint32_t nb = min(nr - k0, (uint32_t)B);
                const uint32_t nb_lane = batch_of_lane(d * R + k0, nb);
                float lr[B][2], fl[B], fr[B];
#pragma unroll
                for (int k = 0; k < B; ++k) {  // (frames past a short batch's end: stale LDS, never used)
                    lr[k][0] = folded[d & 1][0][k0 + k][lane];
                    lr[k][1] = folded[d & 1][1][k0 + k][lane];
                    fl[k] = sides[d & 1][0][k0 + k][lane];
                    fr[k] = sides[d & 1][1][k0 + k][lane];
                }
                auto samples = [&](auto tail_c, auto emit_c) {
                    constexpr bool TAIL = decltype(tail_c)::value, EMIT = decltype(emit_c)::value;
#pragma unroll
                    for (int k = 0; k < B; ++k) {
                        if constexpr (TAIL) {
                            if ((uint32_t)k >= nb) break;
                            if (RAGGED && (uint32_t)k >= nb_lane) {  // this lane's stream has ended: no column ends here
                                if ((lane & 15u) == 0u) flags[d & 1][k0 + k][grp] = 0u;
                                continue;
                            }
                        }
                        const float left = lr[k][0], right = lr[k][1];
                        // derived_frame (:123-125): Left, Right, Mid, Side
                        const float derived = pick(left, right);
                        const bool fin = isfinite(derived);
                        const float bl = fl[k], br = fr[k];
                        float v = pick(bl, br);
                        v = fin ? v : 0.0f;
                        // BandTracker::process (:108-121)
                        float cv = fabsf(v) * gain;
                        cv = isfinite(cv) ? cv : 0.0f;
                        vals[d & 1][0][k0 + k][lane] = cv;
                        if constexpr (HISTORY) {
                            float pw = v * v;
                            pw = isfinite(pw) ? pw : 0.0f;
                            vals[d & 1][1][k0 + k][lane] = pw;
                        }
                        // ingest_derived (:275-291), as selects
                        const bool some = cur_some != 0;
                        cur_min = fin ? (some ? min_finite(cur_min, derived) : derived) : cur_min;
                        cur_max = fin ? (some ? max_finite(cur_max, derived) : derived) : cur_max;
                        cur_last = fin ? derived : cur_last;
                        cur_has_last = fin ? 1 : (some ? 0 : cur_has_last);
                        cur_some = fin ? 1 : cur_some;
                        last_valid = fin ? last_valid : 0;
                        phase += a.step;
                        if constexpr (!EMIT) continue;  // the phase additions were replayed: no column ends in this batch
                        uint32_t flag = 0;
                        if (phase >= 1.0) {  // emit_column (:237-250); uniform over the workgroup (ragged: over the stream's lanes)
                            if constexpr (RAGGED) {
                                if (col < a.max_cols) {
                                    flag = (uint32_t)col + 1u;
                                    if (minmax_lane) write_minmax(a.columns + ((uint64_t)s * kept_cols + col) * 4 + ch);
                                }
                            } else if (col >= a.first_kept) {
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
                        if constexpr (RAGGED) {
                            if ((lane & 15u) == 0u) flags[d & 1][k0 + k][grp] = flag;
                        } else {
                            if (lane == 0) flags[d & 1][k0 + k][0] = flag;
                        }
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
                if (nb == (uint32_t)B && (RAGGED ? !__any(emits) && __all(nb_lane == (uint32_t)B) : !emits)) {
                    constexpr uint32_t per = RAGGED ? 4u : 1u;
                    if (lane < (uint32_t)B * per) (&flags[d & 1][k0][0])[lane] = 0u;
                    samples(F{}, F{});  // the straight-line batch
                } else {
                    samples(T{}, T{});
                }
            }
            if constexpr (RAGGED) {  // (the last iteration carries no data: the window wavefronts read this after its barrier)
                if (r == last_iter && (lane & 15u) == 0u) preview_on[grp] = fmin(fmax(phase, 0.0), 1.0) > 0.0 ? 1u : 0u;
            }
            lds_barrier();
        }
        WF_REPORT("mix")
        const float progress = (float)fmin(fmax(phase, 0.0), 1.0);  // preview (:300-306)
        if (minmax_lane && (RAGGED ? progress > 0.0f : a.write_preview != 0)) write_minmax(a.preview + (uint64_t)s * 4 + ch);
        if constexpr (RAGGED) {
            if (in_bank && (lane & 15u) == 0u) {  // the stream's counters for its next call (every wavefront read them before round -1)
                a.pushes_v[s] = pushes0 + frames_s;
                a.phase_v[s] = phase;
                a.cols_v[s] = (uint32_t)min(col, a.max_cols);
                a.progress_v[s] = progress;
            }
        }
        if (live) {
            WaveLaneState& st = a.state[gid];
            st.cur_min = cur_min; st.cur_max = cur_max; st.cur_last = cur_last; st.last_sample = last_sample;
            st.cur_some = cur_some; st.cur_has_last = cur_has_last; st.last_valid = last_valid;
        }
        return;
    }

    // -------------------------------------------------------------------- window wavefronts: 4 colour, 5 fast history, 6 slow history (data round d = r - 2)
    const uint32_t role = wave - 4;                                 // wave-uniform
    const uint32_t cap = role == 2 ? a.slow_len : a.color_len;      // colour, fast: color_len; slow: slow_len
    const uint32_t mean_len = role == 0 ? a.color_len : a.slow_len; // ring length of the reference's WindowedMeans (dsp.rs:367-370)
    const uint32_t which = role == 0 ? 0u : 1u;                     // |v| gain for the colour window, v^2 for the histories
    Window w;
    {
        double init[4] = {0.0, 0.0, 0.0, 0.0};
        if (live && !reset_s) {
            const double* src = role == 0 ? a.state[gid].color : a.state[gid].hist[role - 1];
            init[0] = src[0]; init[1] = src[1]; init[2] = src[2]; init[3] = src[3];
        }
        w.init(init, cap, pushes0);
    }
    uint64_t pushes = pushes0;
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
    for (uint64_t r = 0; r <= last_iter; ++r) {
        const uint64_t b = r - 2;  // the data round being consumed
        const uint32_t nr = r >= 2 ? frames_of(b) : 0u;
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < nr; k0 += B) {  // the round's batches
            const uint32_t nb = min(nr - k0, (uint32_t)B);
            const uint32_t nb_lane = batch_of_lane(b * R + k0, nb);
            float v[B], old[B];
            uint32_t any_flag = 0;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                v[k] = vals[b & 1][which][k0 + k][lane];
                old[k] = olds[b & 1][role][k0 + k][lane];
                any_flag |= flags[b & 1][k0 + k][RAGGED ? grp : 0u];
            }
            if constexpr (RAGGED) any_flag = (__any(any_flag != 0u) || !__all(nb_lane == (uint32_t)B)) ? 1u : 0u;
            else any_flag = (uint32_t)__builtin_amdgcn_readfirstlane((int)any_flag);
            const uint32_t unf = w.unfilled;  // samples of this batch that precede the window's first expiring value (dsp.rs:336-338)
            auto consume = [&](auto check_c, auto tail_c) {
                constexpr bool CHECK = decltype(check_c)::value, TAIL = decltype(tail_c)::value;
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    if constexpr (TAIL) {
                        if ((uint32_t)k >= nb) break;
                        if (RAGGED && (uint32_t)k >= nb_lane) continue;  // this lane's stream has ended
                    }
                    w.template push<CHECK>((double)v[k], (uint32_t)k >= unf ? (double)old[k] : 0.0);
                    ++pushes;
                    if constexpr (TAIL) {  // also the batches in which a column ends
                        uint32_t flag;
                        if constexpr (RAGGED) flag = flags[b & 1][k0 + k][grp];
                        else flag = (uint32_t)__builtin_amdgcn_readfirstlane((int)flags[b & 1][k0 + k][0]);
                        if (flag && live) write_field(a.columns + ((uint64_t)s * kept_cols + (flag - 1u)) * 4 + ch);
                    }
                }
            };
            using T = std::true_type;
            using F = std::false_type;
            const bool may_refresh = RAGGED ? __any(w.refresh + (uint32_t)B >= w.cap) : w.refresh + (uint32_t)B >= w.cap;
            if (nb == (uint32_t)B && !any_flag && !may_refresh) consume(F{}, F{});  // the straight-line batch
            else if (nb == (uint32_t)B && !any_flag) consume(T{}, F{});
            else consume(T{}, T{});
        }
        lds_barrier();
    }
    WF_REPORT("window")
    if (live && (RAGGED ? preview_on[grp] != 0u : a.write_preview != 0)) write_field(a.preview + (uint64_t)s * 4 + ch);  // preview (:300-306)
    if (live) {
        double out[4];
        w.save(out);
        double* dst = role == 0 ? a.state[gid].color : a.state[gid].hist[role - 1];
        dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
    }
}

// true when the role form applies: band analysis on, windows long enough for the two-round distance between a ring store and the
// load of the same slot (see the memory wavefront).  Measured (tools/bench_meters.py waveform, 64 blocks per call): 3.1x ... 3.2x the
// one-wavefront kernel up to 1024 streams (one workgroup per CU, every wavefront at its own pace), 1.13x with RMS history at 4096
// streams, 0.89x without history at 4096 streams (four workgroups per CU's worth of work: the one-wavefront kernel's 1024
// wavefronts fill the SIMDs and execute fewer instructions in total) — that last shape keeps the one-wavefront kernel.
bool waveform_roles_applicable(const WaveformArgs& a) {
    return a.analyze != 0 && a.color_len >= 32 && a.slow_len >= 32 /* 2 R */ && (a.track_history != 0 || a.n_streams < 2048);
}

void launch_waveform_roles(const WaveformArgs& a, hipStream_t stream) {
    const uint32_t threads = a.n_streams * 16;
    const dim3 grid((threads + 63) / 64);
    if (a.frames_v) {  // ragged banks: per-stream frame counts, push counts, column phases and reset flags
        if (a.track_history) hipLaunchKernelGGL((waveform_roles_kernel<8, 2, true, true>), grid, dim3(448), 0, stream, a);
        else hipLaunchKernelGGL((waveform_roles_kernel<8, 2, false, true>), grid, dim3(320), 0, stream, a);
        return;
    }
    if (a.track_history) hipLaunchKernelGGL((waveform_roles_kernel<8, 2, true, false>), grid, dim3(448), 0, stream, a);
    else hipLaunchKernelGGL((waveform_roles_kernel<8, 2, false, false>), grid, dim3(320), 0, stream, a);
}

}  // namespace omx
