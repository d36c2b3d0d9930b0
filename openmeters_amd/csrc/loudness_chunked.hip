// K4c: the loudness recurrences evaluated chunk-parallel in time (one chunk = one block of a bank call).
//
// The sequential kernels (loudness_kernels.hip) walk a (stream, channel) slot's samples in order: exact reference order, but
// a 1024 x 2-channel shard (BASELINE configs[4]) is 2048 slots = 32 workgroups on a 256-CU part and takes 2.5 ms per 16 384
// frames whatever the arithmetic costs.  What the path computes is linear in its state or has finite memory:
//   K-weighting      one 4th-order f64 TDF-II (reference src/visuals/loudness/processor.rs:153-162): linear
//                    -> pass A (zero-state end state per block), a wave-parallel scan over the blocks with the block
//                       transition T (host, f64), pass B from the TRUE start state
//   WindowedMeans    (src/dsp.rs:264-371) the mean of the newest cap_w squared samples.  The reference keeps it as two
//                    Kahan-Babuska-Neumaier running sums with a periodic exact refresh — an accurate SLIDING evaluation of a
//                    window sum.  Here the same sums come from a prefix over 64-sample sub-block sums (KBN inside a
//                    sub-block, f64 running total Q since the last reset, kept in a per-slot ring): W = Q[end] - Q[start - 1].
//                    Both are accurate to ~1e-15 of the window sum; they differ in the last bits only (bar: 1e-4 dB).
//   true peak        (:123-151) 12- / 24-sample FIR memory: every block is computed from its own samples and the DL - 1
//                    before them, in the reference's accumulation order -> bit-identical to the sequential kernels.
// The squared samples still go to the f64 ring in the sequential kernels' layout, and the channel state (filter, KBN pairs,
// delay line) is written back in their form, so calls may alternate between the two forms.
// Shapes: channels in {1, 2, 4, 8}, block_frames and every window length a multiple of 64 (48 / 96 / 192 kHz), sample counter
// a multiple of 64.  Non-finite PCM is detected in pass A; every later kernel then leaves the state alone and the caller runs
// the sequential kernel instead (processor.rs has no reset in k_weighted: a NaN poisons the filter for good — order matters).
#include "loudness.hpp"

namespace omx {

namespace {

constexpr uint32_t kRow = 64;        // ring row = 64 slots (loudness_kernels.hip)
constexpr int STEP = 16;             // frames per staged PCM tile
constexpr uint32_t SUB = 64;         // samples per sub-block sum

__device__ __forceinline__ void kbn(double& sum, double& corr, double v) {  // dsp.rs:277-285
    const double next = sum + v;
    const bool big_sum = fabs(sum) >= fabs(v);
    const double big = big_sum ? sum : v, small = big_sum ? v : sum;
    corr += (big - next) + small;
    sum = next;
}
__device__ __forceinline__ float power_to_db(float power, float floor) {  // level.rs:28-34
    return power > 0.0f ? fmaxf(logf(power) * 4.3429448f, floor) : floor;
}
__device__ __forceinline__ float ms_to_lufs(double ms, float floor) {  // loudness/processor.rs:57-66
    return ms > 0.0 ? (float)fmax(fma(log10(ms), 10.0, -0.691), (double)floor) : floor;
}
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    return __hiloint2double(__shfl_up(__double2hiint(v), d), __shfl_up(__double2loint(v), d));
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}

// PCM tile of one slot group and one 16-frame step: rows = the 64 / C streams of the group, row stride 17 C floats (the pad
// keeps the per-lane reads conflict-free for C = 1, 2, 4, 8).  One wavefront = one workgroup: LDS traffic is in order.
struct Tile {
    const float* src[4];
    uint32_t dst[4];
    bool live[4];
    float4 pre[4];
    __device__ __forceinline__ void setup(const LoudChunkArgs& a, uint32_t group, uint64_t frame0, uint32_t lane) {
        const uint32_t C = a.channels, row_bytes = 64u * C, per_group = 64u / C;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const uint32_t byte = (lane + 64u * (uint32_t)n) * 16u;
            const uint32_t row = byte / row_bytes, inrow = byte % row_bytes;
            const uint64_t s = (uint64_t)group * per_group + row;
            live[n] = s < a.n_streams;
            src[n] = a.pcm + ((live[n] ? s : 0) * a.frames_total + frame0) * C + inrow / 4u;
            dst[n] = row * 17u * C + inrow / 4u;
        }
    }
    __device__ __forceinline__ void issue(uint32_t step, uint32_t C) {
#pragma unroll
        for (int n = 0; n < 4; ++n) pre[n] = *reinterpret_cast<const float4*>(src[n] + (uint64_t)step * STEP * C);
    }
    __device__ __forceinline__ uint32_t stage(float* tile) {  // returns 1 when a staged sample is not finite
        uint32_t bad = 0;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const float4 p = live[n] ? pre[n] : float4{0.0f, 0.0f, 0.0f, 0.0f};
            bad |= (!isfinite(p.x) || !isfinite(p.y) || !isfinite(p.z) || !isfinite(p.w)) ? 1u : 0u;
            tile[dst[n]] = p.x;
            tile[dst[n] + 1] = p.y;
            tile[dst[n] + 2] = p.z;
            tile[dst[n] + 3] = p.w;
        }
        return bad;
    }
};

}  // namespace

// ---- K-weighting: PASS 0 = zero-state end state of the block; PASS 1 = from the true start state: squared samples -> ring,
// sub-block sums.  grid (slot groups, blocks), 64 threads: lane = slot of the group.
template <int PASS>
__global__ __launch_bounds__(64) void loud_chunk_filter_kernel(LoudChunkArgs a) {
    __shared__ float tile[2][64 * 17];
    if (PASS == 1 && *a.bad != 0u) return;
    const uint32_t lane = threadIdx.x, group = blockIdx.x, c = blockIdx.y;
    const uint32_t C = a.channels, L = a.block_frames, steps = L / STEP;
    const uint32_t chan = group * 64u + lane;
    const bool live = chan < a.n_streams * C;
    Tile t;
    t.setup(a, group, (uint64_t)c * L, lane);
    const uint32_t rd = (lane / C) * 17u * C + (lane % C);
    double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
    double* cf = a.chunk_filter + ((uint64_t)chan * a.n_blocks + c) * 4u;
    if (PASS == 1 && live) {
        f0 = cf[0];
        f1 = cf[1];
        f2 = cf[2];
        f3 = cf[3];
    }
    const double b0 = a.b[0], b1 = a.b[1], b2 = a.b[2], b3 = a.b[3], b4 = a.b[4], a1 = a.a[1], a2 = a.a[2], a3 = a.a[3], a4 = a.a[4];
    double* ring_col = a.ring + (uint64_t)group * a.ring_len * kRow + lane;
    uint64_t pos = (a.frames_seen + (uint64_t)c * L) % a.ring_len;  // ring slot of the block's first sample
    double* sub = a.sub_sums + ((uint64_t)chan * a.n_blocks + c) * (L / SUB);
    double ssum = 0.0, scor = 0.0;
    uint32_t bad = 0;
    t.issue(0, C);
    for (uint32_t step = 0; step < steps; ++step) {
        bad |= t.stage(tile[step & 1u]);
        if (step + 1u < steps) t.issue(step + 1u, C);
        __syncthreads();
        const float* row = tile[step & 1u] + rd;
        float x[STEP];
#pragma unroll
        for (int f = 0; f < STEP; ++f) x[f] = row[f * C];
#pragma unroll
        for (int f = 0; f < STEP; ++f) {  // k_weighted (:153-162), the sequential kernels' statement order
            const double xd = (double)x[f];
            const double y = b0 * xd + f0;
            f0 = b1 * xd + f1 - a1 * y;
            f1 = b2 * xd + f2 - a2 * y;
            f2 = b3 * xd + f3 - a3 * y;
            f3 = b4 * xd - a4 * y;
            if constexpr (PASS == 1) {
                const double filtered = (double)(float)y;  // rounded to f32 before squaring (:161, :276-277)
                double value = filtered * filtered;
                value = isfinite(value) ? value : 0.0;     // WindowedMeans::push (dsp.rs:325)
                if (live) ring_col[pos * kRow] = value;
                pos = pos + 1u == a.ring_len ? 0u : pos + 1u;
                kbn(ssum, scor, value);
            }
        }
        if constexpr (PASS == 1) {
            if ((step + 1u) % (SUB / STEP) == 0u) {
                if (live) sub[(step + 1u) / (SUB / STEP) - 1u] = ssum + scor;
                ssum = scor = 0.0;
            }
        }
    }
    if constexpr (PASS == 0) {
        if (__ballot(bad != 0u) != 0ull && lane == 0) atomicOr(a.bad, 1u);
        if (live) {
            cf[0] = f0;
            cf[1] = f1;
            cf[2] = f2;
            cf[3] = f3;
        }
    }
}

// ---- scan of the filter states over the blocks: wavefront = slot, lane = block (see stereometer_chunked.hip for the scheme)
__global__ __launch_bounds__(256) void loud_scan_filter_kernel(LoudChunkArgs a, const double* __restrict__ Tp /* [6][4][4] */) {
    if (*a.bad != 0u) return;
    const uint32_t chan = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (chan >= a.n_streams * a.channels) return;
    LoudnessChannelState& st = a.state[chan];
    double carry[4] = {st.filter[0], st.filter[1], st.filter[2], st.filter[3]};
    for (uint32_t c0 = 0; c0 < a.n_blocks; c0 += 64u) {
        const uint32_t c = c0 + lane;
        const bool live = c < a.n_blocks;
        double* cf = a.chunk_filter + ((uint64_t)chan * a.n_blocks + (live ? c : c0)) * 4u;
        double x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = live ? cf[k] : 0.0;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double acc = x[k];
#pragma unroll
                for (int m = 0; m < 4; ++m) acc += Tp[k * 4 + m] * carry[m];
                x[k] = acc;
            }
        }
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int d = 1 << step;
            const double* Td = Tp + step * 16;
            double up[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) up[k] = shfl_up_f64(x[k], d);
            if ((int)lane >= d) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double acc = x[k];
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc += Td[k * 4 + m] * up[m];
                    x[k] = acc;
                }
            }
        }
        const uint32_t last = min(a.n_blocks - c0, 64u) - 1u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double start = shfl_up_f64(x[k], 1);
            if (lane == 0) start = carry[k];
            if (live) cf[k] = fabs(start) < 1.0e-30 ? 0.0 : start;  // denormal flush once per block (:281-285)
            const double e = shfl_f64(x[k], (int)last);
            carry[k] = fabs(e) < 1.0e-30 ? 0.0 : e;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) st.filter[k] = carry[k];
    }
}

// ---- true peak of every block (TruePeakMeter::process, :123-151): grid (slot groups, blocks), lane = slot.  Bit-identical to
// the sequential kernels: same samples, same tap order; the DL - 1 samples before the block come from the PCM of the call or,
// for its first block, from the carried delay line.
template <int DL>
__global__ __launch_bounds__(64) void loud_chunk_peak_kernel(LoudChunkArgs a) {
    __shared__ float tile[2][64 * 17];
    const uint32_t lane = threadIdx.x, group = blockIdx.x, c = blockIdx.y;
    const uint32_t C = a.channels, L = a.block_frames, steps = L / STEP;
    const uint32_t chan = group * 64u + lane;
    const bool live = chan < a.n_streams * C;
    const uint32_t s = chan / C, ch = chan % C;
    Tile t;
    t.setup(a, group, (uint64_t)c * L, lane);
    const uint32_t rd = (lane / C) * 17u * C + (lane % C);
    constexpr int H = DL > 1 ? DL - 1 : 1;
    float hist[H];  // hist[0] = newest sample before the block
#pragma unroll
    for (int i = 0; i < H; ++i) hist[i] = 0.0f;
    if (DL > 1 && live) {
        if (c == 0) {
#pragma unroll
            for (int i = 0; i < H; ++i) hist[i] = a.state[chan].delay[i];
        } else {
            const float* p = a.pcm + ((uint64_t)s * a.frames_total + (uint64_t)c * L) * C + ch;
#pragma unroll
            for (int i = 0; i < H; ++i) hist[i] = *(p - (int64_t)(i + 1) * C);  // L >= 64 > DL: inside the call
        }
    }
    float peak = 0.0f;
    t.issue(0, C);
    for (uint32_t step = 0; step < steps; ++step) {
        (void)t.stage(tile[step & 1u]);
        if (step + 1u < steps) t.issue(step + 1u, C);
        __syncthreads();
        const float* row = tile[step & 1u] + rd;
        float ext[STEP + H];  // ext[STEP - 1 - k] = x[k]; ext[STEP + i] = hist[i]
#pragma unroll
        for (int k = 0; k < STEP; ++k) ext[STEP - 1 - k] = live ? row[k * C] : 0.0f;
#pragma unroll
        for (int i = 0; i < H; ++i) ext[STEP + i] = hist[i];
#pragma unroll
        for (int k = 0; k < STEP; ++k) {
            peak = fmaxf(peak, fabsf(ext[STEP - 1 - k]));
            if constexpr (DL == 12) {
#pragma unroll
                for (int ph = 0; ph < 3; ++ph) {
                    float o = 0.0f;
#pragma unroll
                    for (int i = 0; i < 12; ++i) o += ext[STEP - 1 - k + i] * a.fir4[i][ph];
                    peak = fmaxf(peak, fabsf(o));
                }
            } else if constexpr (DL == 24) {
                float o = 0.0f;
#pragma unroll
                for (int i = 0; i < 24; ++i) o += ext[STEP - 1 - k + i] * a.fir2[i];
                peak = fmaxf(peak, fabsf(o));
            }
        }
#pragma unroll
        for (int i = 0; i < H; ++i) hist[i] = ext[i];
    }
    if (!live) return;
    omx_loudness_snapshot* snap = a.snapshots + (uint64_t)s * a.n_blocks + c;
    snap->true_peak_db[ch] = power_to_db(peak * peak, a.floor_db);
    if (ch == 0)
        for (uint32_t i = C; i < OMX_MAX_CHANNELS; ++i) snap->true_peak_db[i] = a.floor_db;  // with_floor (:197-207)
    if (c + 1u == a.n_blocks && *a.bad == 0u) {  // (pass A, which sets the flag, ran before this kernel)
        LoudnessChannelState& st = a.state[chan];
        if constexpr (DL > 1) {
#pragma unroll
            for (int i = 0; i < H; ++i) st.delay[i] = hist[i];
        }
        st.peak = 0.0f;
    }
}

// ---- prefix of the sub-block sums: Q[g] = sum of every squared sample up to the end of sub-block g since the last reset.
// wavefront = slot, lane = sub-block (64 per sweep)
__global__ __launch_bounds__(256) void loud_scan_q_kernel(LoudChunkArgs a) {
    if (*a.bad != 0u) return;
    const uint32_t chan = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (chan >= a.n_streams * a.channels) return;
    const uint64_t n_sub = (uint64_t)a.n_blocks * (a.block_frames / SUB), g0 = a.frames_seen / SUB;  // first new sub-block
    double* q = a.q_ring + (uint64_t)chan * a.q_len;
    double carry = g0 == 0 ? 0.0 : q[(g0 - 1u) & (a.q_len - 1u)];
    const double* sub = a.sub_sums + (uint64_t)chan * n_sub;
    for (uint64_t j0 = 0; j0 < n_sub; j0 += 64u) {
        const uint64_t j = j0 + lane;
        double x = j < n_sub ? sub[j] : 0.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double up = shfl_up_f64(x, d);
            if ((int)lane >= d) x += up;
        }
        x += carry;
        if (j < n_sub) q[(g0 + j) & (a.q_len - 1u)] = x;
        carry = shfl_f64(x, 63);
    }
}

// ---- rebuild Q from the squared-sample ring (after calls that went through the sequential kernels): sub-block sums of the
// newest min(frames_seen, ring_len) samples, then the same prefix.  grid (slot groups, sub-blocks), lane = slot.
__global__ __launch_bounds__(64) void loud_rebuild_sub_kernel(LoudChunkArgs a, uint64_t first_sub, double* out /* [chan][n] */, uint64_t n,
                                                              const uint32_t* only_if) {
    if (only_if && *only_if == 0u) return;
    const uint32_t lane = threadIdx.x, group = blockIdx.x;
    const uint64_t j = blockIdx.y;
    const uint32_t chan = group * 64u + lane;
    if (chan >= a.n_streams * a.channels) return;
    const double* ring_col = a.ring + (uint64_t)group * a.ring_len * kRow + lane;
    double s = 0.0, c = 0.0;
    const uint64_t sample0 = (first_sub + j) * SUB;
    for (uint32_t i = 0; i < SUB; ++i) kbn(s, c, ring_col[((sample0 + i) % a.ring_len) * kRow]);
    out[(uint64_t)chan * n + j] = s + c;
}
__global__ __launch_bounds__(256) void loud_rebuild_q_kernel(LoudChunkArgs a, uint64_t first_sub, const double* sub, uint64_t n,
                                                             const uint32_t* only_if) {
    if (only_if && *only_if == 0u) return;
    const uint32_t chan = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (chan >= a.n_streams * a.channels) return;
    double* q = a.q_ring + (uint64_t)chan * a.q_len;
    double carry = 0.0;  // only differences of Q are used once the windows are full; before that first_sub == 0
    for (uint64_t j0 = 0; j0 < n; j0 += 64u) {
        const uint64_t j = j0 + lane;
        double x = j < n ? sub[(uint64_t)chan * n + j] : 0.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double up = shfl_up_f64(x, d);
            if ((int)lane >= d) x += up;
        }
        x += carry;
        if (j < n) q[(first_sub + j) & (a.q_len - 1u)] = x;
        carry = shfl_f64(x, 63);
    }
    if (first_sub > 0 && lane == 0) q[(first_sub - 1u) & (a.q_len - 1u)] = 0.0;
}

// ---- snapshots (loudness/processor.rs:287-310) and the write-back of the KBN pairs: thread = (stream, block)
__global__ __launch_bounds__(256) void loud_chunk_snapshot_kernel(LoudChunkArgs a) {
    if (*a.bad != 0u) return;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint64_t)a.n_streams * a.n_blocks) return;
    const uint32_t s = (uint32_t)(i / a.n_blocks), c = (uint32_t)(i % a.n_blocks);
    const uint32_t C = a.channels;
    const uint64_t P = a.frames_seen + (uint64_t)(c + 1u) * a.block_frames;  // pushes at the end of this block
    const uint64_t mask = a.q_len - 1u;
    omx_loudness_snapshot* snap = a.snapshots + i;
    double short_term = 0.0, momentary = 0.0;
    const bool last = c + 1u == a.n_blocks;
    for (uint32_t ch = 0; ch < C; ++ch) {
        const uint32_t chan = s * C + ch;
        const double* q = a.q_ring + (uint64_t)chan * a.q_len;
        const double q_end = q[(P / SUB - 1u) & mask];
        double mean[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t m = min(P, a.capacities[w]);  // dsp.rs:367-370: min(count, cap), count itself saturates at the ring
            const uint64_t start_sub = (P - m) / SUB;
            const double base = start_sub == 0 ? 0.0 : q[(start_sub - 1u) & mask];
            const double W = q_end - base;
            mean[w] = W / (double)max(m, (uint64_t)1);
            if (last) {  // the sequential kernels' state: live pair = the window sum, `since refresh` pair = sum since the last
                         // multiple of cap pushes (CompensatedPair::refresh, dsp.rs:287-289, :363); corrections folded in
                LoudnessChannelState& st = a.state[chan];
                const uint64_t refresh_sub = (P / a.capacities[w]) * a.capacities[w] / SUB;
                const double rbase = refresh_sub == 0 ? 0.0 : q[(refresh_sub - 1u) & mask];
                st.sums[w][0] = W;
                st.corrections[w][0] = 0.0;
                st.sums[w][1] = q_end - rbase;
                st.corrections[w][1] = 0.0;
            }
        }
        short_term += mean[0] * a.weights[ch];  // position-weighted, channel order (:292-296)
        momentary += mean[1] * a.weights[ch];
        snap->rms_fast_db[ch] = power_to_db((float)mean[2], a.floor_db);
        snap->rms_slow_db[ch] = power_to_db((float)mean[3], a.floor_db);
    }
    snap->short_term_loudness = ms_to_lufs(short_term, a.floor_db);
    snap->momentary_loudness = ms_to_lufs(momentary, a.floor_db);
    snap->channel_count = C;
    snap->_pad = 0;
    for (int k = 0; k < OMX_MAX_CHANNELS; ++k) snap->positions[k] = a.positions[k];
    for (uint32_t k = C; k < OMX_MAX_CHANNELS; ++k) {  // LoudnessSnapshot::with_floor (:197-207)
        snap->rms_fast_db[k] = a.floor_db;
        snap->rms_slow_db[k] = a.floor_db;
    }
}

// a.frames_seen = the sample counter the ring content corresponds to; only_if: run only when *only_if != 0 (after a fallback)
void launch_loudness_rebuild_q(const LoudChunkArgs& a, double* scratch, const uint32_t* only_if, hipStream_t stream) {
    const uint64_t have = std::min<uint64_t>(a.frames_seen, a.ring_len);
    const uint64_t n = have / SUB, first_sub = (a.frames_seen - have) / SUB;
    if (n == 0) return;
    const uint32_t slots = a.n_streams * a.channels, groups = (slots + 63u) / 64u;
    hipLaunchKernelGGL(loud_rebuild_sub_kernel, dim3(groups, (uint32_t)n), dim3(64), 0, stream, a, first_sub, scratch, n, only_if);
    hipLaunchKernelGGL(loud_rebuild_q_kernel, dim3((slots + 3u) / 4u), dim3(256), 0, stream, a, first_sub, scratch, n, only_if);
}

void launch_loudness_chunked(const LoudChunkArgs& a, const double* d_T, hipStream_t stream) {
    const uint32_t slots = a.n_streams * a.channels, groups = (slots + 63u) / 64u;
    const dim3 grid(groups, a.n_blocks);
    hipLaunchKernelGGL(loud_chunk_filter_kernel<0>, grid, dim3(64), 0, stream, a);
    if (a.delay_len == 12) hipLaunchKernelGGL(loud_chunk_peak_kernel<12>, grid, dim3(64), 0, stream, a);
    else if (a.delay_len == 24) hipLaunchKernelGGL(loud_chunk_peak_kernel<24>, grid, dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(loud_chunk_peak_kernel<0>, grid, dim3(64), 0, stream, a);
    hipLaunchKernelGGL(loud_scan_filter_kernel, dim3((slots + 3u) / 4u), dim3(256), 0, stream, a, d_T);
    hipLaunchKernelGGL(loud_chunk_filter_kernel<1>, grid, dim3(64), 0, stream, a);
    hipLaunchKernelGGL(loud_scan_q_kernel, dim3((slots + 3u) / 4u), dim3(256), 0, stream, a);
    const uint64_t snaps = (uint64_t)a.n_streams * a.n_blocks;
    hipLaunchKernelGGL(loud_chunk_snapshot_kernel, dim3((uint32_t)((snaps + 255u) / 256u)), dim3(256), 0, stream, a);
}

}  // namespace omx
