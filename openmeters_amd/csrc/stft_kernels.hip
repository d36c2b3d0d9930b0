// HIP kernels of the STFT hot path (gfx950 only).
//
//   K0 ingest_project      AudioBlock::stereo_frames + Channel::project + deque push
//                          (reference src/dsp.rs:223-257, src/util/audio/channel.rs:13-21,
//                           src/visuals/spectrogram/processor.rs:412-437)
//   K2 stft_reassigned_4096 the headline kernel: Hilbert analytic signal + three windowed FFTs +
//                          reassignment + ordered compaction, fused, one (stream, hop) per workgroup
//                          (reference spectrogram/processor.rs:318-348, :439-488, :546-567)
//   K1/K2 stft_generic      any power-of-two window / zero-padding, classic or reassigned, radix-2 in
//                          global scratch in the oracle's operation order
//                          (reference spectrogram/processor.rs:306-385)
//   derivative_window       spectral derivative of the window (reference :569-599)
#include "stft_kernels.hpp"

#include "fft_device.hpp"
#ifdef OMX_TUNING
#include "fft_wave_device.hpp"
#endif

namespace omx {

// ================================================================================================
// K0: fold C interleaved channels to [L,R], project, append to the per-stream mono rings.
// ================================================================================================
__device__ __forceinline__ float project_lr(int channel, float left, float right) {
    switch (channel) {  // channel.rs:13-21
        case OMX_CHANNEL_LEFT: return left;
        case OMX_CHANNEL_RIGHT: return right;
        case OMX_CHANNEL_MID: return (left + right) * 0.5f;
        case OMX_CHANNEL_SIDE: return (left - right) * 0.5f;
        default: return 0.0f;
    }
}

constexpr int INGEST_FRAMES_PER_THREAD = 16;
constexpr int INGEST_FRAMES_PER_WG = 256 * INGEST_FRAMES_PER_THREAD;

// One workgroup folds 4096 consecutive frames of one stream (16 per thread, coalesced across the wave
// for every k so 16 loads are in flight per lane).  The newest non-zero position of the chunk goes to
// `partial[s][wg]`; a one-wave finalize kernel folds the partials into last_nonzero[s] (one contended
// 64-bit atomic per wave cost 1.4 ms per step in the first version of this kernel).
__global__ __launch_bounds__(256) void ingest_project_kernel(IngestArgs a) {
    __shared__ long long wave_best[4];
    const uint32_t s = blockIdx.y;
    const uint64_t wg_base = (uint64_t)blockIdx.x * INGEST_FRAMES_PER_WG;
    const uint64_t skip_s = a.skips ? a.skips[s] : a.skip, count_s = a.counts ? a.counts[s] : a.count,
                   head_s = a.heads ? a.heads[s] : (a.per_out ? a.head_o[0] : a.head);
    const float* src = a.pcm + ((uint64_t)s * a.frames_total + skip_s) * a.fmt.channels;
    const uint64_t ring_base = (uint64_t)s * a.cap;
    long long best = -1;
    // two-channel blocks (the common shape): every load of the thread is issued up front, unconditionally (clamped index) — a load
    // inside `if (live)` is waited for on the spot, which left four of the sixteen in flight
    const bool two = a.fmt.channels == 2;
    v2f pre[INGEST_FRAMES_PER_THREAD];
    if (two) {
        // clamped inside the stream's own row of `pcm`: a stream that pushes nothing (count 0, skip = its whole block) still reads a
        // frame of its own
        const uint64_t row_last = a.frames_total ? a.frames_total - 1 : 0;
        const uint64_t last = count_s ? count_s - 1 : 0;
        const float* row = a.pcm + (uint64_t)s * a.frames_total * 2;
#pragma unroll
        for (int k = 0; k < INGEST_FRAMES_PER_THREAD; ++k)
            pre[k] = *reinterpret_cast<const v2f*>(row + min(skip_s + min(wg_base + (uint64_t)k * 256 + threadIdx.x, last), row_last) * 2);
    }
#pragma unroll
    for (int k = 0; k < INGEST_FRAMES_PER_THREAD; ++k) {
        const uint64_t idx = wg_base + (uint64_t)k * 256 + threadIdx.x;
        const bool live = idx < count_s;
        float out0 = 0.0f;
        if (live) {
            // dsp.rs:223-249: left = (0.0 + s0*w00) + s1*w10 + ...  (same order as the general fold; the
            // 1- and 2-channel specialisations of the reference are bit-identical to it, dsp.rs:591-624)
            float left = 0.0f, right = 0.0f, first;
            if (two) {
                const v2f f2 = pre[k];
                first = f2.x;
                left = left + f2.x * a.fmt.m[0][0];
                right = right + f2.x * a.fmt.m[0][1];
                left = left + f2.y * a.fmt.m[1][0];
                right = right + f2.y * a.fmt.m[1][1];
            } else {
                const float* frame = src + idx * a.fmt.channels;
                first = frame[0];
                for (uint32_t c = 0; c < a.fmt.channels; ++c) {
                    const float v = frame[c];
                    left = left + v * a.fmt.m[c][0];
                    right = right + v * a.fmt.m[c][1];
                }
            }
            const uint64_t slot = ring_base + ((head_s + idx) & (a.cap - 1));
#pragma unroll
            for (int o = 0; o < OMX_INGEST_MAX_OUT; ++o) {
                if (o >= a.n_out) break;
                // OMX_PROJECT_RAW: spectrogram's `channels == 1` path pushes the raw samples (:420-428)
                const float v = a.project[o] == OMX_PROJECT_RAW ? first : project_lr(a.project[o], left, right);
                if (o == 0) out0 = v;
                const uint64_t slot_o = a.per_out ? (uint64_t)s * a.cap_o[o] + ((a.head_o[o] + idx) & (a.cap_o[o] - 1)) : slot;
                a.ring[o][slot_o] = v;
            }
        }
        const unsigned long long nz = __ballot(live && out0 != 0.0f);  // audio_last_nonzero (:423-425, :432-434)
        if (nz != 0ull)
            best = (long long)(head_s + wg_base + (uint64_t)k * 256 + (threadIdx.x & ~63u)) + (63 - __clzll((long long)nz));
    }
    if (a.partial_nonzero) {
        if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long m = wave_best[0];
            for (int w = 1; w < 4; ++w) m = wave_best[w] > m ? wave_best[w] : m;
            if (gridDim.x == 1 && a.last_nonzero) {  // one workgroup per stream (every single-block call): no finalize launch
                if (m > a.last_nonzero[s]) a.last_nonzero[s] = m;
            } else {
                a.partial_nonzero[(uint64_t)s * gridDim.x + blockIdx.x] = m;
            }
        }
    }
}

// Two-channel blocks, four consecutive frames per lane and round: the PCM arrives as two 16-byte loads per lane and every ring receives one
// 16-byte store per lane (the one-frame-per-lane kernel above issues 4-byte stores, 32 per thread for two rings: at cfg2's
// 134 MB in / 2 x 67 MB out it ran at 3.2 TB/s).  Same arithmetic per frame, same rings, same newest-non-zero position.
constexpr int INGEST4_ROUNDS = 4;  // 4 rounds x 4 frames x 256 threads = the same 4096 frames per workgroup
__global__ __launch_bounds__(256) void ingest_project4_kernel(IngestArgs a) {
    __shared__ long long wave_best[4];
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const uint32_t s = blockIdx.y;
    const uint64_t wg_base = (uint64_t)blockIdx.x * INGEST_FRAMES_PER_WG;
    const uint64_t skip_s = a.skips ? a.skips[s] : a.skip, count_s = a.counts ? a.counts[s] : a.count,
                   head_s = a.heads ? a.heads[s] : (a.per_out ? a.head_o[0] : a.head);
    const uint64_t row_last = a.frames_total ? a.frames_total - 1 : 0;
    const float* row = a.pcm + (uint64_t)s * a.frames_total * 2;
    long long best = -1;
    f4u lo[INGEST4_ROUNDS], hi[INGEST4_ROUNDS];
    uint64_t idx0[INGEST4_ROUNDS];
#pragma unroll
    for (int k = 0; k < INGEST4_ROUNDS; ++k) {  // every load up front; a lane whose four frames are not all inside the row reads clamped singles
        idx0[k] = wg_base + (uint64_t)k * 1024 + 4u * threadIdx.x;
        const uint64_t f = skip_s + idx0[k];
        if (f + 3 <= row_last) {
            lo[k] = *reinterpret_cast<const f4u*>(row + f * 2);
            hi[k] = *reinterpret_cast<const f4u*>(row + f * 2 + 4);
        } else {
            float t[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2f p = *reinterpret_cast<const v2f*>(row + min(f + (uint64_t)j, row_last) * 2);
                t[2 * j] = p.x;
                t[2 * j + 1] = p.y;
            }
            lo[k] = f4u{t[0], t[1], t[2], t[3]};
            hi[k] = f4u{t[4], t[5], t[6], t[7]};
        }
    }
#pragma unroll
    for (int k = 0; k < INGEST4_ROUNDS; ++k) {
        const float fr[8] = {lo[k].x, lo[k].y, lo[k].z, lo[k].w, hi[k].x, hi[k].y, hi[k].z, hi[k].w};
        float left[4], right[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // dsp.rs:223-249: left = (0.0 + s0*w00) + s1*w10
            left[j] = 0.0f + fr[2 * j] * a.fmt.m[0][0];
            right[j] = 0.0f + fr[2 * j] * a.fmt.m[0][1];
            left[j] = left[j] + fr[2 * j + 1] * a.fmt.m[1][0];
            right[j] = right[j] + fr[2 * j + 1] * a.fmt.m[1][1];
        }
        const uint32_t n_live = idx0[k] >= count_s ? 0u : (uint32_t)min((uint64_t)4, count_s - idx0[k]);
        float out0[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < OMX_INGEST_MAX_OUT; ++o) {
            if (o >= a.n_out) break;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a.project[o] == OMX_PROJECT_RAW ? fr[2 * j] : project_lr(a.project[o], left[j], right[j]);
            if (o == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) out0[j] = v[j];
            }
            const uint64_t cap = a.per_out ? a.cap_o[o] : a.cap, head = a.per_out ? a.head_o[o] : head_s;
            float* ring = a.ring[o] + (uint64_t)s * cap;
            const uint64_t p = (head + idx0[k]) & (cap - 1);
            if (n_live == 4u && p + 3 < cap) {
                *reinterpret_cast<f4u*>(ring + p) = f4u{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((uint32_t)j < n_live) ring[(head + idx0[k] + (uint64_t)j) & (cap - 1)] = v[j];
            }
        }
        // audio_last_nonzero (:423-425, :432-434): the newest non-zero sample of this round
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            const unsigned long long nz = __ballot((uint32_t)j < n_live && out0[j] != 0.0f);
            if (nz != 0ull) {
                const long long cand = (long long)(head_s + wg_base + (uint64_t)k * 1024 + 4u * ((threadIdx.x & ~63u) + (uint32_t)(63 - __clzll((long long)nz))) + (uint32_t)j);
                best = cand > best ? cand : best;
            }
        }
    }
    if (a.partial_nonzero) {
        if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long m = wave_best[0];
            for (int w = 1; w < 4; ++w) m = wave_best[w] > m ? wave_best[w] : m;
            if (gridDim.x == 1 && a.last_nonzero) {
                if (m > a.last_nonzero[s]) a.last_nonzero[s] = m;
            } else {
                a.partial_nonzero[(uint64_t)s * gridDim.x + blockIdx.x] = m;
            }
        }
    }
}

__global__ __launch_bounds__(64) void ingest_finalize_kernel(const long long* partial, uint32_t n_partials,
                                                             long long* last_nonzero) {
    const uint32_t s = blockIdx.x;
    long long m = -1;
    for (uint32_t i = threadIdx.x; i < n_partials; i += 64) {
        const long long v = partial[(uint64_t)s * n_partials + i];
        m = v > m ? v : m;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const long long o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if (threadIdx.x == 0 && m > last_nonzero[s]) last_nonzero[s] = m;
}

uint32_t ingest_partials_per_stream(uint64_t count) { return (uint32_t)((count + INGEST_FRAMES_PER_WG - 1) / INGEST_FRAMES_PER_WG); }

void launch_ingest(const IngestArgs& a, uint32_t n_streams, hipStream_t stream) {
    if (a.count == 0 || n_streams == 0) return;
    const uint32_t wgs = ingest_partials_per_stream(a.count);
    static const bool one_frame_form = tuning_env("OMX_INGEST_SINGLE") != nullptr;  // tuning hook: A/B against the one-frame-per-lane kernel
    if (a.fmt.channels == 2 && !one_frame_form) hipLaunchKernelGGL(ingest_project4_kernel, dim3(wgs, n_streams), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(ingest_project_kernel, dim3(wgs, n_streams), dim3(256), 0, stream, a);
    if (a.partial_nonzero && a.last_nonzero && wgs > 1)
        hipLaunchKernelGGL(ingest_finalize_kernel, dim3(n_streams), dim3(64), 0, stream, a.partial_nonzero, wgs, a.last_nonzero);
}

// ---- several ragged banks, one launch: output o pushes frame F of stream s iff skip_o[s] <= F < skip_o[s] + count_o[s], to slot
// head_o[s] + F - skip_o[s] of its own ring (capacity cap_o).  Frames are indexed by their position in the block, so outputs whose
// banks skip different numbers of leading frames (pending_skip after a hop longer than the window) share the loads all the same.
struct IngestMultiArgs {
    const float* pcm;
    uint64_t frames_total;
    AudioFormatArgs fmt;
    int n_out;
    int project[OMX_INGEST_MAX_OUT];
    float* ring[OMX_INGEST_MAX_OUT];
    uint64_t cap[OMX_INGEST_MAX_OUT];
    const uint32_t* skips[OMX_INGEST_MAX_OUT];
    const uint32_t* counts[OMX_INGEST_MAX_OUT];
    const uint64_t* heads[OMX_INGEST_MAX_OUT];
    long long* last_nonzero;     // of output 0, or nullptr
    long long* partial_nonzero;  // [n_streams][gridDim.x]
};
__global__ __launch_bounds__(256) void ingest_project4_multi_kernel(IngestMultiArgs a) {
    __shared__ long long wave_best[4];
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const uint32_t s = blockIdx.y;
    const uint64_t wg_base = (uint64_t)blockIdx.x * INGEST_FRAMES_PER_WG;
    const uint64_t row_last = a.frames_total ? a.frames_total - 1 : 0;
    const float* row = a.pcm + (uint64_t)s * a.frames_total * 2;
    uint64_t skip_o[OMX_INGEST_MAX_OUT], count_o[OMX_INGEST_MAX_OUT], head_o[OMX_INGEST_MAX_OUT];
#pragma unroll
    for (int o = 0; o < OMX_INGEST_MAX_OUT; ++o) {
        const bool on = o < a.n_out;
        skip_o[o] = on ? a.skips[o][s] : 0;
        count_o[o] = on ? a.counts[o][s] : 0;
        head_o[o] = on ? a.heads[o][s] : 0;
    }
    long long best = -1;
#pragma unroll
    for (int k = 0; k < INGEST4_ROUNDS; ++k) {
        const uint64_t F = wg_base + (uint64_t)k * 1024 + 4u * threadIdx.x;  // first of this lane's four frames
        if (F > row_last) continue;
        float fr[8];
        if (F + 3 <= row_last) {
            const f4u lo = *reinterpret_cast<const f4u*>(row + F * 2), hi = *reinterpret_cast<const f4u*>(row + F * 2 + 4);
            fr[0] = lo.x; fr[1] = lo.y; fr[2] = lo.z; fr[3] = lo.w;
            fr[4] = hi.x; fr[5] = hi.y; fr[6] = hi.z; fr[7] = hi.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2f p = *reinterpret_cast<const v2f*>(row + min(F + (uint64_t)j, row_last) * 2);
                fr[2 * j] = p.x;
                fr[2 * j + 1] = p.y;
            }
        }
        float left[4], right[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // dsp.rs:223-249: left = (0.0 + s0*w00) + s1*w10
            left[j] = 0.0f + fr[2 * j] * a.fmt.m[0][0];
            right[j] = 0.0f + fr[2 * j] * a.fmt.m[0][1];
            left[j] = left[j] + fr[2 * j + 1] * a.fmt.m[1][0];
            right[j] = right[j] + fr[2 * j + 1] * a.fmt.m[1][1];
        }
#pragma unroll
        for (int o = 0; o < OMX_INGEST_MAX_OUT; ++o) {
            if (o >= a.n_out) break;
            float v[4];
            bool live[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = a.project[o] == OMX_PROJECT_RAW ? fr[2 * j] : project_lr(a.project[o], left[j], right[j]);
                const uint64_t f = F + (uint64_t)j;
                live[j] = f >= skip_o[o] && f - skip_o[o] < count_o[o];
            }
            const uint64_t cap = a.cap[o], first = head_o[o] + F - skip_o[o];  // (wraps when F < skip: then no lane of the quad below is live at j = 0)
            float* ring = a.ring[o] + (uint64_t)s * cap;
            const uint64_t p = first & (cap - 1);
            if (live[0] && live[3] && p + 3 < cap) {
                *reinterpret_cast<f4u*>(ring + p) = f4u{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (live[j]) ring[(first + (uint64_t)j) & (cap - 1)] = v[j];
            }
            if (o == 0 && a.partial_nonzero) {  // audio_last_nonzero (:423-425, :432-434): the newest non-zero sample pushed to ring 0
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (live[j] && v[j] != 0.0f) {
                        const long long cand = (long long)(first + (uint64_t)j);
                        best = cand > best ? cand : best;
                    }
            }
        }
    }
    if (a.partial_nonzero) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const long long other = __shfl_xor(best, off);
            best = other > best ? other : best;
        }
        if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long m = wave_best[0];
            for (int w = 1; w < 4; ++w) m = wave_best[w] > m ? wave_best[w] : m;
            if (gridDim.x == 1 && a.last_nonzero) {
                if (m > a.last_nonzero[s]) a.last_nonzero[s] = m;
            } else {
                a.partial_nonzero[(uint64_t)s * gridDim.x + blockIdx.x] = m;
            }
        }
    }
}

// the same for any channel count: one frame per lane and round (the fold of ingest_project_kernel)
__global__ __launch_bounds__(256) void ingest_project_multi_kernel(IngestMultiArgs a) {
    __shared__ long long wave_best[4];
    const uint32_t s = blockIdx.y;
    const uint64_t wg_base = (uint64_t)blockIdx.x * INGEST_FRAMES_PER_WG;
    const uint32_t C = a.fmt.channels;
    const float* row = a.pcm + (uint64_t)s * a.frames_total * C;
    uint64_t skip_o[OMX_INGEST_MAX_OUT], count_o[OMX_INGEST_MAX_OUT], head_o[OMX_INGEST_MAX_OUT];
#pragma unroll
    for (int o = 0; o < OMX_INGEST_MAX_OUT; ++o) {
        const bool on = o < a.n_out;
        skip_o[o] = on ? a.skips[o][s] : 0;
        count_o[o] = on ? a.counts[o][s] : 0;
        head_o[o] = on ? a.heads[o][s] : 0;
    }
    long long best = -1;
    for (int k = 0; k < INGEST_FRAMES_PER_THREAD; ++k) {
        const uint64_t F = wg_base + (uint64_t)k * 256 + threadIdx.x;
        if (F >= a.frames_total) continue;
        const float* frame = row + F * C;
        float left = 0.0f, right = 0.0f;  // dsp.rs:223-249
        const float first = frame[0];
        for (uint32_t c = 0; c < C; ++c) {
            const float v = frame[c];
            left = left + v * a.fmt.m[c][0];
            right = right + v * a.fmt.m[c][1];
        }
#pragma unroll
        for (int o = 0; o < OMX_INGEST_MAX_OUT; ++o) {
            if (o >= a.n_out) break;
            if (!(F >= skip_o[o] && F - skip_o[o] < count_o[o])) continue;
            const float v = a.project[o] == OMX_PROJECT_RAW ? first : project_lr(a.project[o], left, right);
            const uint64_t at = head_o[o] + F - skip_o[o];
            a.ring[o][(uint64_t)s * a.cap[o] + (at & (a.cap[o] - 1))] = v;
            if (o == 0 && v != 0.0f) best = (long long)at > best ? (long long)at : best;  // audio_last_nonzero (:423-425, :432-434)
        }
    }
    if (a.partial_nonzero) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const long long other = __shfl_xor(best, off);
            best = other > best ? other : best;
        }
        if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long m = wave_best[0];
            for (int w = 1; w < 4; ++w) m = wave_best[w] > m ? wave_best[w] : m;
            if (gridDim.x == 1 && a.last_nonzero) {
                if (m > a.last_nonzero[s]) a.last_nonzero[s] = m;
            } else {
                a.partial_nonzero[(uint64_t)s * gridDim.x + blockIdx.x] = m;
            }
        }
    }
}

bool launch_ingest_ragged_parts(const IngestArgs* parts, int n_parts, uint32_t n_streams, hipStream_t stream) {
    if (n_parts <= 0 || n_streams == 0) return true;
    IngestMultiArgs m{};
    m.pcm = parts[0].pcm;
    m.frames_total = parts[0].frames_total;
    m.fmt = parts[0].fmt;
    for (int b = 0; b < n_parts; ++b) {
        const IngestArgs& ia = parts[b];
        if (ia.pcm != m.pcm || ia.frames_total != m.frames_total || !ia.skips || !ia.counts || !ia.heads) return false;
        if (b > 0 && ia.last_nonzero) return false;  // only the first part can track it
        for (int o = 0; o < ia.n_out; ++o) {
            if (m.n_out >= OMX_INGEST_MAX_OUT) return false;
            m.project[m.n_out] = ia.project[o];
            m.ring[m.n_out] = ia.ring[o];
            m.cap[m.n_out] = ia.cap;
            m.skips[m.n_out] = ia.skips;
            m.counts[m.n_out] = ia.counts;
            m.heads[m.n_out] = ia.heads;
            ++m.n_out;
        }
    }
    if (m.n_out == 0 || m.frames_total == 0) return true;
    m.last_nonzero = parts[0].last_nonzero;
    m.partial_nonzero = parts[0].partial_nonzero;
    const uint32_t wgs = ingest_partials_per_stream(m.frames_total);
    if (m.fmt.channels == 2) hipLaunchKernelGGL(ingest_project4_multi_kernel, dim3(wgs, n_streams), dim3(256), 0, stream, m);
    else hipLaunchKernelGGL(ingest_project_multi_kernel, dim3(wgs, n_streams), dim3(256), 0, stream, m);
    if (m.partial_nonzero && m.last_nonzero && wgs > 1)
        hipLaunchKernelGGL(ingest_finalize_kernel, dim3(n_streams), dim3(64), 0, stream, m.partial_nonzero, wgs, m.last_nonzero);
    return true;
}

void launch_ingest_slots(const float* d_pcm, uint64_t frames, const AudioFormatArgs& fmt, const IngestSlots* const* banks, int n_banks,
                         uint32_t n_streams, hipStream_t stream) {
    IngestArgs ia{};
    ia.pcm = d_pcm;
    ia.frames_total = frames;
    ia.fmt = fmt;
    ia.per_out = 1;
    for (int b = 0; b < n_banks; ++b) {
        const IngestSlots& sl = *banks[b];
        if (sl.count == 0 || sl.n == 0) continue;
        if (ia.n_out == 0) {
            ia.skip = sl.skip;
            ia.count = sl.count;
            ia.last_nonzero = sl.last_nonzero;        // (only a bank listed first can track it: out0 of the kernel)
            ia.partial_nonzero = sl.partial_nonzero;
        }
        for (int o = 0; o < sl.n && ia.n_out < OMX_INGEST_MAX_OUT; ++o) {
            ia.project[ia.n_out] = sl.project[o];
            ia.ring[ia.n_out] = sl.ring[o];
            ia.cap_o[ia.n_out] = sl.cap[o];
            ia.head_o[ia.n_out] = sl.head[o];
            ++ia.n_out;
        }
    }
    ia.cap = ia.cap_o[0];
    ia.head = ia.head_o[0];
    if (ia.n_out) launch_ingest(ia, n_streams, stream);
}

// ================================================================================================
// Ragged banks: per-stream frame indexing on the device (bit-exact integers: `ready = (pending - read_len) / hop + 1`)
// ================================================================================================
__global__ __launch_bounds__(64) void spectrogram_plan_kernel(SpectrogramPlanArgs a) {
    const uint32_t s = blockIdx.x * 64u + threadIdx.x;
    if (s >= a.n_streams) return;
    uint64_t head = a.head[s], tail = a.tail[s], pending_skip = a.pending_skip[s];
    uint32_t reset = a.reset_flag[s];
    if (a.reset_mask && a.reset_mask[s]) {  // reset_audio (:212-217)
        tail = head;
        pending_skip = 0;
        a.last_nonzero[s] = -1;
        reset = 1u;
    }
    const uint64_t frames = a.frames[s];
    uint32_t n_cols = 0, skip32 = 0, count32 = 0;
    const uint64_t head_before = head;
    if (frames != 0) {  // block.is_empty() -> None, nothing else happens (:490-492)
        // push_audio (:412-437)
        const uint64_t skip = min(pending_skip, frames);
        pending_skip -= skip;
        const uint64_t count = frames - skip;
        skip32 = (uint32_t)skip;
        count32 = (uint32_t)count;
        head += count;
        // process_ready_windows (:281-388)
        const uint64_t pending = head - tail;
        const uint64_t ready = pending >= a.read_len ? (pending - a.read_len) / a.hop + 1u : 0u;
        const uint64_t skip_cols = ready > a.retained ? ready - a.retained : 0u;
        {   // advance_audio(skip_cols * hop) (:406-410)
            const uint64_t cnt = skip_cols * a.hop, len = head - tail, taken = min(cnt, len);
            tail += taken;
            pending_skip += cnt - taken;
        }
        n_cols = (uint32_t)min(ready - skip_cols, (uint64_t)a.max_cols);
        a.col_tail[s] = tail;
        {   // one advance_audio(hop) per column (:384): once the buffer runs dry every further hop is all `missing`
            const uint64_t cnt = (uint64_t)n_cols * a.hop, len = head - tail, taken = min(cnt, len);
            tail += taken;
            pending_skip += cnt - taken;
        }
    } else {
        a.col_tail[s] = tail;
    }
    a.ing_skip[s] = skip32;
    a.ing_count[s] = count32;
    a.ing_head[s] = head_before;
    a.n_cols[s] = n_cols;
    a.reset_out[s] = reset;
    if (n_cols != 0) reset = 0u;  // std::mem::take (:511): consumed by the update this call returns
    a.head[s] = head;
    a.tail[s] = tail;
    a.pending_skip[s] = pending_skip;
    a.reset_flag[s] = reset;
}
void launch_spectrogram_plan(const SpectrogramPlanArgs& a, hipStream_t stream) {
    if (a.n_streams == 0) return;
    hipLaunchKernelGGL(spectrogram_plan_kernel, dim3((a.n_streams + 63u) / 64u), dim3(64), 0, stream, a);
}

__global__ __launch_bounds__(256) void ring_rehome_kernel(const float* from, uint64_t from_cap, float* to, uint64_t to_cap,
                                                          const uint64_t* head, const uint64_t* tail) {
    const uint32_t s = blockIdx.x;
    const uint64_t h = head[s], t = tail[s];
    for (uint64_t p = t + threadIdx.x; p < h; p += 256u) to[(uint64_t)s * to_cap + (p & (to_cap - 1u))] = from[(uint64_t)s * from_cap + (p & (from_cap - 1u))];
}
// update_config on a bank whose positions are per stream (spectrogram/processor.rs:518-543 + rebuild_fft :275-278): every stream keeps
// its newest `keep` pending samples (0: a rate change drops them all), forgets what it still had to skip, and flags its next update
__global__ __launch_bounds__(256) void spectrogram_ragged_config_kernel(uint32_t n_streams, const uint64_t* head, uint64_t* tail, uint64_t* pending_skip,
                                                                        uint32_t* reset_flag, long long* last_nonzero, int trim, uint64_t keep,
                                                                        int zero_skip, int clear_nonzero) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_streams) return;
    if (trim) {
        const uint64_t pending = head[s] - tail[s];
        if (pending > keep) tail[s] = head[s] - keep;
    }
    if (zero_skip) pending_skip[s] = 0;
    if (clear_nonzero) last_nonzero[s] = -1;
    reset_flag[s] = 1;
}
void launch_spectrogram_ragged_config(uint32_t n_streams, const uint64_t* head, uint64_t* tail, uint64_t* pending_skip, uint32_t* reset_flag,
                                      long long* last_nonzero, bool trim, uint64_t keep, bool zero_skip, bool clear_nonzero, hipStream_t stream) {
    if (n_streams == 0) return;
    hipLaunchKernelGGL(spectrogram_ragged_config_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, n_streams, head, tail, pending_skip,
                       reset_flag, last_nonzero, trim ? 1 : 0, keep, zero_skip ? 1 : 0, clear_nonzero ? 1 : 0);
}

void launch_ring_rehome(const float* from, uint64_t from_cap, float* to, uint64_t to_cap, const uint64_t* head, const uint64_t* tail,
                        uint32_t n_streams, hipStream_t stream) {
    if (n_streams == 0) return;
    hipLaunchKernelGGL(ring_rehome_kernel, dim3(n_streams), dim3(256), 0, stream, from, from_cap, to, to_cap, head, tail);
}

// ================================================================================================
// Shared pieces of the STFT kernels
// ================================================================================================
__device__ __forceinline__ float power_to_db_dev(float power, float floor) {  // level.rs:28-34
    return power > 0.0f ? fmaxf(logf(power) * 4.3429448f, floor) : floor;
}
__device__ __forceinline__ uint16_t pack_classic_db_dev(float db) {  // spectrogram/processor.rs:103-108
    const float SCALE = 65535.0f / 156.0f;
    float v = roundf((db - (-144.0f)) * SCALE);
    v = v < 0.0f ? 0.0f : (v > 65535.0f ? 65535.0f : v);
    return (v == v) ? (uint16_t)v : (uint16_t)0;
}

// XCD-aware block -> (stream, column) map: block b runs on XCD b % 8 (observed dispatch order), so
// stream s is pinned to XCD s % 8 and its consecutive columns are consecutive blocks on that XCD:
// the 97 % overlap between neighbouring windows is served from that XCD's L2.
__device__ __forceinline__ bool block_to_stream_column(uint32_t n_streams, uint32_t n_cols, uint32_t& s, uint32_t& col) {
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, q = b >> 3;
    s = (q / n_cols) * 8u + xcd;
    col = q % n_cols;
    return s < n_streams;
}
uint32_t stream_column_grid(uint32_t n_streams, uint32_t n_cols) { return ((n_streams + 7u) / 8u) * 8u * n_cols; }

// spectrogram/processor.rs:459-485 for one bin; returns keep flag
struct ReassignConsts {
    float bin_hz, max_hz, inv_2pi, inv_hop, latency_hops;
};
__device__ __forceinline__ bool reassign_bin(uint32_t i, v2f b, v2f d, v2f t, float norm, const ReassignConsts& c,
                                             omx_spectrogram_point& p) {
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

// Branch-free form for the fused kernels: the nine bins of a thread become nine independent dependency chains the scheduler
// can interleave (the early returns above compile to exec-mask regions that serialise them), and 1/pow is one v_rcp_f32
// (1 ulp) instead of the 10-instruction IEEE sequence.  pow only scales the two CORRECTION terms (|d_omega| <~ a few bins,
// |t_hat| <~ W/2 samples), so a 1-ulp reciprocal moves freq_hz by < 1e-9 of Nyquist and time_offset by < 1e-6 hop.
// A dropped bin (power under the floor, pow == 0 included) may carry NaN in `p`; the caller only stores kept points.
__device__ __forceinline__ bool reassign_bin_fast(uint32_t i, v2f b, v2f d, v2f t, float norm, const ReassignConsts& c,
                                                  omx_spectrogram_point& p) {
    const float pow = b.x * b.x + b.y * b.y;
    const float scaled_power = pow * norm;
    const float inv_pow = __builtin_amdgcn_rcpf(pow);
    const float d_omega = -(d.y * b.x - d.x * b.y) * inv_pow;
    const float freq_hz = (float)i * c.bin_hz + d_omega * c.inv_2pi;
    p.time_offset = (t.x * b.x + t.y * b.y) * inv_pow * c.inv_hop - c.latency_hops;
    p.freq_hz = freq_hz;
    p.power = scaled_power;
    return !(scaled_power < 1e-14f) && (freq_hz > 0.0f && c.max_hz - freq_hz > 0.0f);
}

// ================================================================================================
// K2: fused reassigned STFT, W = F = 4096, H = 8192.  256 threads, two padded 4096-complex LDS
// buffers (68 KiB) -> two workgroups per CU.  Algorithm (all f32):
//   1. 8192 real samples packed as 4096 complex -> FFT4096 (the real-FFT split is folded into step 2)
//   2. Hilbert: Re(analytic) = 4096 x - X[0]/2 + X[4096](-1)^n/2 needs no transform; Im(analytic) is an
//      inverse REAL FFT of -i X[k] = one 4096-point complex inverse (5 FFT-4096 per frame in total)
//   3. s = analytic[2048 .. 6144); three windowed forward FFT4096 (w, w', t*w)
//   4. per-bin reassignment + ordered compaction (ascending bin), 12-byte points
// ================================================================================================
// Compile-time switches of the fused kernel (A/B-tested on MI355X; see DESIGN.md §4 and profiles/).
// Round 5: this kernel — the round-1 form of K2, OMX_OPT_KERNEL_FORM = 1 — and every variant of it are compiled into the TUNING library only
// (make TUNING=1); the product runs the tri kernel (stft4096_tri_kernels.hip) for every window of the reference.
#ifdef OMX_TUNING
template <uint32_t COLS, bool TW2_LDS_, bool TW3_REGS_, bool DUAL_, bool PINGPONG_, bool ONEBUF_ = false, int MINW = 2,
          bool RECOMPUTE_S_ = false, bool TWIN_CALC_ = false, bool PHASES_ = false,
          bool EARLY_ = false, bool PHASE_WAIT_ = false, bool FAST_REASSIGN_ = false,
          int KNOCK_ = 0>
struct K2Variant {
    // tuning only, WRONG results: leave one stage out to price it by the kernel time that disappears (1 reassignment +
    // compaction + stores, 2 third transform, 3 paired transforms, 4 inverse transform, 5 forward transform)
    static constexpr int KNOCK = KNOCK_;
    static constexpr bool FAST_REASSIGN = FAST_REASSIGN_;  // branch-free reassignment with v_rcp_f32
    static constexpr bool PHASE_WAIT = PHASE_WAIT_;  // tuning build: every phase mark drains vmcnt / lgkmcnt first
    static constexpr bool EARLY = EARLY_;          // table / ring loads issued one transform ahead of their use (needs DUAL)
    static constexpr bool PHASES = PHASES_;        // tuning build: thread 0 accumulates shader-clock cycles per phase
    static constexpr bool TWIN_CALC = TWIN_CALC_;  // t*w rebuilt from w in registers (:601-608) instead of a third table
    static constexpr bool RECOMPUTE_S = RECOMPUTE_S_;  // rebuild the analytic slice per windowed FFT instead of holding it
    static constexpr bool ONEBUF = ONEBUF_;        // one 34 KiB FFT buffer + 16 KiB imag[] (53 KiB -> 3 workgroups per CU)
    static constexpr int MIN_WAVES = MINW;         // __launch_bounds__ waves per SIMD
    static constexpr uint32_t COLS_PER_WG = COLS;  // consecutive columns of one stream per workgroup
    static constexpr bool TW2_LDS = TW2_LDS_;      // pass-2 twiddles from a 2 KiB LDS table
    static constexpr bool TW3_REGS = TW3_REGS_;    // pass-3 twiddles resident in VGPRs
    static constexpr bool DUAL = DUAL_;            // window and derivative-window FFTs run together
    static constexpr bool PINGPONG = PINGPONG_;    // single transforms alternate between the two LDS buffers
};

// tuning only (K2Variant::PHASES): cycles spent by thread 0 of every workgroup between phase marks, barrier waits included
__device__ unsigned long long g_k2_phase_cycles[K2_PHASES];

template <class V>
__global__ __launch_bounds__(256, V::MIN_WAVES) void stft_reassigned_4096_kernel(StftFastArgs a) {
    static_assert(!V::ONEBUF || (!V::DUAL && !V::PINGPONG), "one buffer: sequential in-place transforms only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* A = reinterpret_cast<v2f*>(smem_raw);
    v2f* B = A + FFT4096_LDS;                                      // ONEBUF: only its first 16 KiB exist (imag[])
    v2f* tw2_lds = B + (V::ONEBUF ? 2048 : FFT4096_LDS);           // [256] exp(-2 pi i k / 256)
    uint32_t* scan = reinterpret_cast<uint32_t*>(tw2_lds + 256);   // [9][4] wave counts
    float* hil = reinterpret_cast<float*>(scan + 36);              // X[0]/2, X[4096]/2

    const uint32_t chunks = (a.n_cols + V::COLS_PER_WG - 1) / V::COLS_PER_WG;
    uint32_t s, chunk;
    if (!block_to_stream_column(a.n_streams, chunks, s, chunk)) return;
    if constexpr (V::KNOCK == 6) {  // empty workgroups: dispatch + LDS / register allocation cost of the launch shape
        if (threadIdx.x == 0 && s == 0xFFFFFFFFu) a.counts[0] = 0;
        return;
    }
    const int j = threadIdx.x;
    const unsigned ju = threadIdx.x;  // unsigned 32-bit indices let global accesses use the SGPR-base + VGPR-offset form
    const float* ring = a.ring + (uint64_t)s * a.cap;
    const uint64_t mask = a.cap - 1;
    const uint32_t mask32 = (uint32_t)mask;  // cap <= 2^30 (checked on the host): ring offsets fit 32 bits
    const uint32_t bytemask = mask32 << 2;   // byte offsets as uint32: SGPR base + VGPR offset addressing
    const char* ring_bytes = reinterpret_cast<const char*>(ring);
    const long long last_nonzero = a.last_nonzero[s];
    const int lane = j & 63, wave = j >> 6;
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};

    using TW = TwiddleSource<V::TW2_LDS, V::TW3_REGS>;
    TW tw;
    tw.j = (unsigned)j;
    tw.tw3_global = a.tw4096;
    tw.tw2 = V::TW2_LDS ? tw2_lds : a.tw256;
    if constexpr (V::TW3_REGS) {
#pragma unroll
        for (int t = 1; t < 16; ++t) tw.tw3[t - 1] = a.tw4096[ju * (unsigned)t];
    }
    if constexpr (V::TW2_LDS) {
        tw2_lds[j] = a.tw256[ju];
        // no barrier of its own: the table is first read in pass 2 of the first transform, behind that transform's pass-1
        // barrier — a barrier here would put the twiddle loads and the ring loads of the window in two serial round trips
        if constexpr (V::COLS_PER_WG > 1 || !V::PINGPONG) __syncthreads();
    }

    long long phase_t = 0;
    if constexpr (V::PHASES) phase_t = clock64();
    auto mark = [&](int i) {
        if constexpr (V::PHASES) {
            if constexpr (V::PHASE_WAIT) {  // charge every outstanding load to the phase that issued it
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j == 0) {
                const long long now = clock64();
                atomicAdd(&g_k2_phase_cycles[i], (unsigned long long)(now - phase_t));
                phase_t = now;
            }
        }
    };
    const uint32_t col_end = min(stft_cols(a, s), (chunk + 1) * V::COLS_PER_WG);  // ragged banks: this stream's own column count
    const uint64_t tail_s = stft_tail(a, s);
    for (uint32_t col = chunk * V::COLS_PER_WG; col < col_end; ++col) {
        const uint64_t p0 = tail_s + (uint64_t)col * a.hop;  // absolute position of this window's first sample
        uint32_t* count_out = a.counts + (uint64_t)s * a.n_cols + col;
        const float *win = a.window, *dwin = a.dwindow, *twin = a.twindow, *bnorm = a.bin_norm;
        const v2f* tw8192 = a.tw8192;
        if constexpr (V::COLS_PER_WG > 1) {
            // The tables are thread-invariant; without this the compiler hoists ~80 loads out of the column
            // loop and spills.  Re-reading them per column from L1/L2 is the cheaper side of that trade.
            asm volatile("" : "+s"(win), "+s"(dwin), "+s"(twin), "+s"(bnorm), "+s"(tw8192));
        }
        // silent fast path (:307-316): no non-zero sample at or after the front of the pending buffer
        if (last_nonzero < (long long)p0) {
            if (j == 0) *count_out = 0;
            continue;
        }

        // ---- 1. packed real FFT of the 8192-sample window ----------------------------------------------
        v2f v[16];
        const uint32_t p32 = (uint32_t)p0;  // (p0 + i) & mask == (p32 + i) & mask32
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
        if constexpr (V::COLS_PER_WG > 1) __syncthreads();  // previous column may still be reading A / B / scan
        v2f w8[16];  // EARLY: exp(-2 pi i k / 8192), k = j + 256 t, in flight during the forward transform
        if constexpr (V::EARLY) {
#pragma unroll
            for (int t = 0; t < 16; ++t) w8[t] = tw8192[ju + 256u * (unsigned)t];
        }
        if constexpr (V::PHASES) __syncthreads();
        mark(0);  // setup + window load
        if constexpr (V::KNOCK != 5) fft4096t<false, V::PINGPONG>(v, A, B, j, tw);  // v[t] = Zf[j + 256 t]; last read: B (pingpong) or A
        mark(1);  // forward packed FFT

        // ---- 2. Hilbert transform with ONE half-length inverse ----------------------------------------
        // analytic[n] = sum_{k=1..4096} X[k] e^{+2 pi i k n / 8192} (X[0] dropped, no x2, unnormalised; :546-557).
        // Its real part needs no transform:  Re = 4096 x[n] - X[0]/2 + X[4096] (-1)^n / 2.
        // Its imaginary part is half of the real sequence with spectrum V[k] = -i X[k] (0 < k < 4096), an inverse
        // REAL FFT = one 4096-point complex inverse of
        //     Z'[k] = ( conj(w^k) (Zf[k] + conj Zf[N-k]) - w^k (Zf[k] - conj Zf[N-k]) ) / 2,   Z'[0] = 0,
        // whose output holds (Im analytic[2m], Im analytic[2m+1]) in (re, im).  w = exp(-2 pi i / 8192).
        v2f* X = (V::PINGPONG || V::ONEBUF) ? A : B;  // exchange buffer
        v2f* Y = V::PINGPONG ? B : A;
        if constexpr (!V::PINGPONG) __syncthreads();
        if constexpr (V::KNOCK != 7) {
#pragma unroll
        for (int t = 0; t < 16; ++t) X[pad16(j + 256 * t)] = v[t];
        }
        if (j == 0) {
            hil[0] = (v[0].x + v[0].y) * 0.5f;  // X[0] / 2
            hil[1] = (v[0].x - v[0].y) * 0.5f;  // X[4096] / 2
        }
        __syncthreads();
        mark(8);  // (sub-phase) natural-order write of the forward spectrum + barrier
        v2f y[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned k = (unsigned)(j + 256 * t);
            const v2f z = v[t];
            const v2f zr = V::KNOCK == 7 ? z : X[pad16((int)((4096u - k) & 4095u))];
            const v2f sum{z.x + zr.x, z.y - zr.y};   // Zf[k] + conj Zf[N-k]  (the 1/2 lives in the twiddle table)
            const v2f dif{z.x - zr.x, z.y + zr.y};   // Zf[k] - conj Zf[N-k]
            const v2f w = V::EARLY ? w8[t] : tw8192[k];
            y[t] = cmulc(sum, w) - cmul(dif, w);
            if (t == 0 && k == 0) y[t] = v2f{0.0f, 0.0f};
        }
        const float half_x0 = hil[0], half_xn = hil[1];
        if constexpr (V::ONEBUF) __syncthreads();  // partners are read from the buffer the inverse is about to overwrite
        // Y was last read before the barrier above; X is released by pass 1's barrier (ping-pong writes it in pass 2)
        float pw[16], pdw[16], pxr[16];  // EARLY: window tables and the real part's samples, in flight during the inverse
        if constexpr (V::EARLY) {
            static_assert(!V::EARLY || (V::DUAL && !V::RECOMPUTE_S), "EARLY is written for the paired-transform form");
            const uint32_t qe = p32 + 2048u + ju;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if constexpr (V::KNOCK == 9) {
                    pw[t] = 0.5f + (float)t;
                    pdw[t] = 0.25f - (float)t;
                    pxr[t] = v[t].x;
                    continue;
                }
                pw[t] = win[ju + 256u * (unsigned)t];
                pdw[t] = dwin[ju + 256u * (unsigned)t];
                pxr[t] = *reinterpret_cast<const float*>(ring_bytes + (((qe + 256u * (unsigned)t) << 2) & bytemask));
            }
        }
        mark(2);  // Hilbert spectrum build
        if constexpr (V::KNOCK != 4) fft4096t<true, V::PINGPONG>(y, Y, X, j, tw);  // y[t] = (Im a[2m], Im a[2m+1]), m = j + 256 t
        mark(3);  // inverse FFT

        // ---- 3. gather s[i] = analytic[2048 + i], i = j + 256 t ---------------------------------------
        // ping-pong: the inverse read X (= A) last, so Y (= B) is free; in place: it read Y (= A) last, X (= B) is free
        float* imag = reinterpret_cast<float*>(B);
        if constexpr (V::KNOCK != 8) {
#pragma unroll
        for (int t = 4; t < 12; ++t) *reinterpret_cast<v2f*>(imag + 2 * (j + 256 * t - 1024)) = y[t];
        __syncthreads();
        }
        const float parity = (j & 1) ? -half_xn : half_xn;  // (-1)^n, n = 2048 + i has the parity of j
        const uint32_t q0 = p32 + 2048u + ju;
        auto analytic = [&](int t) -> v2f {  // s[j + 256 t]
            const float xr = V::EARLY ? pxr[t] : ring[(q0 + 256u * (unsigned)t) & mask32];
            return v2f{4096.0f * xr - half_x0 + parity, V::KNOCK == 8 ? y[t].x + y[t].y : imag[j + 256 * t]};
        };
        v2f bb[9], bd[9], bt[9];
        float pn[9];  // EARLY: bin normalisation
        if constexpr (V::RECOMPUTE_S) {
            static_assert(V::ONEBUF, "RECOMPUTE_S keeps imag[] alive, which needs the separate imag region");
            v2f vv[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const v2f sa = analytic(t);
                const float w = win[j + 256 * t];
                vv[t] = v2f{sa.x * w, sa.y * w};
            }
            fft4096t<false, false>(vv, A, B, j, tw);
#pragma unroll
            for (int t = 0; t < 9; ++t) bb[t] = vv[t];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const v2f sa = analytic(t);
                const float w = dwin[j + 256 * t];
                vv[t] = v2f{sa.x * w, sa.y * w};
            }
            __syncthreads();
            fft4096t<false, false>(vv, A, B, j, tw);
#pragma unroll
            for (int t = 0; t < 9; ++t) bd[t] = vv[t];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const v2f sa = analytic(t);
                const float w = twin[j + 256 * t];
                vv[t] = v2f{sa.x * w, sa.y * w};
            }
            __syncthreads();
            fft4096t<false, false>(vv, A, B, j, tw);
#pragma unroll
            for (int t = 0; t < 9; ++t) bt[t] = vv[t];
        } else {
        v2f sv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) sv[t] = analytic(t);
        __syncthreads();

        // ---- three windowed FFTs; keep bins j + 256 t (t < 8) and bin 2048 (thread 0, t = 8) ----------
        if constexpr (V::DUAL) {
            v2f vb[16], vd[16], vt[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float w = V::EARLY ? pw[t] : win[j + 256 * t], dw = V::EARLY ? pdw[t] : dwin[j + 256 * t];
                vb[t] = v2f{sv[t].x * w, sv[t].y * w};
                vd[t] = v2f{sv[t].x * dw, sv[t].y * dw};
                if constexpr (V::EARLY) {
                    const float wt = ((float)(j + 256 * t) - 2047.5f) * w;  // compute_time_weighted (:601-608), same rounding
                    vt[t] = v2f{sv[t].x * wt, sv[t].y * wt};
                }
            }
            mark(4);  // analytic gather + windowing
            if constexpr (V::KNOCK != 3) fft4096t_dual<false>(vb, vd, A, B, j, tw);
            mark(5);  // dual FFT
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                bb[t] = vb[t];
                bd[t] = vd[t];
            }
            if constexpr (!V::EARLY) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float w = V::TWIN_CALC ? ((float)(j + 256 * t) - 2047.5f) * win[j + 256 * t] : twin[j + 256 * t];
                    vt[t] = v2f{sv[t].x * w, sv[t].y * w};
                }
            } else {
#pragma unroll
                for (int t = 0; t < 9; ++t) pn[t] = bnorm[(t < 8 || j == 0) ? ju + 256u * (unsigned)t : 0u];  // in flight during the third transform
            }
            __syncthreads();  // the dual transform's last pass still reads A and B
            if constexpr (V::KNOCK != 2) fft4096t<false, V::PINGPONG>(vt, A, B, j, tw);
            mark(6);  // time-weighted FFT
#pragma unroll
            for (int t = 0; t < 9; ++t) bt[t] = vt[t];
        } else {
            v2f vv[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float w = win[j + 256 * t];
                vv[t] = v2f{sv[t].x * w, sv[t].y * w};
            }
            fft4096t<false, V::PINGPONG>(vv, A, B, j, tw);
#pragma unroll
            for (int t = 0; t < 9; ++t) bb[t] = vv[t];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float w = dwin[j + 256 * t];
                vv[t] = v2f{sv[t].x * w, sv[t].y * w};
            }
            if constexpr (!V::PINGPONG) __syncthreads();
            fft4096t<false, V::PINGPONG>(vv, A, B, j, tw);
#pragma unroll
            for (int t = 0; t < 9; ++t) bd[t] = vv[t];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float w = twin[j + 256 * t];
                vv[t] = v2f{sv[t].x * w, sv[t].y * w};
            }
            if constexpr (!V::PINGPONG) __syncthreads();
            fft4096t<false, V::PINGPONG>(vv, A, B, j, tw);
#pragma unroll
            for (int t = 0; t < 9; ++t) bt[t] = vv[t];
        }
        }

        // ---- 4. reassignment + ordered compaction -------------------------------------------------------
        if constexpr (V::KNOCK == 1) {  // keep the transforms alive: one checksum word per thread instead of the points
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc += bb[t].x + bd[t].y + bt[t].x + pn[t];
            reinterpret_cast<float*>(a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride)[j] = acc;
            if (j == 0) *count_out = 0;
            continue;
        }
        omx_spectrogram_point pts[9];
        unsigned long long masks[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint32_t bin = (uint32_t)(j + 256 * t);
            bool keep = false;
            if constexpr (V::FAST_REASSIGN) {
                keep = reassign_bin_fast(bin, bb[t], bd[t], bt[t], pn[t], rc, pts[t]) && (t < 8 || j == 0);
            } else {
                if (t < 8 || j == 0) keep = reassign_bin(bin, bb[t], bd[t], bt[t], V::EARLY ? pn[t] : bnorm[bin], rc, pts[t]);
            }
            masks[t] = __ballot(keep);
            if (lane == 0) scan[t * 4 + wave] = (uint32_t)__popcll(masks[t]);
        }
        mark(9);   // (sub-phase) reassignment arithmetic + ballots
        __syncthreads();
        mark(10);  // (sub-phase) barrier behind the wave counts
        omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
        // all 36 wave counts first (nine 16-byte LDS reads in flight together), then the stores: read-wait-store per bin row
        // exposed the LDS latency nine times
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
        if constexpr (V::PHASES) __syncthreads();
        mark(7);  // reassignment + compaction + stores
    }
}

#ifdef OMX_TUNING  // rejected design, kept as the documented negative result in the tuning build only
// ================================================================================================
// K2W: the same fused pipeline with ONE WAVEFRONT PER FRAME (4 frames per 256-thread workgroup, no s_barrier).
// Each lane keeps 64 complex values; a 4096-point FFT is two radix-64 passes with one wave-private LDS exchange
// (half the LDS traffic of the radix-16 form).  The inverse transform is conj(FFT(conj(.))).  Registers instead of
// occupancy: ~400 VGPRs, one wave per SIMD, latency hidden by the straight-line instruction stream itself.
// ================================================================================================
constexpr uint32_t K2W_FRAMES_PER_WG = 4;

__global__ __launch_bounds__(256, 1) void stft_reassigned_4096_wave_kernel(StftFastArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2f* buf = reinterpret_cast<v2f*>(smem_raw) + wave * FFT4096W_LDS;
    const uint32_t chunks = (a.n_cols + K2W_FRAMES_PER_WG - 1) / K2W_FRAMES_PER_WG;
    uint32_t s, chunk;
    if (!block_to_stream_column(a.n_streams, chunks, s, chunk)) return;
    const uint32_t col = chunk * K2W_FRAMES_PER_WG + (uint32_t)wave;
    if (col >= a.n_cols) return;  // no workgroup barriers anywhere: a wave may leave early
    const float* ring = a.ring + (uint64_t)s * a.cap;
    const uint64_t mask = a.cap - 1;
    const uint64_t p0 = a.tail + (uint64_t)col * a.hop;
    uint32_t* count_out = a.counts + (uint64_t)s * a.n_cols + col;
    if (a.last_nonzero[s] < (long long)p0) {  // silent fast path (:307-316)
        if (lane == 0) *count_out = 0;
        return;
    }
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};

    // ---- 1. packed real FFT of the 8192-sample window ----------------------------------------------
    v2f v[64];
    if ((p0 & 1ull) == 0) {
#pragma unroll
        for (int t = 0; t < 64; ++t) v[t] = *reinterpret_cast<const v2f*>(ring + ((p0 + 2u * (uint32_t)(lane + 64 * t)) & mask));
    } else {
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            const uint64_t q = p0 + 2u * (uint32_t)(lane + 64 * t);
            v[t] = v2f{ring[q & mask], ring[(q + 1) & mask]};
        }
    }
    fft4096_wave(v, buf, lane, a.tw4096);  // Zf[lane + 64 t] = v[DFT64_OUT(t)]

    // ---- 2. Hilbert: Z'[k] from Zf[k], Zf[N-k] (see the radix-16 kernel for the derivation); inverse = conj FFT conj
#pragma unroll
    for (int t = 0; t < 64; ++t) buf[lane + 65 * t] = v[DFT64_OUT(t)];  // natural order, pad64(lane + 64 t)
    const float z0x = __shfl(v[DFT64_OUT(0)].x, 0), z0y = __shfl(v[DFT64_OUT(0)].y, 0);
    const float half_x0 = (z0x + z0y) * 0.5f, half_xn = (z0x - z0y) * 0.5f;  // X[0]/2, X[4096]/2
    __builtin_amdgcn_wave_barrier();
    // own Zf[k] is re-read from LDS in natural order so the 64 result registers can be reused for Z'
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = 16 * c + i;
            const unsigned k = (unsigned)(lane + 64 * t);
            const v2f z = buf[lane + 65 * t];
            const unsigned kr = (4096u - k) & 4095u;
            const v2f zr = buf[kr + (kr >> 6)];
            const v2f sum{z.x + zr.x, z.y - zr.y};   // Zf[k] + conj Zf[N-k]  (the 1/2 lives in the twiddle table)
            const v2f dif{z.x - zr.x, z.y + zr.y};   // Zf[k] - conj Zf[N-k]
            const v2f w = a.tw8192[k];
            const v2f y = cmulc(sum, w) - cmul(dif, w);
            v[t] = (k == 0) ? v2f{0.0f, 0.0f} : v2f{y.x, -y.y};  // conj(Z'[k])
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    fft4096_wave(v, buf, lane, a.tw4096);  // conj of it = (Im a[2m], Im a[2m+1]), m = lane + 64 t

    // ---- 3. redistribute the imaginary parts: imag[t] = Im analytic[2048 + lane + 64 t] ---------------
    float* fbuf = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int t = 16; t < 48; ++t) {
        const v2f o = v[DFT64_OUT(t)];
        *reinterpret_cast<v2f*>(fbuf + 2 * (lane + 64 * (t - 16))) = v2f{o.x, -o.y};
    }
    __builtin_amdgcn_wave_barrier();
    float imag[64];
#pragma unroll
    for (int t = 0; t < 64; ++t) imag[t] = fbuf[lane + 64 * t];
    __builtin_amdgcn_wave_barrier();
    const float parity = (lane & 1) ? -half_xn : half_xn;  // (-1)^n, n = 2048 + i has the parity of the lane
    const uint64_t q0 = p0 + 2048u + (unsigned)lane;

    // ---- three windowed FFTs; lane keeps bins lane + 64 t (t < 32) and lane 0 also bin 2048 ------------
    v2f bb[33], bd[33];
#pragma nounroll
    for (int q = 0; q < 3; ++q) {
        const float* win = q == 0 ? a.window : (q == 1 ? a.dwindow : a.twindow);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int t = 16 * c + i;
                const float xr = ring[(q0 + 64u * (unsigned)t) & mask];
                const float w = win[lane + 64 * t];
                const float re = 4096.0f * xr - half_x0 + parity;
                v[t] = v2f{re * w, imag[t] * w};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        fft4096_wave(v, buf, lane, a.tw4096);
        if (q == 0) {
#pragma unroll
            for (int t = 0; t < 33; ++t) bb[t] = v[DFT64_OUT(t)];
        } else if (q == 1) {
#pragma unroll
            for (int t = 0; t < 33; ++t) bd[t] = v[DFT64_OUT(t)];
        }
    }

    // ---- 4. reassignment + ordered compaction (wave-local: bins ascend with t, then with the lane) ------
    omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
    uint32_t running = 0;
#pragma unroll
    for (int t = 0; t < 33; ++t) {
        const uint32_t bin = (uint32_t)(lane + 64 * t);
        omx_spectrogram_point p;
        bool keep = false;
        if (t < 32 || lane == 0) keep = reassign_bin(bin, bb[t], bd[t], v[DFT64_OUT(t)], a.bin_norm[bin], rc, p);
        const unsigned long long m = __ballot(keep);
        if (keep) out[running + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = p;
        running += (uint32_t)__popcll(m);
    }
    if (lane == 0) *count_out = running;
}

static void launch_k2_wave(const StftFastArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)K2W_FRAMES_PER_WG * FFT4096W_LDS * sizeof(v2f);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_4096_wave_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const uint32_t chunks = (a.n_cols + K2W_FRAMES_PER_WG - 1) / K2W_FRAMES_PER_WG;
    hipLaunchKernelGGL(stft_reassigned_4096_wave_kernel, dim3(stream_column_grid(a.n_streams, chunks)), dim3(256), lds, stream, a);
}
#endif  // OMX_TUNING

template <class V>
static void launch_k2_variant(const StftFastArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)((V::ONEBUF ? FFT4096_LDS + 2048 : 2 * FFT4096_LDS) + 256) * sizeof(v2f) + 9 * 4 * sizeof(uint32_t) +
                       2 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_4096_kernel<V>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    size_t pad = 0;
#ifdef OMX_TUNING
    // OMX_K2_LDS_PAD (bytes): occupancy experiment only — a larger LDS request lowers the workgroups per CU
    static const size_t env_pad = [] {
        const char* e = tuning_env("OMX_K2_LDS_PAD");
        return e ? (size_t)atol(e) : (size_t)0;
    }();
    pad = env_pad;
    if (pad)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_reassigned_4096_kernel<V>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + pad));
#endif
    const uint32_t chunks = (a.n_cols + V::COLS_PER_WG - 1) / V::COLS_PER_WG;
    hipLaunchKernelGGL(stft_reassigned_4096_kernel<V>, dim3(stream_column_grid(a.n_streams, chunks)), dim3(256), lds + pad, stream, a);
}

unsigned long long* k2_phase_buffer() {
    void* p = nullptr;
    OMX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_k2_phase_cycles)));
    return static_cast<unsigned long long*>(p);
}
void k2_phase_cycles(unsigned long long out[K2_PHASES], bool reset) {
    OMX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k2_phase_cycles), K2_PHASES * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long zero[K2_PHASES] = {};
        OMX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_k2_phase_cycles), zero, sizeof(zero)));
    }
}

using K2Default = K2Variant<1, true, true, true, true, false, 2, false, true, false, true>;
#endif  // OMX_TUNING (the round-1 kernel and its variants)

int stft_reassigned_4096_transforms_per_frame() { return 4; }

void launch_stft_reassigned_4096(const StftFastArgs& a, int form, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return;
#ifndef OMX_TUNING
    // product: one form — three workgroups per CU, every window of the reference (1 ... 4 cosine terms) applied on the bins.  Forms 1
    // (round 1) and 2 (round 2, pair kernel) are refused by omx_spectrogram_bank_set_option; they live in the tuning library.
    (void)form;
    launch_stft_reassigned_4096_tri(a, stream);
#else
    if (form == 0 && a.win_terms == 2) {  // 50 / 51: the structural alternatives of stft4096_swz_kernels.hip (same columns, slower)
        static const int swz_variant = [] {
            const char* e = getenv("OMX_K2_VARIANT");
            return e ? atoi(e) : 0;
        }();
        if (swz_variant == 50) { launch_stft_reassigned_4096_swz_pair(a, stream); return; }
        if (swz_variant == 51) { launch_stft_reassigned_4096_col(a, stream); return; }
    }
    {
        static const int env_form = [] {  // tuning build: OMX_K2_FORM = 2 pins the pair kernel (tools/k2_forms.sh)
            const char* e = tuning_env("OMX_K2_FORM");
            return e ? atoi(e) : -1;
        }();
        const int f = (form == 0 && env_form >= 0) ? env_form : form;
        // round 4: three workgroups per CU, every window of the reference applied on the bins (stft4096_tri_kernels.hip)
        if (f == 0 && a.cos_terms >= 1 && a.cos_terms <= 4) { launch_stft_reassigned_4096_tri(a, stream); return; }
        // round 2: two workgroups per CU, two LDS buffers (Hann / Hamming)
        if (f == 2 && a.win_terms == 2) { launch_stft_reassigned_4096_pair(a, stream); return; }
    }
    // Tuning build only (make TUNING=1 -> libomx_hip_tuning.so, loaded through OMX_HIP_LIB): OMX_K2_VARIANT selects an A/B
    // build of the kernel.  100 / 1 / 2 / 3 / 12 / 13 / 14 / 20 compute the same columns as the default; 7 / 8 / 9 add phase
    // timing; 41-49 are KNOCK-OUT builds that leave one stage out and return WRONG columns (pricing a stage by the kernel
    // time that disappears).  None of this is compiled into the product library.
    // Measured on MI355X, 65 536 frames per launch (kernel ms): 100 -> 2.54, 1 -> 2.57, 2 -> 2.43, 3 -> 2.34,
    // 12 -> 2.22, default (12 + loads issued one transform ahead, t*w rebuilt in registers, 32-bit offsets) -> 2.10;
    // persistent multi-column loops (4.5 ms) and 3-workgroup/CU single-buffer forms (2.8-5.6 ms) lost to register spills
    // and were removed except variant 13, kept as the documented negative result.
    static const int variant = [] {
        const char* e = getenv("OMX_K2_VARIANT");
        return e ? atoi(e) : 0;
    }();
    switch (variant) {                                  //     cols tw2lds tw3reg dual  pingpong onebuf minw recompute
        case 100: launch_k2_variant<K2Variant<1, false, false, true, false>>(a, stream); return;  // round-1 first form
        case 1: launch_k2_variant<K2Variant<1, false, false, true, true>>(a, stream); return;
        case 2: launch_k2_variant<K2Variant<1, true, false, true, false>>(a, stream); return;
        case 3: launch_k2_variant<K2Variant<1, false, true, true, false>>(a, stream); return;
        case 13: launch_k2_variant<K2Variant<1, true, true, false, false, true, 3, true>>(a, stream); return;
        case 20: launch_k2_wave(a, stream); return;  // one wavefront per frame
        case 9: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, true, true>>(a, stream); return;
        case 41: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, false, true, false, false, 1>>(a, stream); return;
        case 42: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, false, true, false, false, 2>>(a, stream); return;
        case 43: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, false, true, false, false, 3>>(a, stream); return;
        case 44: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, false, true, false, false, 4>>(a, stream); return;
        case 45: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, false, true, false, false, 5>>(a, stream); return;
        case 47: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, false, true, false, false, 7>>(a, stream); return;
        case 48: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, false, true, false, false, 8>>(a, stream); return;
        case 49: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, false, true, false, false, 9>>(a, stream); return;
        case 46: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, false, true, false, false, 6>>(a, stream); return;
        case 14: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, false, true, false, true>>(a, stream); return;
        case 8: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, true, true, true>>(a, stream); return;
        case 7: launch_k2_variant<K2Variant<1, true, true, true, true, false, 2, false, true, true, false>>(a, stream); return;
        case 12: launch_k2_variant<K2Variant<1, true, true, true, true>>(a, stream); return;  // default until the early-load form
        default: break;
    }
    launch_k2_variant<K2Default>(a, stream);
#endif  // OMX_TUNING
}

// ================================================================================================
// Generic spectrogram kernel: any power-of-two W, F = W * zp, H = next_pow2(2W); persistent
// workgroups with a global-memory workspace each; operation order follows the reference loop.
// ================================================================================================
__global__ __launch_bounds__(256) void stft_generic_kernel(StftGenericArgs a) {
    __shared__ uint32_t scan[4];
    __shared__ uint32_t running_sh;
    __shared__ float mean_sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char generic_smem[];
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    const uint64_t total = (uint64_t)a.n_streams * a.n_cols;
    // the working set (H + 3F complex values, or F for classic columns) lives in LDS when it fits one CU — same radix-2
    // butterflies in the same order, so the output stays bit-identical to the global-workspace form
    v2f* ws = a.workspace ? a.workspace + (uint64_t)blockIdx.x * a.workspace_stride : reinterpret_cast<v2f*>(generic_smem);
    const uint32_t bins = a.fft_size / 2 + 1;
    const ReassignConsts rc{a.bin_hz, a.max_hz, a.inv_2pi, a.inv_hop, a.latency_hops};

    for (uint64_t item = blockIdx.x; item < total; item += gridDim.x) {
        const uint32_t s = (uint32_t)(item / a.n_cols), col = (uint32_t)(item % a.n_cols);
        if (a.cols && col >= a.cols[s]) continue;  // ragged banks: past this stream's own column count (workgroup-uniform)
        const uint64_t p0 = (a.tails ? a.tails[s] : a.tail) + (uint64_t)col * a.hop;
        const float* ring = a.ring + (uint64_t)s * a.cap;
        const uint64_t mask = a.cap - 1;
        const bool silent = a.last_nonzero[s] < (long long)p0;
        __syncthreads();
        if (a.reassign) {
            omx_spectrogram_point* out = a.points + ((uint64_t)s * a.n_cols + col) * a.column_stride;
            uint32_t* count_out = a.counts + (uint64_t)s * a.n_cols + col;
            if (silent) {
                if (tid == 0) *count_out = 0;
                continue;
            }
            v2f* hil = ws;
            v2f* spec = ws + a.hilbert_len;
            for (uint32_t i = tid; i < a.hilbert_len; i += nt) hil[i] = v2f{ring[(p0 + i) & mask], 0.0f};
            fft_radix2(hil, a.hilbert_len, a.log_hilbert, a.tw_hilbert, false, tid, nt);
            for (uint32_t i = tid; i < a.hilbert_len; i += nt)
                if (i == 0 || i > a.hilbert_len / 2) hil[i] = v2f{0.0f, 0.0f};  // :554-555
            fft_radix2(hil, a.hilbert_len, a.log_hilbert, a.tw_hilbert, true, tid, nt);
            const uint32_t center = (a.hilbert_len - a.window_size) / 2;
            for (uint32_t i = tid; i < a.fft_size; i += nt) {  // :559-567 x3
                v2f b{0.0f, 0.0f}, d{0.0f, 0.0f}, t{0.0f, 0.0f};
                if (i < a.window_size) {
                    const v2f z = hil[center + i];
                    const float w = a.window[i], dw = a.dwindow[i], tw = a.twindow[i];
                    b = v2f{z.x * w, z.y * w};
                    d = v2f{z.x * dw, z.y * dw};
                    t = v2f{z.x * tw, z.y * tw};
                }
                spec[i] = b;
                spec[a.fft_size + i] = d;
                spec[2 * a.fft_size + i] = t;
            }
            for (int q = 0; q < 3; ++q)
                fft_forward_any(spec + (uint64_t)q * a.fft_size, a.fft_size, a.log_fft, a.tw_fft, spec + 3ull * a.fft_size, a.blu, tid, nt);
            if (tid == 0) running_sh = 0;
            __syncthreads();
            for (uint32_t base = 0; base < bins; base += nt) {  // ordered compaction, ascending bins
                const uint32_t i = base + tid;
                omx_spectrogram_point p;
                bool keep = false;
                if (i < bins)
                    keep = reassign_bin(i, spec[i], spec[a.fft_size + i], spec[2 * a.fft_size + i], a.bin_norm[i], rc, p);
                const unsigned long long m = __ballot(keep);
                const unsigned lane = tid & 63, wave = tid >> 6;
                if (lane == 0) scan[wave] = (uint32_t)__popcll(m);
                __syncthreads();
                uint32_t before = running_sh, all = 0;
                for (unsigned w = 0; w < (nt + 63) / 64; ++w) {
                    if (w < wave) before += scan[w];
                    all += scan[w];
                }
                if (keep) out[before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = p;
                __syncthreads();
                if (tid == 0) running_sh += all;
                __syncthreads();
            }
            if (tid == 0) *count_out = running_sh;
        } else {
            uint16_t* out = a.codes + ((uint64_t)s * a.n_cols + col) * a.column_stride;
            if (silent) {
                const uint16_t floor_code = pack_classic_db_dev(-140.0f);
                for (uint32_t i = tid; i < bins; i += nt) out[i] = floor_code;
                continue;
            }
            // window.rs:66-88: mean is the reference's sequential f32 sum (kept sequential here: this is
            // the generic, order-faithful kernel)
            if (tid == 0) {
                float sum = -0.0f;
                for (uint32_t i = 0; i < a.window_size; ++i) sum = sum + ring[(p0 + i) & mask];
                mean_sh = sum / (float)a.window_size;
            }
            __syncthreads();
            const float mean = mean_sh;
            for (uint32_t i = tid; i < a.fft_size; i += nt) {
                float x = 0.0f;
                if (i < a.window_size) x = (ring[(p0 + i) & mask] - mean) * a.window[i];
                ws[i] = v2f{x, 0.0f};
            }
            fft_forward_any(ws, a.fft_size, a.log_fft, a.tw_fft, ws + a.fft_size, a.blu, tid, nt);
            for (uint32_t i = tid; i < bins; i += nt) {  // :369-379
                const v2f c = ws[i];
                out[i] = pack_classic_db_dev(power_to_db_dev((c.x * c.x + c.y * c.y) * a.bin_norm[i], -140.0f));
            }
        }
    }
}

void launch_stft_generic(const StftGenericArgs& a, uint32_t n_workgroups, hipStream_t stream) {
    if (a.n_cols == 0 || a.n_streams == 0) return;
    const size_t lds = a.workspace ? 0 : (size_t)a.workspace_stride * sizeof(v2f);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL(stft_generic_kernel, dim3(n_workgroups), dim3(256), lds, stream, a);
}

// ================================================================================================
// compute_derivative_spectral (spectrogram/processor.rs:569-599) on the device, once per config.
// ================================================================================================
__global__ __launch_bounds__(256) void derivative_window_kernel(const float* window, uint32_t n, uint32_t logn,
                                                                const v2f* tw, v2f* scratch, float* out) {
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t i = tid; i < n; i += nt) scratch[i] = v2f{window[i], 0.0f};
    fft_radix2(scratch, n, logn, tw, false, tid, nt);
    const float scale = 6.28318530717958647692f / (float)n;
    const uint32_t half = n / 2;
    for (uint32_t k = tid; k < n; k += nt) {
        v2f b = scratch[k];
        if (k == 0 || k == half) b = v2f{0.0f, 0.0f};  // n is a power of two >= 2 here, so even
        if (k >= 1) {
            const float omega = scale * ((float)k - (k > half ? (float)n : 0.0f));
            b = v2f{-omega * b.y, omega * b.x};
        }
        scratch[k] = b;
    }
    fft_radix2(scratch, n, logn, tw, true, tid, nt);
    const float inv_n = 1.0f / (float)n;
    for (uint32_t i = tid; i < n; i += nt) out[i] = scratch[i].x * inv_n;
}

void launch_derivative_window(const float* window, uint32_t n, const void* tw, void* scratch, float* out,
                              hipStream_t stream) {
    hipLaunchKernelGGL(derivative_window_kernel, dim3(1), dim3(256), 0, stream, window, n, log2_exact(n),
                       reinterpret_cast<const v2f*>(tw), reinterpret_cast<v2f*>(scratch), out);
}

}  // namespace omx
