// S DspBatchers with their samples resident on the device (reference src/meter.rs:27-80: one DspBatcher per capture; SURVEY §8f rank 1,
// "device-side batching").  The host batcher (batcher.cpp) is the reference's structure for ONE capture: a host vector of pending
// samples, every packet copied through it.  A service with a thousand captures would run that loop a thousand times per packet round
// and copy every sample twice on the host.  Here the packet LENGTHS stay host values (they come from the capture API, and the chunk
// plan — DspBatcher::push's integer arithmetic — is a pure function of them and of the pending counts), the SAMPLES never leave the
// device: one launch assembles the chunks of every capture, a second one keeps the remainders.
//
// DspBatcher::push (:40-69), per capture, in frames (batch = 256 and chunk = 1024 frames at 48 kHz, scaled by the rate, :20-25):
//   pending > 0:  take = min(batch - pending, n); pending + take == batch -> ONE chunk of `batch` frames (the pending ones + the first
//                 `take` of the packet), otherwise everything is appended and the push ends
//   then          ready = (n - take) / batch * batch frames go out as chunks of up to `chunk` frames, the rest is kept
// A push of S packets therefore yields ROUNDS: round r holds the r-th chunk of every capture that has one — what
// omx_capture_group_ingest_ragged takes per call (one block per capture, registry.rs:396-418).
#include <algorithm>
#include <vector>

#include "common.hpp"

namespace omx {

namespace {

struct BatcherPlan {  // one capture's push, from (pending, n) alone — the same arithmetic on the host and in the kernels
    uint32_t take;       // packet frames that complete (or extend) the pending partial batch
    uint32_t completes;  // 1: round 0 is the completed batch
    uint32_t ready;      // frames of whole batches behind `take`
    uint32_t rest;       // frames kept for the next push
};
__host__ __device__ inline uint32_t least(uint32_t x, uint32_t y) { return x < y ? x : y; }
__host__ __device__ inline BatcherPlan batcher_plan(uint32_t pending, uint32_t n, uint32_t batch) {
    BatcherPlan p{0, 0, 0, 0};
    if (pending != 0u) {
        p.take = least(batch - pending, n);
        p.completes = pending + p.take == batch ? 1u : 0u;
        if (!p.completes) {  // (then take == n: everything was appended)
            p.rest = pending + n;
            return p;
        }
    }
    p.ready = (n - p.take) / batch * batch;
    p.rest = n - p.take - p.ready;
    return p;
}
__host__ __device__ inline uint32_t batcher_rounds(const BatcherPlan& p, uint32_t chunk) { return p.completes + (p.ready + chunk - 1u) / chunk; }

struct BatcherArgs {
    const float* packets;     // [n_captures][packet_stride][channels]
    uint64_t packet_stride;   // frames
    const uint32_t* table;    // [n_captures][2]: pending frames before the push (after a clear: 0), packet frames
    float* pending;           // [n_captures][batch][channels]
    float* rounds;            // [n_rounds][n_captures][chunk][channels]
    uint32_t n_captures, channels, batch, chunk, n_rounds;
};

}  // namespace

// thread = (capture, round, float of the chunk)
__global__ __launch_bounds__(256) void batcher_rounds_kernel(BatcherArgs a) {
    const uint32_t s = blockIdx.y, r = blockIdx.z;
    const uint32_t pending = a.table[2u * s], n = a.table[2u * s + 1u];
    const BatcherPlan p = batcher_plan(pending, n, a.batch);
    if (r >= batcher_rounds(p, a.chunk)) return;
    const uint32_t C = a.channels;
    const float* packet = a.packets + (uint64_t)s * a.packet_stride * C;
    float* out = a.rounds + ((uint64_t)r * a.n_captures + s) * a.chunk * C;
    uint32_t frames, from_pending = 0u, offset;  // the chunk = from_pending frames of the pending buffer, then packet frames from `offset`
    if (p.completes && r == 0u) {
        frames = a.batch;
        from_pending = pending;
        offset = 0u;
    } else {
        const uint32_t j = r - p.completes;
        offset = p.take + j * a.chunk;
        frames = least(a.chunk, p.ready - j * a.chunk);
    }
    const uint32_t total = frames * C, lead = from_pending * C;
    const float* pend = a.pending + (uint64_t)s * a.batch * C;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u)
        out[i] = i < lead ? pend[i] : packet[(uint64_t)offset * C + (i - lead)];
}

// thread = (capture, float kept): behind the rounds kernel in the stream (it reads the pending frames this one overwrites)
__global__ __launch_bounds__(256) void batcher_keep_kernel(BatcherArgs a) {
    const uint32_t s = blockIdx.y;
    const uint32_t pending = a.table[2u * s], n = a.table[2u * s + 1u];
    const BatcherPlan p = batcher_plan(pending, n, a.batch);
    const uint32_t C = a.channels;
    const float* packet = a.packets + (uint64_t)s * a.packet_stride * C;
    float* pend = a.pending + (uint64_t)s * a.batch * C;
    const bool appended = pending != 0u && !p.completes;  // the packet went behind the pending frames, which stay
    const uint32_t dst0 = appended ? pending * C : 0u, src0 = appended ? 0u : (p.take + p.ready) * C;
    const uint32_t count = (appended ? n : p.rest) * C;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) pend[dst0 + i] = packet[src0 + i];
}

// ingest_silence (meter.rs:145-166) for S captures: only the FIRST chunk of a capture's silence can hold samples (its pending partial
// batch, completed with zeros); every later chunk is zeros.  One workgroup per capture: round 0's row, then the pending row.
// table [n_captures][4]: pending frames before, frames of the capture's first chunk (0: none), 1 when that chunk completes the pending
// batch, pending frames afterwards
__global__ __launch_bounds__(256) void batcher_silence_kernel(BatcherArgs a) {
    const uint32_t s = blockIdx.x, C = a.channels;
    const uint32_t pending = a.table[4u * s], first = a.table[4u * s + 1u], completes = a.table[4u * s + 2u], after = a.table[4u * s + 3u];
    float* pend = a.pending + (uint64_t)s * a.batch * C;
    float* out = a.rounds + (uint64_t)s * a.chunk * C;  // round 0
    const uint32_t lead = completes ? pending * C : 0u;
    for (uint32_t i = threadIdx.x; i < first * C; i += 256u) out[i] = i < lead ? pend[i] : 0.0f;
    __syncthreads();  // (the pending frames are read above and overwritten below)
    const uint32_t from = (completes || pending == 0u) ? 0u : pending * C;  // no chunk at all: the zeros go behind the pending frames
    for (uint32_t i = from + threadIdx.x; i < after * C; i += 256u) pend[i] = 0.0f;
}

namespace {

uint32_t scaled_frames(uint32_t frames_at_48k, float sample_rate) {  // meter.rs:20-25 (frames; the reference counts samples = frames x channels)
    const double v = std::round((double)frames_at_48k * (double)sample_rate / (double)kDefaultSampleRate);
    return (uint32_t)f2usize(std::fmax(v, 1.0));
}
bool same_format(const omx_audio_format& a, const omx_audio_format& b) {  // derive(PartialEq) on AudioFormat
    return a.generation == b.generation && a.sample_rate == b.sample_rate && a.channels == b.channels &&
           std::memcmp(a.positions, b.positions, sizeof(a.positions)) == 0;
}

}  // namespace

}  // namespace omx

struct omx_batcher_bank {
    bool zero_rounds = false;   // the last push was silence: rounds >= 1 share one slab of zeros
    uint32_t n_captures = 0;
    uint64_t max_packet_frames = 0;
    bool has_format = false;
    omx_audio_format format{};
    uint32_t batch = 0, chunk = 0;
    std::vector<uint32_t> pending;                     // frames, per capture
    std::vector<std::vector<uint32_t>> round_frames;   // [round][capture] of the last push
    uint32_t n_rounds = 0;
    omx::DeviceBuffer<float> d_pending, d_rounds;
    omx::DeviceBuffer<uint32_t> d_table;
    omx::BlobStaging staging;
    std::vector<uint32_t> table;
};

extern "C" {

int omx_batcher_bank_create(uint32_t n_captures, uint64_t max_packet_frames, omx_batcher_bank** out) {
    return omx::guarded([&] {
        if (!out || n_captures == 0 || max_packet_frames == 0 || max_packet_frames > 0x3FFFFFFFull) {
            omx::set_last_error("omx_batcher_bank_create: n_captures and max_packet_frames (< 2^30) must be positive");
            return (int)OMX_ERR_INVALID;
        }
        const int ready = omx::device_ready();  // (never a host fallback: the point of this bank is that the samples stay on the device)
        if (ready < 0) return ready;
        auto* b = new omx_batcher_bank();
        b->n_captures = n_captures;
        b->max_packet_frames = max_packet_frames;
        b->pending.assign(n_captures, 0u);
        *out = b;
        return (int)OMX_NONE;
    });
}
void omx_batcher_bank_destroy(omx_batcher_bank* b) { delete b; }

int omx_batcher_bank_push(omx_batcher_bank* b, const float* d_packets, uint64_t packet_stride, const uint32_t* packet_frames,
                          const uint8_t* clear_mask, const omx_audio_format* format, void* stream_v, uint32_t* n_rounds) {
    return omx::guarded([&] {
        if (!b || !format || !packet_frames || !n_rounds) {
            omx::set_last_error("omx_batcher_bank_push: null argument");
            return (int)OMX_ERR_INVALID;
        }
        omx::bind_thread_device();
        const hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
        const uint32_t S = b->n_captures, C = std::min<uint32_t>(std::max<uint32_t>(format->channels, 1), OMX_MAX_CHANNELS);
        uint32_t longest = 0;
        for (uint32_t s = 0; s < S; ++s) {
            if (packet_frames[s] > b->max_packet_frames || packet_frames[s] > packet_stride) {
                omx::set_last_error("omx_batcher_bank_push: packet_frames[s] exceeds max_packet_frames or packet_stride");
                return (int)OMX_ERR_INVALID;
            }
            longest = std::max(longest, packet_frames[s]);
        }
        if (longest != 0 && !d_packets) {
            omx::set_last_error("omx_batcher_bank_push: d_packets is null");
            return (int)OMX_ERR_INVALID;
        }
        if (b->has_format && !omx::same_format(b->format, *format)) std::fill(b->pending.begin(), b->pending.end(), 0u);  // :46-48, every capture
        b->has_format = true;
        b->format = *format;
        b->batch = omx::scaled_frames(256, format->sample_rate);   // DSP_BATCH_FRAMES_AT_48K
        b->chunk = omx::scaled_frames(1024, format->sample_rate);  // MAX_DSP_INGEST_FRAMES_AT_48K
        const uint32_t batch = b->batch, chunk = b->chunk;
        // the plan: per capture (pending, n) -> its chunks; round r = the r-th chunk of every capture that has one
        b->table.resize((size_t)S * 2);
        uint32_t rounds = 0;
        for (uint32_t s = 0; s < S; ++s) {
            if (clear_mask && clear_mask[s]) b->pending[s] = 0u;  // DspBatcher::clear of this capture (:76-79)
            b->table[2 * s] = b->pending[s];
            b->table[2 * s + 1] = packet_frames[s];
            rounds = std::max(rounds, omx::batcher_rounds(omx::batcher_plan(b->pending[s], packet_frames[s], batch), chunk));
        }
        b->round_frames.assign(rounds, std::vector<uint32_t>(S, 0u));
        for (uint32_t s = 0; s < S; ++s) {
            const omx::BatcherPlan p = omx::batcher_plan(b->pending[s], packet_frames[s], batch);
            uint32_t r = 0;
            if (p.completes) b->round_frames[r++][s] = batch;
            for (uint32_t off = 0; off < p.ready; off += chunk) b->round_frames[r++][s] = std::min(chunk, p.ready - off);
            b->pending[s] = p.rest;
        }
        b->n_rounds = rounds;
        b->zero_rounds = false;
        *n_rounds = rounds;
        if (longest == 0) return (int)OMX_NONE;  // nothing arrived anywhere: no chunk, nothing to keep
        b->d_pending.reserve((size_t)S * batch * C);  // (a format change re-sizes it; the pending counts were cleared above)
        b->d_table.reserve((size_t)S * 2);
        b->staging.upload(b->table.data(), b->table.size() * sizeof(uint32_t), b->d_table.ptr, stream);
        if (rounds) b->d_rounds.reserve((size_t)rounds * S * chunk * C);
        omx::BatcherArgs a{};
        a.packets = d_packets;
        a.packet_stride = packet_stride;
        a.table = b->d_table.ptr;
        a.pending = b->d_pending.ptr;
        a.rounds = b->d_rounds.ptr;
        a.n_captures = S;
        a.channels = C;
        a.batch = batch;
        a.chunk = chunk;
        a.n_rounds = rounds;
        if (rounds) {
            const uint32_t gx = std::min<uint32_t>((chunk * C + 255u) / 256u, 16u);
            hipLaunchKernelGGL(omx::batcher_rounds_kernel, dim3(gx, S, rounds), dim3(256), 0, stream, a);
        }
        hipLaunchKernelGGL(omx::batcher_keep_kernel, dim3(std::min<uint32_t>((batch * C + 255u) / 256u, 8u), S), dim3(256), 0, stream, a);
        OMX_HIP(hipGetLastError());
        return (int)OMX_NONE;
    });
}

int omx_batcher_bank_push_silence(omx_batcher_bank* b, const uint64_t* silence_frames, const omx_audio_format* format, void* stream_v,
                                  uint32_t* n_rounds, uint8_t* reset_out) {
    return omx::guarded([&] {
        if (!b || !format || !silence_frames || !n_rounds) {
            omx::set_last_error("omx_batcher_bank_push_silence: null argument");
            return (int)OMX_ERR_INVALID;
        }
        omx::bind_thread_device();
        const hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
        const uint32_t S = b->n_captures, C = std::min<uint32_t>(std::max<uint32_t>(format->channels, 1), OMX_MAX_CHANNELS);
        if (b->has_format && !omx::same_format(b->format, *format)) std::fill(b->pending.begin(), b->pending.end(), 0u);
        b->has_format = true;
        b->format = *format;
        b->batch = omx::scaled_frames(256, format->sample_rate);
        b->chunk = omx::scaled_frames(1024, format->sample_rate);
        const uint32_t batch = b->batch, chunk = b->chunk;
        const double lim = std::fmax(std::round(2.0 * (double)format->sample_rate), 1.0);  // MAX_SILENCE_SECONDS (:18, :151-153)
        const uint64_t limit = lim >= 18446744073709551615.0 ? UINT64_MAX : (uint64_t)lim;
        const uint64_t piece_cap = (uint64_t)4096 * OMX_MAX_CHANNELS / std::max<uint32_t>(format->channels, 1);  // the scratch's frames (:96, :159)
        std::vector<std::vector<uint32_t>> per(S);  // every capture's chunk lengths, in order
        b->table.assign((size_t)S * 4, 0u);
        uint32_t rounds = 0;
        bool any = false;
        for (uint32_t s = 0; s < S; ++s) {
            if (reset_out) reset_out[s] = 0;
            const uint32_t pending0 = b->pending[s];
            b->table[4 * s] = pending0;
            b->table[4 * s + 3] = pending0;
            if (silence_frames[s] == 0) continue;
            if (silence_frames[s] > limit) {  // batcher.reset(manager) (:154-157): the caller resets that capture's visuals
                b->pending[s] = 0;
                b->table[4 * s + 3] = 0;
                if (reset_out) reset_out[s] = 1;
                continue;
            }
            any = true;
            bool first_piece = true;
            for (uint64_t remaining = silence_frames[s]; remaining > 0;) {  // one DspBatcher::push per piece of the scratch (:160-165)
                const uint32_t piece = (uint32_t)std::min<uint64_t>(remaining, piece_cap);
                const omx::BatcherPlan p = omx::batcher_plan(b->pending[s], piece, batch);
                if (p.completes) {
                    if (first_piece && per[s].empty()) b->table[4 * s + 2] = 1u;
                    per[s].push_back(batch);
                }
                for (uint32_t off = 0; off < p.ready; off += chunk) per[s].push_back(std::min(chunk, p.ready - off));
                b->pending[s] = p.rest;
                remaining -= piece;
                first_piece = false;
            }
            b->table[4 * s + 1] = per[s].empty() ? 0u : per[s][0];
            b->table[4 * s + 3] = b->pending[s];
            rounds = std::max<uint32_t>(rounds, (uint32_t)per[s].size());
        }
        b->round_frames.assign(rounds, std::vector<uint32_t>(S, 0u));
        for (uint32_t s = 0; s < S; ++s)
            for (size_t r = 0; r < per[s].size(); ++r) b->round_frames[r][s] = per[s][r];
        b->n_rounds = rounds;
        b->zero_rounds = true;
        *n_rounds = rounds;
        if (!any) return (int)OMX_NONE;
        b->d_pending.reserve((size_t)S * batch * C);
        b->d_table.reserve((size_t)S * 4);
        b->staging.upload(b->table.data(), b->table.size() * sizeof(uint32_t), b->d_table.ptr, stream);
        b->d_rounds.reserve((size_t)2 * S * chunk * C);  // round 0, and the zeros every later round is
        OMX_HIP(hipMemsetAsync(b->d_rounds.ptr + (size_t)S * chunk * C, 0, (size_t)S * chunk * C * sizeof(float), stream));
        omx::BatcherArgs a{};
        a.table = b->d_table.ptr;
        a.pending = b->d_pending.ptr;
        a.rounds = b->d_rounds.ptr;
        a.n_captures = S;
        a.channels = C;
        a.batch = batch;
        a.chunk = chunk;
        a.n_rounds = rounds;
        hipLaunchKernelGGL(omx::batcher_silence_kernel, dim3(S), dim3(256), 0, stream, a);
        OMX_HIP(hipGetLastError());
        return (int)OMX_NONE;
    });
}

int omx_batcher_bank_round(const omx_batcher_bank* b, uint32_t r, const float** d_pcm, uint64_t* chunk_capacity, const uint32_t** frames) {
    return omx::guarded([&] {
        if (!b || r >= b->n_rounds) {
            omx::set_last_error("omx_batcher_bank_round: no such round in the last push");
            return (int)OMX_ERR_INVALID;
        }
        const uint32_t C = std::min<uint32_t>(std::max<uint32_t>(b->format.channels, 1), OMX_MAX_CHANNELS);
        const size_t slab = b->zero_rounds ? std::min<size_t>(r, 1) : r;  // (silence: every round behind the first is the same zeros)
        if (d_pcm) *d_pcm = b->d_rounds.ptr + slab * b->n_captures * b->chunk * C;
        if (chunk_capacity) *chunk_capacity = b->chunk;
        if (frames) *frames = b->round_frames[r].data();
        return (int)OMX_NONE;
    });
}

uint64_t omx_batcher_bank_pending(omx_batcher_bank* b, uint32_t s, float* dst, uint64_t cap, void* stream_v) {
    if (!b || s >= b->n_captures) return 0;
    const uint32_t C = std::min<uint32_t>(std::max<uint32_t>(b->format.channels, 1), OMX_MAX_CHANNELS);
    const uint64_t n = (uint64_t)b->pending[s] * C;
    if (dst && n && b->d_pending.ptr) {
        const int rc = omx::guarded([&] {
            omx::bind_thread_device();
            const hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
            OMX_HIP(hipMemcpyAsync(dst, b->d_pending.ptr + (size_t)s * b->batch * C, std::min<uint64_t>(n, cap) * sizeof(float), hipMemcpyDeviceToHost, stream));
            OMX_HIP(hipStreamSynchronize(stream));
            return (int)OMX_NONE;
        });
        if (rc < 0) return 0;
    }
    return n;
}

}  // extern "C"
