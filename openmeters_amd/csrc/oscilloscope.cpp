// Host side of the oscilloscope path: reference src/visuals/oscilloscope/processor.rs:570-767 (config
// rebuild + epoch, history sizing, trace/trigger-source routing) driving kernels K6/K7.
#include "oscilloscope.hpp"

namespace omx {

void oscilloscope_config_default(omx_oscilloscope_config* c) {  // :27-43
    std::memset(c, 0, sizeof(*c));
    c->sample_rate = kDefaultSampleRate;
    c->segment_duration = 0.02f;
    c->trigger_mode = OMX_TRIGGER_STABLE;
    c->num_cycles = 2;
    c->trigger_source = OMX_CHANNEL_MID;
    c->channel_1 = OMX_CHANNEL_MID;
    c->channel_2 = OMX_CHANNEL_NONE;
}

static bool config_eq(const omx_oscilloscope_config& a, const omx_oscilloscope_config& b) {  // derive(PartialEq)
    const bool mode_eq = a.trigger_mode == b.trigger_mode && (a.trigger_mode != OMX_TRIGGER_STABLE || a.num_cycles == b.num_cycles);
    return a.sample_rate == b.sample_rate && a.segment_duration == b.segment_duration && mode_eq &&
           a.trigger_source == b.trigger_source && a.channel_1 == b.channel_1 && a.channel_2 == b.channel_2;
}
static uint32_t trigger_kernel_len_host(float period, float rate) {  // :184-189
    return (uint32_t)std::min<size_t>(f2usize((double)std::fmax(std::round(std::fmax(rate * 0.04f, period * 2.0f)), 2.0f)), 0x7FFFFFFFu);
}
static uint32_t stable_history_frames(uint32_t max_period, uint64_t cycles, float sample_rate) {  // :761-767
    const float max_period_f = (float)max_period;
    const uint64_t max_kernel = trigger_kernel_len_host(max_period_f, sample_rate);
    const uint64_t max_tail = std::max<uint64_t>((uint64_t)max_period * std::max<uint64_t>(cycles, 1) + 1, (max_kernel + 1) / 2);
    const uint64_t max_search = f2usize((double)std::ceil(max_period_f * 1.5f));
    return (uint32_t)std::min<uint64_t>(max_kernel / 2 + max_tail + max_search + 2, 0x7FFFFFFFu);
}

OscilloscopeBank::OscilloscopeBank(const omx_oscilloscope_config& cfg, uint32_t n_streams) : n_streams_(n_streams) {
    trig_.reserve((size_t)n_streams_ * kScopeTraces);
    rebuild(cfg);
}

void OscilloscopeBank::rebuild(const omx_oscilloscope_config& cfg) {  // Self::new (:579-587)
    cfg_ = cfg;
    has_history_channels_ = false;
    for (int t = 0; t < kScopeTraces; ++t) len_[t] = 0;
    pending_unlock_ = true;
    if (ragged_ && r_pos_.ptr) {
        std::vector<uint64_t> zero((size_t)n_streams_ * kScopeTraces * 2, 0);
        r_pos_.upload(zero, last_stream_);
        for (int t = 0; t < kScopeTraces; ++t) head_[t] = 0;
    }
}

void OscilloscopeBank::clear_history() {  // :714-723
    epoch_ += 1;
    has_history_channels_ = false;
    for (int t = 0; t < kScopeTraces; ++t) len_[t] = 0;
    pending_unlock_ = true;
    if (ragged_ && r_pos_.ptr) {  // every stream's deques are emptied; their heads only matter modulo the ring
        std::vector<uint64_t> zero((size_t)n_streams_ * kScopeTraces * 2, 0);
        r_pos_.upload(zero, last_stream_);
        for (int t = 0; t < kScopeTraces; ++t) head_[t] = 0;
    }
}

long long OscilloscopeBank::debug_resume_block(uint32_t s) const {
    if (!last_capped_ || s >= n_streams_) return -1;
    uint32_t v = 0;
    OMX_HIP(hipMemcpyAsync(&v, resume_blk_.ptr + s, sizeof(v), hipMemcpyDeviceToHost, last_launch_stream_));
    OMX_HIP(hipStreamSynchronize(last_launch_stream_));
    return (long long)v;
}

void OscilloscopeBank::reset_audio() {  // :593-600 (epoch survives the snapshot reset)
    clear_history();
    ragged_ = false;  // (lengths are zero: the positions are common again)
}

void OscilloscopeBank::update_config(const omx_oscilloscope_config& cfg) {  // :752-758
    if (!config_eq(cfg_, cfg)) {
        const uint64_t epoch = epoch_ + 1;
        rebuild(cfg);
        epoch_ = epoch;
    }
}

int OscilloscopeBank::process(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels_in,
                              float sample_rate_in, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream) {  // :611-712
    if (ragged_) {
        set_last_error("oscilloscope bank: per-stream positions are in use (process_ragged); reset_audio() returns the bank to lock-step calls");
        return OMX_ERR_INVALID;
    }
    return process_impl(pcm, pcm_on_device, block_frames, n_blocks, channels_in, sample_rate_in, positions, stream, nullptr);
}

int OscilloscopeBank::process_ragged(const float* d_pcm, uint64_t block_frames, uint64_t max_blocks, const uint32_t* n_blocks,
                                     const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                     const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_oscilloscope_ragged_update* out) {
    bool any = false;
    for (uint32_t s = 0; s < n_streams_; ++s) {
        if (n_blocks[s] > max_blocks) {
            set_last_error("oscilloscope process_ragged: n_blocks[s] > max_blocks");
            return OMX_ERR_INVALID;
        }
        any = any || n_blocks[s] != 0 || (reset_mask && reset_mask[s]);
    }
    last_stream_ = stream;
    if (!any || block_frames == 0) return OMX_NONE;
    const RaggedCall rc{n_blocks, reset_mask, out};
    return process_impl(d_pcm, true, block_frames, std::max<uint64_t>(max_blocks, 1), channels, sample_rate, positions, stream, &rc);
}

// One block per capture, each of its own length: what VisualManager::ingest_samples hands OscilloscopeProcessor::process_block
// (registry.rs:396-418) when every capture has its own batcher (meter.rs:40-69: 1 ... 4 quanta per chunk, ONE call per chunk) — one
// trigger evaluation and one StableTrigger state update per chunk (oscilloscope/processor.rs:611-712).
int OscilloscopeBank::process_chunks(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask,
                                     uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                     omx_oscilloscope_ragged_update* out) {
    if (frames_capacity == 0 || frames_capacity > 0x7FFFFFFFull) {
        set_last_error("oscilloscope process_chunks: frames_capacity must be in 1 ... 2^31 - 1");
        return OMX_ERR_INVALID;
    }
    h_blocks_.resize(n_streams_);
    bool any = false;
    for (uint32_t s = 0; s < n_streams_; ++s) {
        if (frames[s] > frames_capacity) {
            set_last_error("oscilloscope process_chunks: frames[s] > frames_capacity");
            return OMX_ERR_INVALID;
        }
        h_blocks_[s] = frames[s] != 0 ? 1u : 0u;  // block.is_empty(): nothing happens
        any = any || frames[s] != 0 || (reset_mask && reset_mask[s]);
    }
    last_stream_ = stream;
    if (!any) return OMX_NONE;
    RaggedCall rc{h_blocks_.data(), reset_mask, out};
    rc.frames_v = frames;
    rc.row_frames = frames_capacity;
    // the rings are sized by the longest block a chunk call can carry, so that a later, longer chunk never has to grow them
    return process_impl(d_pcm, true, frames_capacity, 1, channels, sample_rate, positions, stream, &rc);
}

int OscilloscopeBank::process_impl(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels_in,
                                   float sample_rate_in, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                   const RaggedCall* ragged) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    last_stream_ = stream;
    if (block_frames == 0 || n_blocks == 0) return OMX_NONE;
    if (block_frames > 0x7FFFFFFFull || n_blocks > 0x7FFFFFFFull) unsupported("oscilloscope block shape beyond 2^31");
    const float sample_rate = sanitize_sample_rate(sample_rate_in);
    if (cfg_.sample_rate != sample_rate) {
        omx_oscilloscope_config c = cfg_;
        c.sample_rate = sample_rate;
        update_config(c);
    }
    if (has_history_channels_ && history_channels_ != channels) clear_history();
    has_history_channels_ = true;
    history_channels_ = channels;

    const float sr = cfg_.sample_rate;
    const uint32_t base_frames = (uint32_t)std::min<size_t>(f2usize((double)std::fmax(std::round(sr * cfg_.segment_duration), 1.0f)), 0x3FFFFFFFu);
    const uint32_t max_period = (uint32_t)f2usize((double)std::ceil(sr / 20.0f));
    const uint32_t probe_frames = std::max<uint32_t>((uint32_t)f2usize((double)std::round(sr * 0.1f)), max_period * 2);
    const uint32_t trigger_frames = cfg_.trigger_mode == OMX_TRIGGER_ZERO_CROSSING ? base_frames + max_period
                                                                                   : stable_history_frames(max_period, cfg_.num_cycles, sr);
    const uint32_t history_frames = std::max(std::max(probe_frames, base_frames), trigger_frames);
    const uint32_t trace_channels[2] = {cfg_.channel_1, cfg_.channel_2};
    const bool active[2] = {trace_channels[0] != OMX_CHANNEL_NONE, trace_channels[1] != OMX_CHANNEL_NONE};
    int matching = -1;
    for (int s = 0; s < 2; ++s)
        if (trace_channels[s] == cfg_.trigger_source) { matching = s; break; }
    if (matching >= 0 && !active[matching]) matching = -1;
    const bool separate = matching < 0 && cfg_.trigger_source != OMX_CHANNEL_NONE;
    if (cfg_.trigger_source == OMX_CHANNEL_NONE) len_[2] = 0;

    // ---- device buffers
    const uint32_t max_kernel = trigger_kernel_len_host((float)(max_period + 2), sr) + 8;
    const uint32_t fft_size = (uint32_t)next_pow2((uint64_t)probe_frames + max_period + 1);
    // Wide form (scope_fast_kernels.hip) for every call shape of the configurations whose autocorrelation is an 8192-point
    // transform and whose trigger arrays fit the LDS of one CU: every block of the call is pushed into the rings first (they
    // hold the history plus the whole call), the period estimates — a pure function of the trace — are computed for all
    // (stream, block, view) in parallel, and one workgroup per stream runs the stateful trigger pass in timeline order.
    const uint64_t total_frames = block_frames * n_blocks;
    const bool wide = fft_size == 8192 && scope_trigger_lds_bytes(max_kernel, max_period) <= 152 * 1024 && total_frames <= (1ull << 22);
    // 54.6 ... 218 kHz (88.2 / 96 / 176.4 / 192 kHz): the autocorrelation is a 16 384- or 32 768-point transform.  Pushed-first form too:
    // estimates by scope_estimate_big_kernel (LDS transforms), then the one-workgroup-per-stream kernel on those estimates (round 4;
    // before, that kernel ran a radix-2 transform in global memory per block: 21x the 48 kHz time per block at 96 kHz)
    const bool big = !wide && (fft_size == 16384 || fft_size == 32768) && total_frames <= (1ull << 22);
    const bool pushed_first = wide || big;
    const uint64_t cap = std::max<uint64_t>(  // never shrinks: call shapes may alternate
        cap_, next_pow2((uint64_t)history_frames + std::max<uint64_t>(pushed_first ? total_frames : block_frames, 4096)));
    if (ragged_ && !pushed_first && (cap != cap_ || !rings_.ptr))
        unsupported("oscilloscope process_ragged: block_frames grew beyond the ring sized at the first ragged call");
    if (cap != cap_ || !rings_.ptr) {
        DeviceBuffer<float> bigger;
        bigger.reserve((size_t)(cap * kScopeTraces * n_streams_));
        OMX_HIP(hipMemsetAsync(bigger.ptr, 0, bigger.count * sizeof(float), stream));
        if (rings_.ptr && cap_) {  // the deques keep their absolute positions, only the modulus changes
            ScopeArgs ra{};
            ra.n_streams = n_streams_;
            for (int t = 0; t < kScopeTraces; ++t) {
                ra.head[t] = head_[t];
                ra.len[t] = len_[t];
            }
            launch_scope_rehome(rings_.ptr, cap_, bigger.ptr, cap, ragged_ ? r_pos_.ptr : nullptr, ra, std::min<uint64_t>(history_frames, cap_), stream);
            OMX_HIP(hipStreamSynchronize(stream));  // the old rings are freed below
        }
        std::swap(rings_.ptr, bigger.ptr);
        std::swap(rings_.count, bigger.count);
        cap_ = cap;
    }
    if (max_kernel != max_kernel_ || !reference_.ptr) {
        max_kernel_ = max_kernel;
        reference_.reserve((size_t)n_streams_ * kScopeTraces * max_kernel);
        pending_unlock_ = true;  // reference layout changed: the stored templates are void
    }
    if (fft_size != fft_size_) {
        fft_size_ = fft_size;
        tw_fft_.upload(twiddle_table(fft_size, fft_size / 2), stream);
        if (fft_size == 8192 || fft_size == 16384 || fft_size == 32768) {  // the LDS transforms: exp(-2 pi i k / 256), exp(-2 pi i k / M), M = fft_size / 2
            tw256_.upload(twiddle_table(256, 256), stream);
            tw4096_.upload(twiddle_table(fft_size / 2, fft_size / 2), stream);
        }
    }
    const uint64_t scratch_stride = scope_scratch_floats(max_kernel, 0, probe_frames, max_period);
    scratch_.reserve((size_t)(scratch_stride * n_streams_));
    const bool fft_in_lds = (uint64_t)fft_size * sizeof(v2f) <= 64 * 1024;
    if (!fft_in_lds && !big) fft_global_.reserve((size_t)n_streams_ * fft_size * 2);
    if (pending_unlock_) {
        OMX_HIP(hipMemsetAsync(trig_.ptr, 0, trig_.count * sizeof(ScopeTriggerState), stream));
        pending_unlock_ = false;
    }
    headers_.reserve((size_t)(n_streams_ * n_blocks), host_outputs_ && n_streams_ * n_blocks <= 4096);
    samples_.reserve((size_t)n_streams_ * 2 * kScopeTarget, host_outputs_ && n_streams_ <= 4);
    const bool chunk_call = ragged && ragged->frames_v;
    const uint64_t total = chunk_call ? ragged->row_frames : block_frames * n_blocks;  // frames per row of `pcm`
    const float* d_pcm = pcm;
    if (!pcm_on_device) {
        const size_t n = (size_t)n_streams_ * total * channels;
        d_pcm = staging_.stage(pcm, n, stream);
    }

    ScopeArgs sa{};
    sa.pcm = d_pcm;
    sa.frames_total = total;
    sa.block_frames = (uint32_t)block_frames;
    sa.n_blocks = (uint32_t)n_blocks;
    sa.n_streams = n_streams_;
    sa.fmt = make_format(channels, positions);
    sa.sample_rate = sr;
    sa.trigger_mode = cfg_.trigger_mode;
    sa.num_cycles = (uint32_t)std::min<uint64_t>(cfg_.num_cycles, 0x7FFFFFFFu);
    sa.trace_channel[0] = trace_channels[0];
    sa.trace_channel[1] = trace_channels[1];
    sa.trigger_source = cfg_.trigger_source;
    sa.matching_trace = matching;
    sa.separate_source = separate ? 1 : 0;
    sa.base_frames = base_frames;
    sa.max_period = max_period;
    sa.probe_frames = probe_frames;
    sa.history_frames = history_frames;
    sa.rings = rings_.ptr;
    sa.cap = cap_;
    for (int t = 0; t < kScopeTraces; ++t) {
        sa.head[t] = head_[t];
        sa.len[t] = len_[t];
    }
    sa.trig = trig_.ptr;
    sa.reference = reference_.ptr;
    sa.scratch = scratch_.ptr;
    sa.scratch_stride = scratch_stride;
    sa.max_kernel = max_kernel;
    sa.fft_size = fft_size;
    sa.log_fft = log2_exact(fft_size);
    sa.tw_fft = reinterpret_cast<const v2f*>(tw_fft_.ptr);
    sa.fft_global = (fft_in_lds || big) ? nullptr : reinterpret_cast<v2f*>(fft_global_.ptr);
    const bool fast_acf = fft_size_ == 8192 && fft_in_lds && tw4096_.ptr;
    sa.tw256 = (fast_acf || big) ? reinterpret_cast<const v2f*>(tw256_.ptr) : nullptr;
    sa.tw4096 = (fast_acf || big) ? reinterpret_cast<const v2f*>(tw4096_.ptr) : nullptr;
    sa.lds_scratch = (fast_acf && scope_lds_scratch_bytes(max_kernel, sa.max_period, sa.probe_frames) <= 150 * 1024) ? 1u : 0u;
    sa.pre_pushed = big ? 1u : 0u;
    if (big) sa.lds_scratch = scope_locate_lds_bytes(max_kernel, sa.max_period) <= 150 * 1024 ? 2u : 0u;  // 96 kHz: 134 KiB; 192 kHz: global scratch
    sa.headers = headers_.ptr;
    sa.samples = samples_.ptr;
    static const bool phase_timing = [] {
        const char* e = tuning_env("OMX_SCOPE_PHASES");
        return e && atoi(e) != 0;
    }();
    sa.phase_timing = phase_timing ? 1u : 0u;
    sa.estimates = nullptr;
    // the wide trigger pass on capped LDS + hand-over (scope_fast_kernels.hip: launch_oscilloscope_big).  At 48 kHz the worst case fits a CU
    // and the capped form (64 KiB, tuning hook OMX_SCOPE_WIDE_CAPPED) measured 3 % slower at cfg4 and 3 ... 8 % slower at the streaming
    // cadence: the dry run and the hand-over launch cost more than the freed LDS returns
    static const bool wide_capped = tuning_env("OMX_SCOPE_WIDE_CAPPED") != nullptr;
    if (big || (wide && wide_capped)) {
        static const bool legacy = tuning_env("OMX_SCOPE_BIG_LEGACY") != nullptr;   // tuning hook: A/B against the per-stream kernel alone
        if (!(big && legacy)) {
            resume_blk_.reserve(n_streams_);
            resume_pos_.reserve((size_t)n_streams_ * kScopeTraces * 2);
            sa.resume_blk = resume_blk_.ptr;
            sa.resume_pos = resume_pos_.ptr;
        }
    }
    last_capped_ = sa.resume_blk != nullptr;
    last_launch_stream_ = stream;
    if (ragged) {
        if (!ragged_) {  // every stream starts from the bank's common positions
            std::vector<uint64_t> seed((size_t)n_streams_ * kScopeTraces * 2);
            for (uint32_t s = 0; s < n_streams_; ++s)
                for (int t = 0; t < kScopeTraces; ++t) {
                    seed[((size_t)s * kScopeTraces + t) * 2] = head_[t];
                    seed[((size_t)s * kScopeTraces + t) * 2 + 1] = len_[t];
                }
            r_pos_.upload(seed, stream);
            r_epoch_.upload(std::vector<uint64_t>(n_streams_, epoch_), stream);
            ragged_ = true;
        }
        r_staging_.upload(ragged->n_blocks, ragged->reset_mask, n_streams_, r_blocks_, r_mask_, stream, chunk_call ? ragged->frames_v : nullptr,
                          &r_frames_);
        sa.frames_v = chunk_call ? r_frames_.ptr : nullptr;
        sa.pos_v = r_pos_.ptr;
        sa.blocks_v = r_blocks_.ptr;
        sa.reset_v = r_mask_.ptr;
        sa.epoch_v = r_epoch_.ptr;
        if (pushed_first) {
            estimates_.reserve((size_t)n_streams_ * n_blocks * kScopeTraces);
            sa.estimates = estimates_.ptr;
            sa.est_view_count = 0;
            for (int t = 0; t < kScopeTraces; ++t)
                if (t < 2 ? active[t] : separate) sa.est_views[sa.est_view_count++] = (uint32_t)t;
            if (wide) launch_oscilloscope_fast(sa, stream);
            else launch_oscilloscope_big(sa, stream);
        } else {
            launch_oscilloscope(sa, stream);
        }
        OMX_HIP(hipGetLastError());
        last_blocks_ = n_blocks;
        if (ragged->out) {
            ragged->out->n_streams = n_streams_;
            ragged->out->max_blocks = n_blocks;
            ragged->out->d_n_blocks = r_blocks_.ptr;
            ragged->out->d_epochs = r_epoch_.ptr;
            ragged->out->d_headers = reinterpret_cast<const omx_oscilloscope_block_header*>(headers_.ptr);
            ragged->out->d_samples = samples_.ptr;
        }
        return OMX_PRODUCED;
    }
    if (pushed_first) {
        estimates_.reserve((size_t)n_streams_ * n_blocks * kScopeTraces);
        sa.estimates = estimates_.ptr;
        // the views a block of this call can ask an estimate for (:683-700): the linked view alone while every pushed trace holds
        // the same number of samples (lock-step pushes from a common reset: always, short of a bookkeeping surprise)
        const int linked_view = matching >= 0 ? matching : (separate ? 2 : -1);
        bool same_len = true;
        uint64_t common = ~0ull;
        for (int t = 0; t < kScopeTraces; ++t)
            if (t < 2 ? active[t] : separate) {
                if (common == ~0ull) common = len_[t];
                same_len = same_len && len_[t] == common;
            }
        sa.est_view_count = 0;
        if (linked_view >= 0 && same_len) {
            sa.est_views[sa.est_view_count++] = (uint32_t)linked_view;
        } else {
            for (int t = 0; t < kScopeTraces; ++t)
                if (t < 2 ? active[t] : separate) sa.est_views[sa.est_view_count++] = (uint32_t)t;
        }
        if (wide) launch_oscilloscope_fast(sa, stream);
        else launch_oscilloscope_big(sa, stream);
    } else {
        launch_oscilloscope(sa, stream);
    }
    OMX_HIP(hipGetLastError());

    for (int t = 0; t < kScopeTraces; ++t) {  // same bookkeeping as the kernel (:673-681)
        const bool on = t < 2 ? active[t] : separate;
        if (on) {
            head_[t] += total;
            len_[t] = std::min<uint64_t>(len_[t] + total, history_frames);
        } else {
            len_[t] = 0;
        }
    }
    last_blocks_ = n_blocks;
    return OMX_PRODUCED;
}

int OscilloscopeBank::fetch_header(uint64_t stream_index, uint64_t block, ScopeBlockHeader* dst, hipStream_t stream) {
    if (stream_index >= n_streams_ || block >= last_blocks_) {
        set_last_error("oscilloscope fetch: index out of range");
        return OMX_ERR_INVALID;
    }
    copy_out(dst, headers_.ptr + stream_index * last_blocks_ + block, sizeof(*dst), headers_.pinned, stream);
    return OMX_NONE;
}

int OscilloscopeBank::fetch_samples(uint64_t stream_index, float* dst, uint64_t count, hipStream_t stream) {
    if (stream_index >= n_streams_ || count > 2 * kScopeTarget) return OMX_ERR_INVALID;
    copy_out(dst, samples_.ptr + stream_index * 2 * kScopeTarget, count * sizeof(float), samples_.pinned, stream);
    return OMX_NONE;
}

}  // namespace omx
