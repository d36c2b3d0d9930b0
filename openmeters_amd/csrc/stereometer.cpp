// Host side of the stereometer path: reference src/visuals/stereometer/processor.rs:64-212 (history
// length bookkeeping, config updates, EMA alpha) and src/dsp.rs:399-420 (RBJ Butterworth design).
#include "stereometer.hpp"

#include <cstdlib>

namespace omx {

constexpr float kBandSplitsHz[2] = {200.0f, 2000.0f};  // reference src/util/audio.rs:26

void stereometer_config_default(omx_stereometer_config* c) {  // :11-21
    std::memset(c, 0, sizeof(*c));
    c->sample_rate = kDefaultSampleRate;
    c->segment_duration = 0.02f;
    c->target_sample_count = 2000;
    c->correlation_window = 0.05f;
    c->analyze_bands = 0;
    c->emit_band_points = 0;
}

BiquadCoef make_biquad(bool highpass, float sample_rate, float frequency) {  // dsp.rs:402-420
    float ratio = frequency / sample_rate;
    ratio = ratio < 1.0e-6f ? 1.0e-6f : (ratio > 0.49f ? 0.49f : ratio);
    const float ang = kTau * ratio;
    const float sn = std::sin(ang), cs = std::cos(ang);
    const float alpha = sn * kFrac1Sqrt2;
    const float gain = highpass ? 1.0f + cs : 1.0f - cs;
    const float sign = highpass ? -1.0f : 1.0f;
    const float inv_a0 = 1.0f / (1.0f + alpha);
    BiquadCoef c;
    c.b[0] = gain * 0.5f * inv_a0;
    c.b[1] = gain * inv_a0 * sign;
    c.b[2] = gain * 0.5f * inv_a0;
    c.a[0] = -2.0f * cs * inv_a0;
    c.a[1] = (1.0f - alpha) * inv_a0;
    return c;
}

static double ema_alpha(float sample_rate, float window) {  // :210-212
    return 1.0 - std::exp(-1.0 / std::fmax((double)sample_rate * (double)window, 1.0));
}

StereometerBank::StereometerBank(const omx_stereometer_config& cfg, uint32_t n_streams) : n_streams_(n_streams) {
    state_.reserve((size_t)n_streams_ * 4);
    init(cfg);
}

void StereometerBank::init(const omx_stereometer_config& in) {  // ::new (:75-86)
    cfg_ = in;
    cfg_.analyze_bands = (cfg_.analyze_bands || cfg_.emit_band_points) ? 1 : 0;
    cfg_.emit_band_points = cfg_.emit_band_points ? 1 : 0;
    cfg_._pad = 0;
    history_channels_ = 0;
    for (int b = 0; b < 4; ++b) hist_len_[b] = hist_pos_[b] = 0;
    ragged_zero_mask_ |= 0xFu;
    alpha_ = ema_alpha(cfg_.sample_rate, cfg_.correlation_window);
    pending_full_reset_ = true;
}

uint32_t StereometerBank::segment_frames() const {  // :142-144
    return (uint32_t)std::min<size_t>(f2usize((double)std::fmax(std::round(cfg_.sample_rate * cfg_.segment_duration), 1.0f)),
                                      0x7FFFFFFFu);
}

// Zero-input transition of the band cascades over `frames` frames (and its powers 2, 4 ... 32), in f64 from the f32 coefficients: column m of T_band is the
// state after `frames` steps of Biquad::process (dsp.rs:422-432) with x = 0 started from the unit state e_m.  State order:
// [stage A element 0 (z0, z1), A element 1, stage B element 0, B element 1]; the low band has stage A only.
static std::vector<double> band_transitions(const BiquadCoef& lp_lo, const BiquadCoef& hp_lo, const BiquadCoef& lp_hi, const BiquadCoef& hp_hi,
                                            uint64_t frames) {
    std::vector<double> T(3 * 6 * 64, 0.0);  // [band][power 1, 2, 4 ... 32][8][8]
    const BiquadCoef* stage_a[3] = {&lp_lo, &hp_lo, &hp_lo};
    const BiquadCoef* stage_b[3] = {nullptr, &lp_hi, &hp_hi};
    for (int band = 0; band < 3; ++band) {
        const int n = stage_b[band] ? 8 : 4;
        for (int m = 0; m < n; ++m) {
            double z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            z[m] = 1.0;
            for (uint64_t f = 0; f < frames; ++f) {
                double x = 0.0;
                for (int e = 0; e < n / 2; ++e) {
                    const BiquadCoef& c = e < 2 ? *stage_a[band] : *stage_b[band];
                    const double out = (double)c.b[0] * x + z[2 * e];
                    z[2 * e] = (double)c.b[1] * x - (double)c.a[0] * out + z[2 * e + 1];
                    z[2 * e + 1] = (double)c.b[2] * x - (double)c.a[1] * out;
                    x = out;
                }
            }
            for (int k = 0; k < n; ++k) T[(size_t)band * 384 + (size_t)k * 8 + m] = z[k];
        }
        // T^2, T^4 ... T^32 by repeated squaring (the scan's doubling steps)
        for (int p = 1; p < 6; ++p) {
            const double* prev = T.data() + (size_t)band * 384 + (size_t)(p - 1) * 64;
            double* next = T.data() + (size_t)band * 384 + (size_t)p * 64;
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 8; ++j) {
                    long double acc = 0.0L;
                    for (int k = 0; k < 8; ++k) acc += (long double)prev[i * 8 + k] * (long double)prev[k * 8 + j];
                    next[i * 8 + j] = (double)acc;
                }
        }
    }
    return T;
}

void StereometerBank::reset_audio() {  // :92-97
    for (int b = 0; b < 4; ++b) hist_len_[b] = 0;
    pending_full_reset_ = true;  // band_splitter.clear() + correlators = default
    ragged_ = false;             // every deque is empty: the (lock-step) host positions serve again
    ragged_zero_mask_ = 0;
}

void StereometerBank::update_config(const omx_stereometer_config& in) {  // :183-207
    omx_stereometer_config cfg = in;
    cfg.analyze_bands = (cfg.analyze_bands || cfg.emit_band_points) ? 1 : 0;
    cfg.emit_band_points = cfg.emit_band_points ? 1 : 0;
    cfg._pad = 0;
    const bool rate_changed = cfg_.sample_rate != cfg.sample_rate;
    const bool window_changed = std::fabs(cfg_.correlation_window - cfg.correlation_window) > std::numeric_limits<float>::epsilon();
    const bool bands_changed = cfg_.analyze_bands != cfg.analyze_bands;
    cfg_ = cfg;
    if (rate_changed) {
        init(cfg_);
    } else {
        if (window_changed) alpha_ = ema_alpha(cfg.sample_rate, cfg.correlation_window);
        if (bands_changed) pending_band_reset_ = true;  // new splitter + correlators[1..] = default
    }
    if (!cfg.emit_band_points) {
        for (int b = 1; b < 4; ++b) hist_len_[b] = 0;
        ragged_zero_mask_ |= 0xEu;
    }
}

int StereometerBank::process(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels_in,
                             float sample_rate_in, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                             omx_stereometer_bank_update* out) {  // :99-182
    if (ragged_) {
        set_last_error("stereometer bank: per-stream positions are in use (process_ragged); reset_audio() returns the bank to lock-step calls");
        return OMX_ERR_INVALID;
    }
    return process_impl(pcm, pcm_on_device, block_frames, n_blocks, channels_in, sample_rate_in, positions, stream, out, nullptr);
}

int StereometerBank::process_ragged(const float* d_pcm, uint64_t block_frames, uint64_t max_blocks, const uint32_t* n_blocks,
                                    const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                    const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_stereometer_ragged_update* out) {
    bool any = false;
    for (uint32_t s = 0; s < n_streams_; ++s) {
        if (n_blocks[s] > max_blocks) {
            set_last_error("stereometer process_ragged: n_blocks[s] > max_blocks");
            return OMX_ERR_INVALID;
        }
        any = any || n_blocks[s] != 0 || (reset_mask && reset_mask[s]);
    }
    last_stream_ = stream;
    if (!any || block_frames == 0) return OMX_NONE;
    const RaggedCall rc{n_blocks, reset_mask, out};
    return process_impl(d_pcm, true, block_frames, std::max<uint64_t>(max_blocks, 1), channels, sample_rate, positions, stream, nullptr, &rc);
}

// One block per capture, each of its own length: what VisualManager::ingest_samples hands StereometerProcessor::process_block
// (registry.rs:396-418) when every capture has its own batcher (meter.rs:40-69: 1 ... 4 quanta per chunk, ONE call per chunk).
int StereometerBank::process_chunks(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask,
                                    uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                    omx_stereometer_ragged_update* out) {
    if (frames_capacity == 0 || frames_capacity > 0xFFFFFFFFull) {
        set_last_error("stereometer process_chunks: frames_capacity must be in 1 ... 2^32 - 1");
        return OMX_ERR_INVALID;
    }
    h_blocks_.resize(n_streams_);
    bool any = false;
    uint64_t longest = 1;
    for (uint32_t s = 0; s < n_streams_; ++s) {
        if (frames[s] > frames_capacity) {
            set_last_error("stereometer process_chunks: frames[s] > frames_capacity");
            return OMX_ERR_INVALID;
        }
        h_blocks_[s] = frames[s] != 0 ? 1u : 0u;  // block.is_empty(): nothing happens
        longest = std::max<uint64_t>(longest, frames[s]);
        any = any || frames[s] != 0 || (reset_mask && reset_mask[s]);
    }
    last_stream_ = stream;
    if (!any) return OMX_NONE;
    RaggedCall rc{h_blocks_.data(), reset_mask, out};
    rc.frames_v = frames;
    rc.row_frames = frames_capacity;
    return process_impl(d_pcm, true, longest, 1, channels, sample_rate, positions, stream, nullptr, &rc);
}

int StereometerBank::process_impl(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels_in,
                                  float sample_rate_in, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                  omx_stereometer_bank_update* out, const RaggedCall* ragged) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    last_stream_ = stream;
    if (block_frames == 0 || n_blocks == 0) return OMX_NONE;
    if (block_frames > 0xFFFFFFFFull || n_blocks > 0xFFFFFFFFull) unsupported("stereometer block shape beyond 2^32");
    const float sample_rate = sanitize_sample_rate(sample_rate_in);
    if (cfg_.sample_rate != sample_rate) {
        omx_stereometer_config c = cfg_;
        c.sample_rate = sample_rate;
        update_config(c);
    }
    if (history_channels_ != channels) {
        hist_len_[0] = 0;
        ragged_zero_mask_ |= 1u;
        history_channels_ = channels;
    }
    const uint32_t frames = segment_frames();
    if (!history_.ptr) {
        hist_frames_ = frames;
        history_.reserve((size_t)n_streams_ * 4 * frames * 2);
        OMX_HIP(hipMemsetAsync(history_.ptr, 0, history_.count * sizeof(float), stream));
        for (int b = 0; b < 4; ++b) hist_len_[b] = hist_pos_[b] = 0;
    } else if (frames != hist_frames_ && ragged_) {
        unsupported("stereometer: the segment length changed while the bank holds per-stream history positions (reset_audio first)");
    } else if (frames != hist_frames_) {
        // the deques survive a segment-length change (:183-207) and are trimmed to the new length (:146-150): carry the newest
        // min(len, frames) pairs of every band into a ring of the new length
        uint64_t keep[4];
        for (int b = 0; b < 4; ++b) keep[b] = std::min<uint64_t>(hist_len_[b], frames);
        history_next_.reserve((size_t)n_streams_ * 4 * frames * 2);
        OMX_HIP(hipMemsetAsync(history_next_.ptr, 0, (size_t)n_streams_ * 4 * frames * 2 * sizeof(float), stream));
        launch_stereometer_rehome(history_.ptr, history_next_.ptr, n_streams_, hist_frames_, frames, hist_pos_, keep, stream);
        OMX_HIP(hipGetLastError());
        std::swap(history_.ptr, history_next_.ptr);
        std::swap(history_.count, history_next_.count);
        hist_frames_ = frames;
        for (int b = 0; b < 4; ++b) hist_len_[b] = keep[b];
    }
    if (pending_full_reset_) {
        OMX_HIP(hipMemsetAsync(state_.ptr, 0, state_.count * sizeof(StereoLaneState), stream));
        pending_full_reset_ = pending_band_reset_ = false;
    } else if (pending_band_reset_) {
        for (uint32_t s = 0; s < n_streams_; ++s)
            OMX_HIP(hipMemsetAsync(state_.ptr + (size_t)s * 4 + 1, 0, 3 * sizeof(StereoLaneState), stream));
        pending_band_reset_ = false;
    }
    const bool chunk_call = ragged && ragged->frames_v;
    const uint64_t total = chunk_call ? ragged->row_frames : block_frames * n_blocks;  // frames per row of `pcm`
    const float* d_pcm = pcm;
    if (!pcm_on_device) {
        const size_t n = (size_t)n_streams_ * total * channels;
        d_pcm = staging_.stage(pcm, n, stream);
    }
    correlations_.reserve((size_t)(n_streams_ * n_blocks * 4), host_outputs_ && n_streams_ * n_blocks <= 4096);

    StereometerArgs sa{};
    sa.pcm = d_pcm;
    sa.frames_total = total;
    sa.block_frames = (uint32_t)block_frames;
    sa.n_blocks = (uint32_t)n_blocks;
    sa.n_streams = n_streams_;
    sa.fmt = make_format(channels, positions);
    const BiquadCoef lp_lo = make_biquad(false, cfg_.sample_rate, kBandSplitsHz[0]);
    const BiquadCoef hp_lo = make_biquad(true, cfg_.sample_rate, kBandSplitsHz[0]);
    const BiquadCoef lp_hi = make_biquad(false, cfg_.sample_rate, kBandSplitsHz[1]);
    const BiquadCoef hp_hi = make_biquad(true, cfg_.sample_rate, kBandSplitsHz[1]);
    sa.stage_a[1] = lp_lo; sa.use_a[1] = 1; sa.use_b[1] = 0; sa.stage_b[1] = lp_lo;
    sa.stage_a[2] = hp_lo; sa.use_a[2] = 1; sa.stage_b[2] = lp_hi; sa.use_b[2] = 1;
    sa.stage_a[3] = hp_lo; sa.use_a[3] = 1; sa.stage_b[3] = hp_hi; sa.use_b[3] = 1;
    sa.stage_a[0] = lp_lo; sa.stage_b[0] = lp_lo; sa.use_a[0] = sa.use_b[0] = 0;
    sa.analyze_bands = cfg_.analyze_bands;
    sa.emit_band_points = cfg_.emit_band_points;
    sa.alpha = alpha_;
    sa.state = state_.ptr;
    sa.history = history_.ptr;
    sa.hist_frames = frames;
    for (int b = 0; b < 4; ++b) sa.hist_pos[b] = hist_pos_[b];
    sa.correlations = correlations_.ptr;
    // chunk-parallel evaluation for bank-sized calls (stereometer_chunked.hip); everything else — single-stream handles, short
    // calls, other channel counts — stays on the sequential kernels, whose results are bit-identical to the reference's order
    const bool shape_ok = channels == 2 && block_frames % 16 == 0 && block_frames >= 32 && n_blocks >= 2 && !chunk_call;
    // by shape: whatever the bank size — the sequential kernels cost ~27 us per block however few streams there are, the chunk form ~0.08 ms
    // of launches plus its work (tools/bench_meter_forms.py: 1 stream x 64 blocks 1.75 -> 0.10 ms, 16 x 8 0.23 -> 0.07 ms; until round 4
    // the rule also asked for >= 512 (stream, block) items)
    const bool chunked = shape_ok && chunked_mode_ != 0 && (chunked_mode_ == 1 || n_blocks >= 4);
    last_form_ = chunked ? 2 : 1;
    auto run_chunked = [&]() {  // (after the plan kernel in a ragged call: the per-stream history positions are its output)
        // chunks shorter than blocks while the call has too few (stream, block) items to give every SIMD two wavefronts
        static const int forced_cpb = [] {
            const char* e = tuning_env("OMX_STEREO_CPB");  // tuning hook: 1 / 2 / 4
            return e ? std::atoi(e) : 0;
        }();
        uint32_t cpb = 1;
        while (cpb < 4 && (uint64_t)n_streams_ * n_blocks * cpb < 32768 && block_frames % (cpb * 2 * 16) == 0 && block_frames / (cpb * 2) >= 64) cpb *= 2;
        if (forced_cpb == 1 || forced_cpb == 2 || forced_cpb == 4)
            if (block_frames % ((uint64_t)forced_cpb * 16) == 0 && block_frames / forced_cpb >= 32) cpb = (uint32_t)forced_cpb;
        const uint64_t chunk_frames = block_frames / cpb;
        if (transition_rate_ != cfg_.sample_rate || transition_frames_ != chunk_frames) {
            transition_.upload(band_transitions(lp_lo, hp_lo, lp_hi, hp_hi, chunk_frames), stream);
            transition_rate_ = cfg_.sample_rate;
            transition_frames_ = chunk_frames;
        }
        const uint64_t items = (uint64_t)n_streams_ * n_blocks * cpb;
        chunk_state_.reserve((size_t)items * 3 * 16);
        chunk_moments_.reserve((size_t)items * 12);
        bad_.reserve(1);
        state_backup_.reserve(state_.count);
        OMX_HIP(hipMemsetAsync(bad_.ptr, 0, sizeof(uint32_t), stream));
        OMX_HIP(hipMemcpyAsync(state_backup_.ptr, state_.ptr, state_.count * sizeof(StereoLaneState), hipMemcpyDeviceToDevice, stream));
        StereoChunkArgs ca{};
        ca.pcm = d_pcm;
        ca.frames_total = total;
        ca.block_frames = (uint32_t)chunk_frames;
        ca.n_blocks = (uint32_t)(n_blocks * cpb);
        ca.cpb = cpb;
        ca.n_streams = n_streams_;
        ca.m00 = sa.fmt.m[0][0];
        ca.m10 = sa.fmt.m[1][0];
        ca.m01 = sa.fmt.m[0][1];
        ca.m11 = sa.fmt.m[1][1];
        ca.lp_lo = lp_lo;
        ca.hp_lo = hp_lo;
        ca.lp_hi = lp_hi;
        ca.hp_hi = hp_hi;
        ca.analyze_bands = cfg_.analyze_bands;
        ca.emit_band_points = cfg_.emit_band_points;
        ca.alpha = alpha_;
        ca.state = state_.ptr;
        ca.history = history_.ptr;
        ca.hist_frames = frames;
        for (int b = 0; b < 4; ++b) ca.hist_pos[b] = hist_pos_[b];
        ca.correlations = correlations_.ptr;
        ca.chunk_state = chunk_state_.ptr;
        ca.chunk_moments = chunk_moments_.ptr;
        ca.bad = bad_.ptr;
        ca.blocks_v = sa.blocks_v;
        ca.reset_v = sa.reset_v;
        ca.start_v = sa.start_v;
        launch_stereometer_chunked(ca, transition_.ptr, std::pow(1.0 - alpha_, (double)chunk_frames), stream);
        OMX_HIP(hipGetLastError());
        // non-finite input / output breaks the linearity the chunks rely on (Biquad::process resets, dsp.rs:428-431): the
        // sequential kernel then redoes the whole call from the saved state (it exits at once when the flag is clear)
        sa.run_if = bad_.ptr;
        sa.state_in = state_backup_.ptr;
    };
    if (chunked && !ragged) run_chunked();
    if (ragged) {
        if (!ragged_) {  // every stream starts from the bank's common positions and lengths
            std::vector<uint64_t> pos((size_t)n_streams_ * 4), len((size_t)n_streams_ * 4);
            for (uint32_t s = 0; s < n_streams_; ++s)
                for (int b = 0; b < 4; ++b) {
                    pos[(size_t)s * 4 + b] = hist_pos_[b];
                    len[(size_t)s * 4 + b] = hist_len_[b];
                }
            r_pos_.upload(pos, stream);
            r_len_.upload(len, stream);
            ragged_ = true;
            ragged_zero_mask_ = 0;  // (the host lengths just uploaded already carry it)
        }
        r_start_.reserve((size_t)n_streams_ * 4);
        r_valid_.reserve((size_t)n_streams_ * 4);
        r_staging_.upload(ragged->n_blocks, ragged->reset_mask, n_streams_, r_blocks_, r_mask_, stream, chunk_call ? ragged->frames_v : nullptr,
                          &r_frames_);
        produced_.reserve((size_t)(n_streams_ * n_blocks));
        StereoPlanArgs pa{};
        pa.n_streams = n_streams_;
        pa.max_blocks = (uint32_t)n_blocks;
        pa.block_frames = (uint32_t)block_frames;
        pa.hist_frames = frames;
        pa.analyze_bands = cfg_.analyze_bands;
        pa.emit_band_points = cfg_.emit_band_points;
        pa.blocks = r_blocks_.ptr;
        pa.frames = chunk_call ? r_frames_.ptr : nullptr;
        pa.reset = r_mask_.ptr;
        pa.pos = r_pos_.ptr;
        pa.len = r_len_.ptr;
        pa.start = r_start_.ptr;
        pa.produced = produced_.ptr;
        pa.band_valid = r_valid_.ptr;
        pa.zero_len_mask = ragged_zero_mask_;
        ragged_zero_mask_ = 0;
        launch_stereometer_ragged_plan(pa, stream);
        OMX_HIP(hipGetLastError());
        sa.blocks_v = r_blocks_.ptr;
        sa.reset_v = r_mask_.ptr;
        sa.start_v = r_start_.ptr;
        sa.frames_v = chunk_call ? r_frames_.ptr : nullptr;
        OMX_HIP(hipMemsetAsync(correlations_.ptr, 0, (size_t)(n_streams_ * n_blocks * 4) * sizeof(float), stream));  // slots past a stream's own blocks
        if (chunked) run_chunked();
        launch_stereometer(sa, stream);
        OMX_HIP(hipGetLastError());
        const uint32_t target = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cfg_.target_sample_count, 1), frames);  // :152
        last_target_ = target;
        last_blocks_ = n_blocks;
        produced_host_.clear();
        points_.reserve((size_t)n_streams_ * 4 * target * 2, false);
        launch_stereometer_points_ragged(history_.ptr, n_streams_, frames, r_pos_.ptr, r_valid_.ptr, target, points_.ptr, stream);
        OMX_HIP(hipGetLastError());
        if (ragged->out) {
            ragged->out->n_streams = n_streams_;
            ragged->out->max_blocks = n_blocks;
            ragged->out->target = target;
            ragged->out->d_n_blocks = r_blocks_.ptr;
            ragged->out->d_correlations = correlations_.ptr;
            ragged->out->d_produced = produced_.ptr;
            ragged->out->d_points = points_.ptr;
            ragged->out->d_band_valid = r_valid_.ptr;
        }
        return OMX_PRODUCED;
    }
    launch_stereometer(sa, stream);
    OMX_HIP(hipGetLastError());

    // deque lengths per block (:116, :129, :146-150): produced iff the full-band history is full
    produced_host_.assign((size_t)n_blocks, 0);
    const uint64_t len_before = hist_len_[0];
    for (uint64_t blk = 0; blk < n_blocks; ++blk) {
        hist_len_[0] = std::min<uint64_t>(hist_len_[0] + block_frames, frames);
        if (cfg_.analyze_bands && cfg_.emit_band_points)
            for (int b = 1; b < 4; ++b) hist_len_[b] = std::min<uint64_t>(hist_len_[b] + block_frames, frames);
        produced_host_[blk] = hist_len_[0] >= frames ? 1u : 0u;
    }
    hist_pos_[0] += total;
    if (cfg_.analyze_bands && cfg_.emit_band_points)
        for (int b = 1; b < 4; ++b) hist_pos_[b] += total;
    produced_.reserve((size_t)(n_streams_ * n_blocks));
    // the per-block "history full" flags follow from three scalars: written by a kernel, so the call stays asynchronous (a host
    // staging copy had to be synchronised before returning, which serialised the caller's other streams behind this bank)
    launch_stereometer_produced(produced_.ptr, n_streams_, (uint32_t)n_blocks, len_before, (uint32_t)block_frames, frames, stream);
    OMX_HIP(hipGetLastError());

    const uint32_t target = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cfg_.target_sample_count, 1), frames);  // :152
    last_target_ = target;
    last_blocks_ = n_blocks;
    const bool produced_last = produced_host_.back() != 0;
    for (int b = 0; b < 4; ++b) {
        const bool in_play = b == 0 || cfg_.emit_band_points;
        band_valid_[b] = (produced_last && in_play && hist_len_[b] >= frames) ? 1u : 0u;
    }
    if (produced_last) {
        points_.reserve((size_t)n_streams_ * 4 * target * 2, host_outputs_ && n_streams_ <= 4);
        launch_stereometer_points(history_.ptr, n_streams_, frames, hist_pos_, band_valid_, target, points_.ptr, stream);
        OMX_HIP(hipGetLastError());
    }
    if (out) {
        out->n_streams = n_streams_;
        out->n_blocks = n_blocks;
        out->target = target;
        out->d_correlations = correlations_.ptr;
        out->d_points = produced_last ? points_.ptr : nullptr;
        out->d_produced = produced_.ptr;
    }
    return produced_last ? OMX_PRODUCED : OMX_NONE;
}

int StereometerBank::fetch(uint64_t stream_index, uint64_t block, float correlations[4], uint32_t* produced, hipStream_t stream) {
    if (stream_index >= n_streams_ || block >= last_blocks_) {
        set_last_error("stereometer fetch: index out of range");
        return OMX_ERR_INVALID;
    }
    copy_out(correlations, correlations_.ptr + (stream_index * last_blocks_ + block) * 4, 4 * sizeof(float), correlations_.pinned, stream);
    if (produced) {
        if (ragged_) copy_out(produced, produced_.ptr + stream_index * last_blocks_ + block, sizeof(uint32_t), false, stream);
        else *produced = produced_host_[block];
    }
    return OMX_NONE;
}

int StereometerBank::fetch_points(uint64_t stream_index, uint32_t band, float* dst, uint64_t* n_pairs, hipStream_t stream) {
    if (stream_index >= n_streams_ || band >= 4) return OMX_ERR_INVALID;
    uint32_t valid = band_valid_[band];
    if (ragged_) copy_out(&valid, r_valid_.ptr + stream_index * 4 + band, sizeof(uint32_t), false, stream);  // per stream
    if (!valid) {
        *n_pairs = 0;
        return OMX_NONE;
    }
    copy_out(dst, points_.ptr + (stream_index * 4 + band) * (uint64_t)last_target_ * 2, (size_t)last_target_ * 2 * sizeof(float),
             points_.pinned, stream);
    *n_pairs = last_target_;
    return OMX_NONE;
}

}  // namespace omx
