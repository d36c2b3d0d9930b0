// Host side of the waveform path: reference src/visuals/waveform/processor.rs:31-52 (config normalisation), :147-211
// (lifecycle), :308-352 (process_block / update_config).  The fractional column phase is advanced on the host with the
// reference's exact f64 add / compare / subtract sequence so the number of emitted columns is known before launch.
#include "waveform.hpp"

namespace omx {

constexpr size_t kWfMaxColumns = 8192;        // :11
constexpr float kWfDefaultScroll = 300.0f;    // :13
constexpr float kWfMinRuntimeScroll = 1.0f;   // :15
constexpr float kBandSplits[2] = {200.0f, 2000.0f};

void waveform_config_default(omx_waveform_config* c) {  // :31-40
    c->sample_rate = kDefaultSampleRate;
    c->scroll_speed = kWfDefaultScroll;
    c->max_columns = kWfMaxColumns;
    c->analyze_bands = 1;
    c->track_history = 0;
}
static omx_waveform_config normalized(omx_waveform_config c) {  // :42-52
    c.sample_rate = sanitize_sample_rate(c.sample_rate);
    c.scroll_speed = (std::isfinite(c.scroll_speed) && c.scroll_speed > 0.0f) ? std::fmax(c.scroll_speed, kWfMinRuntimeScroll) : kWfDefaultScroll;
    c.max_columns = std::min<uint64_t>(std::max<uint64_t>(c.max_columns, 1), kWfMaxColumns);
    c.analyze_bands = c.analyze_bands ? 1 : 0;
    c.track_history = (c.track_history && c.analyze_bands) ? 1 : 0;
    return c;
}
static uint32_t window_len(size_t at_reference_rate, float sample_rate) {  // :78-82
    sample_rate = std::fmin(sample_rate, 1000000.0f);
    return (uint32_t)std::max<size_t>(f2usize((double)std::round((float)at_reference_rate * sample_rate / 44100.0f)), 1);
}

WaveformBank::WaveformBank(const omx_waveform_config& cfg, uint32_t n_streams) : n_streams_(n_streams) {
    cfg_ = normalized(cfg);
    state_.reserve((size_t)n_streams_ * 16);
}

void WaveformBank::rebuild() {  // :175-184
    column_phase_ = 0.0;
    ragged_zero_phase_ = true;
    clear_minmax_ = true;
    if (analysis_) reset_trackers();
    reset_pending_ = true;
}
void WaveformBank::reset_trackers() {  // :199-201 (band_analysis is None when analyze_bands is off)
    analysis_ = cfg_.analyze_bands != 0;
    clear_trackers_ = true;
    pushes_ = 0;
    ragged_zero_pushes_ = true;
}
void WaveformBank::prepare(hipStream_t) {  // :169-173
    if (cfg_.analyze_bands && !analysis_) reset_trackers();
}
void WaveformBank::update_config(const omx_waveform_config& in) {  // :336-352
    const omx_waveform_config n = normalized(in);
    const bool rebuild_all = cfg_.sample_rate != n.sample_rate;
    const bool reset_analysis = cfg_.analyze_bands != n.analyze_bands || cfg_.track_history != n.track_history;
    cfg_ = n;
    if (rebuild_all) rebuild();
    else if (reset_analysis && analysis_) reset_trackers();
}

int WaveformBank::process(const float* pcm, bool pcm_on_device, uint64_t frames, uint32_t channels_in, float sample_rate_in,
                          const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_waveform_bank_update* out) {  // :308-334
    if (ragged_) {
        set_last_error("waveform bank: per-stream positions are in use (process_ragged); reset_audio() returns the bank to lock-step calls");
        return OMX_ERR_INVALID;
    }
    return process_impl(pcm, pcm_on_device, frames, channels_in, sample_rate_in, positions, stream, out, nullptr);
}

int WaveformBank::process_ragged(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask,
                                 uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                 omx_waveform_ragged_update* out) {
    bool any = false;
    for (uint32_t s = 0; s < n_streams_; ++s) {
        if (frames[s] > frames_capacity) {
            set_last_error("waveform process_ragged: frames[s] > frames_capacity");
            return OMX_ERR_INVALID;
        }
        any = any || frames[s] != 0 || (reset_mask && reset_mask[s]);
    }
    last_stream_ = stream;
    if (!any || frames_capacity == 0) return OMX_NONE;
    const RaggedCall rc{frames, reset_mask, out};
    return process_impl(d_pcm, true, frames_capacity, channels, sample_rate, positions, stream, nullptr, &rc);
}

int WaveformBank::process_impl(const float* pcm, bool pcm_on_device, uint64_t frames, uint32_t channels_in, float sample_rate_in,
                               const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_waveform_bank_update* out,
                               const RaggedCall* ragged) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    last_stream_ = stream;
    if (frames == 0) return OMX_NONE;
    const float sample_rate = sanitize_sample_rate(sample_rate_in);
    if (channels != source_channels_ || cfg_.sample_rate != sample_rate) {
        source_channels_ = channels;
        cfg_.sample_rate = sample_rate;
        rebuild();
    }
    prepare(stream);
    const uint32_t color_len = window_len(2048, cfg_.sample_rate), slow_len = window_len(16384, cfg_.sample_rate);
    if (color_len != color_len_ || slow_len != slow_len_ || !color_ring_.ptr) {
        color_len_ = color_len;
        slow_len_ = slow_len;
        color_ring_.reserve((size_t)color_len * n_streams_ * 16);
        hist_ring_.reserve((size_t)slow_len * n_streams_ * 16);
        clear_trackers_ = true;
        pushes_ = 0;
        ragged_zero_pushes_ = true;
    }
    if (clear_minmax_ && clear_trackers_) {
        OMX_HIP(hipMemsetAsync(state_.ptr, 0, state_.count * sizeof(WaveLaneState), stream));
    } else if (clear_minmax_ || clear_trackers_) {
        // partial clears are rare (config toggles): do them on the host side of the struct
        std::vector<WaveLaneState> h(state_.count);
        OMX_HIP(hipMemcpyAsync(h.data(), state_.ptr, h.size() * sizeof(WaveLaneState), hipMemcpyDeviceToHost, stream));
        OMX_HIP(hipStreamSynchronize(stream));
        for (auto& st : h) {
            if (clear_trackers_) {
                std::memset(st.za, 0, sizeof(st.za));
                std::memset(st.zb, 0, sizeof(st.zb));
                std::memset(st.color, 0, sizeof(st.color));
                std::memset(st.hist, 0, sizeof(st.hist));
            }
            if (clear_minmax_) {
                st.cur_min = st.cur_max = st.cur_last = st.last_sample = 0.0f;
                st.cur_some = st.cur_has_last = st.last_valid = 0;
            }
        }
        OMX_HIP(hipMemcpyAsync(state_.ptr, h.data(), h.size() * sizeof(WaveLaneState), hipMemcpyHostToDevice, stream));
        OMX_HIP(hipStreamSynchronize(stream));
    }
    if (clear_trackers_) {
        OMX_HIP(hipMemsetAsync(color_ring_.ptr, 0, color_ring_.count * sizeof(float), stream));
        OMX_HIP(hipMemsetAsync(hist_ring_.ptr, 0, hist_ring_.count * sizeof(float), stream));
    }
    clear_minmax_ = clear_trackers_ = false;

    const double step = std::min(std::max((double)cfg_.scroll_speed / (double)cfg_.sample_rate, 0.0), 1.0);
    if (ragged) {
        // every stream advances its own phase on the device; a call emits at most floor(phase + frames x step) <= max_cols columns
        const uint64_t max_cols = (uint64_t)std::floor(1.0 + (double)frames * step) + 1;
        if (max_cols > cfg_.max_columns)  // cap_pending_columns (:293-298) would drop the oldest ones: not modelled per stream
            unsupported("waveform process_ragged: frames_capacity x scroll_speed / sample_rate exceeds max_columns");
        if (!ragged_) {  // every stream starts from the bank's common push count and column phase
            r_pushes_.upload(std::vector<uint64_t>(n_streams_, pushes_), stream);
            r_phase_.upload(std::vector<double>(n_streams_, column_phase_), stream);
            ragged_ = true;
            ragged_zero_phase_ = ragged_zero_pushes_ = false;
        }
        if (ragged_zero_phase_) OMX_HIP(hipMemsetAsync(r_phase_.ptr, 0, n_streams_ * sizeof(double), stream));
        if (ragged_zero_pushes_) OMX_HIP(hipMemsetAsync(r_pushes_.ptr, 0, n_streams_ * sizeof(uint64_t), stream));
        ragged_zero_phase_ = ragged_zero_pushes_ = false;
        r_frames_.reserve(n_streams_);
        r_mask_.reserve(n_streams_);
        r_cols_.reserve(n_streams_);
        r_progress_.reserve(n_streams_);
        r_staging_.upload(ragged->frames, ragged->reset_mask, n_streams_, r_frames_.ptr, r_mask_.ptr, stream);
        columns_.reserve((size_t)(n_streams_ * max_cols * 4), false);
        preview_.reserve((size_t)n_streams_ * 4, false);
        WaveformArgs wa{};
        wa.pcm = pcm;
        wa.frames = frames;  // the row stride of pcm
        wa.n_streams = n_streams_;
        wa.fmt = make_format(channels, positions);
        wa.analyze = analysis_ ? 1 : 0;
        wa.track_history = (analysis_ && cfg_.track_history) ? 1 : 0;
        wa.lp_lo = make_biquad(false, cfg_.sample_rate, kBandSplits[0]);
        wa.hp_lo = make_biquad(true, cfg_.sample_rate, kBandSplits[0]);
        wa.lp_hi = make_biquad(false, cfg_.sample_rate, kBandSplits[1]);
        wa.hp_hi = make_biquad(true, cfg_.sample_rate, kBandSplits[1]);
        wa.step = step;
        wa.color_len = color_len_;
        wa.slow_len = slow_len_;
        wa.color_ring = color_ring_.ptr;
        wa.hist_ring = hist_ring_.ptr;
        wa.state = state_.ptr;
        wa.columns = columns_.ptr;
        wa.preview = preview_.ptr;
        wa.frames_v = r_frames_.ptr;
        wa.reset_v = r_mask_.ptr;
        wa.pushes_v = r_pushes_.ptr;
        wa.phase_v = r_phase_.ptr;
        wa.cols_v = r_cols_.ptr;
        wa.progress_v = r_progress_.ptr;
        wa.max_cols = max_cols;
        launch_waveform(wa, stream);
        OMX_HIP(hipGetLastError());
        last_cols_ = max_cols;
        if (ragged->out) {
            ragged->out->n_streams = n_streams_;
            ragged->out->max_columns = max_cols;
            ragged->out->d_n_columns = r_cols_.ptr;
            ragged->out->d_columns = columns_.ptr;
            ragged->out->d_preview = preview_.ptr;
            ragged->out->d_preview_progress = r_progress_.ptr;
            ragged->out->d_reset = r_mask_.ptr;
        }
        reset_pending_ = false;
        return OMX_PRODUCED;
    }
    // fractional column phase (:253-254, :287-291), exactly as the reference accumulates it
    double phase = column_phase_;
    uint64_t n_emit = 0;
    for (uint64_t f = 0; f < frames; ++f) {
        phase += step;
        if (phase >= 1.0) {
            ++n_emit;
            phase -= 1.0;
        }
    }
    const uint64_t kept = std::min<uint64_t>(n_emit, cfg_.max_columns);  // cap_pending_columns (:293-298)
    columns_.reserve((size_t)(n_streams_ * std::max<uint64_t>(kept, 1) * 4), host_outputs_ && n_streams_ * kept <= 16384);
    preview_.reserve((size_t)n_streams_ * 4, host_outputs_ && n_streams_ <= 64);

    const float* d_pcm = pcm;
    if (!pcm_on_device) {
        const size_t n = (size_t)n_streams_ * frames * channels;
        d_pcm = staging_.stage(pcm, n, stream);
    }
    const float progress = (float)std::min(std::max(phase, 0.0), 1.0);  // preview (:300-306)
    WaveformArgs wa{};
    wa.pcm = d_pcm;
    wa.frames = frames;
    wa.n_streams = n_streams_;
    wa.fmt = make_format(channels, positions);
    wa.analyze = analysis_ ? 1 : 0;
    wa.track_history = (analysis_ && cfg_.track_history) ? 1 : 0;
    wa.lp_lo = make_biquad(false, cfg_.sample_rate, kBandSplits[0]);
    wa.hp_lo = make_biquad(true, cfg_.sample_rate, kBandSplits[0]);
    wa.lp_hi = make_biquad(false, cfg_.sample_rate, kBandSplits[1]);
    wa.hp_hi = make_biquad(true, cfg_.sample_rate, kBandSplits[1]);
    wa.step = step;
    wa.column_phase = column_phase_;
    wa.pushes = pushes_;
    wa.color_len = color_len_;
    wa.slow_len = slow_len_;
    wa.color_ring = color_ring_.ptr;
    wa.hist_ring = hist_ring_.ptr;
    wa.state = state_.ptr;
    wa.n_emit = n_emit;
    wa.first_kept = n_emit - kept;
    wa.columns = columns_.ptr;
    wa.preview = preview_.ptr;
    wa.write_preview = progress > 0.0f ? 1 : 0;
    launch_waveform(wa, stream);
    OMX_HIP(hipGetLastError());
    column_phase_ = phase;
    if (analysis_) pushes_ += frames;
    last_cols_ = kept;
    if (out) {
        out->n_streams = n_streams_;
        out->n_columns = kept;
        out->d_columns = columns_.ptr;
        out->d_preview = preview_.ptr;
        out->reset = reset_pending_ ? 1 : 0;
        out->preview_some = wa.write_preview;
        out->preview_progress = progress;
        out->_pad = 0;
    }
    reset_pending_ = false;
    return OMX_PRODUCED;
}

int WaveformBank::fetch(uint64_t stream_index, omx_wave_column* columns, omx_wave_column* preview, hipStream_t stream) {
    if (stream_index >= n_streams_) return OMX_ERR_INVALID;
    if (columns && last_cols_)
        copy_out(columns, columns_.ptr + stream_index * last_cols_ * 4, last_cols_ * 4 * sizeof(omx_wave_column), columns_.pinned, stream);
    if (preview) copy_out(preview, preview_.ptr + stream_index * 4, 4 * sizeof(omx_wave_column), preview_.pinned, stream);
    return OMX_NONE;
}

}  // namespace omx
