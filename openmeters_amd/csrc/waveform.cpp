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
    clear_minmax_ = true;
    if (analysis_) reset_trackers();
    reset_pending_ = true;
}
void WaveformBank::reset_trackers() {  // :199-201 (band_analysis is None when analyze_bands is off)
    analysis_ = cfg_.analyze_bands != 0;
    clear_trackers_ = true;
    pushes_ = 0;
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

    // fractional column phase (:253-254, :287-291), exactly as the reference accumulates it
    const double step = std::min(std::max((double)cfg_.scroll_speed / (double)cfg_.sample_rate, 0.0), 1.0);
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
