// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of reference src/visuals/spectrum/processor.rs:16-425 (SpectrumProcessor:
// 1-2 projected traces, real FFT, none/exponential/peak-hold averaging, raw + A-weighted dB).
#pragma once
#include <complex>
#include <deque>
#include <vector>

#include "fft.hpp"
#include "primitives.hpp"

namespace omxo {

constexpr float DEFAULT_SPECTRUM_DB_FLOOR = -100.0f;       // :22
constexpr size_t DEFAULT_SPECTRUM_HOP_DIVISOR = 16;        // :24
constexpr size_t DEFAULT_SPECTRUM_FFT_SIZE = 16384;        // :25

struct SpectrumConfig {  // :39-51
    float sample_rate = DEFAULT_SAMPLE_RATE;
    size_t fft_size = DEFAULT_SPECTRUM_FFT_SIZE;
    size_t hop_size = DEFAULT_SPECTRUM_FFT_SIZE / DEFAULT_SPECTRUM_HOP_DIVISOR;
    uint32_t window = OMX_WINDOW_HANN;
    uint32_t averaging_mode = OMX_AVERAGING_NONE;
    float averaging_param = 0.0f;
    uint32_t source = OMX_CHANNEL_MID;
    uint32_t secondary_source = OMX_CHANNEL_NONE;
    float floor_db = DEFAULT_SPECTRUM_DB_FLOOR;

    void normalize() {  // :53-62
        sample_rate = sanitize_sample_rate(sample_rate);
        fft_size = std::max<size_t>(fft_size, 1);
        if (hop_size == 0) hop_size = std::max<size_t>(fft_size / DEFAULT_SPECTRUM_HOP_DIVISOR, 1);
        floor_db = sanitize_negative_db(floor_db, DEFAULT_SPECTRUM_DB_FLOOR);
    }
};

// :410-425
inline float a_weight(float freq_hz) {
    const double C1 = 20.598997 * 20.598997;
    const double C2 = 107.65265 * 107.65265;
    const double C3 = 737.86223 * 737.86223;
    const double C4 = 12194.217 * 12194.217;
    if (freq_hz <= 0.0f) return -std::numeric_limits<float>::infinity();
    const double f = (double)freq_hz;
    const double f2 = f * f;
    const double numerator = C4 * f2 * f2;
    const double denom = (f2 + C1) * std::sqrt((f2 + C2) * (f2 + C3)) * (f2 + C4);
    const double ra = numerator / denom;
    return (float)(20.0 * std::log10(ra) + 2.0);
}

// :332-336
inline float smoothing_state_floor(const std::vector<float>& weighting_db, float floor) {
    float headroom = 0.0f;
    for (float w : weighting_db) headroom = rmax(headroom, w);
    return rmax(db_to_power(floor - headroom), std::numeric_limits<float>::min());
}

struct SpectrumLevelBuffers {  // :325-403
    std::vector<float> smoothed_power, scratch_power;
    float state_floor = 0.0f;

    void reset(size_t bins, float floor, bool smoothing) {  // :339-347
        state_floor = floor;
        if (smoothing) smoothed_power.assign(bins, 0.0f);
        else smoothed_power.clear();
        scratch_power.assign(bins, 0.0f);
    }

    // outputs[0] = weighted, outputs[1] = raw (:391)
    void update_outputs(uint32_t mode, float param, std::vector<float> outputs[2], const std::vector<float>& weighting_db,
                        float dt_seconds, float floor) {  // :349-402
        const size_t bins = scratch_power.size();
        for (int o = 0; o < 2; ++o)
            if (outputs[o].size() != bins) outputs[o].resize(bins, floor);
        const std::vector<float>* powers = &scratch_power;
        if (mode == OMX_AVERAGING_EXPONENTIAL) {
            const float alpha = rclamp(param, 0.0f, 0.9999f);
            const size_t n = std::min(smoothed_power.size(), scratch_power.size());
            for (size_t i = 0; i < n; ++i) {
                float& avg = smoothed_power[i];
                const float power = scratch_power[i];
                avg = (avg <= 0.0f) ? power : avg * alpha + power * (1.0f - alpha);
                if (avg < state_floor) avg = 0.0f;
            }
            powers = &smoothed_power;
        } else if (mode == OMX_AVERAGING_PEAK_HOLD) {
            const float decay = db_to_power(-rmax(param, 0.0f) * dt_seconds);
            const size_t n = std::min(smoothed_power.size(), scratch_power.size());
            for (size_t i = 0; i < n; ++i) {
                float& hold = smoothed_power[i];
                hold = rmax(hold * decay, scratch_power[i]);
                if (hold < state_floor) hold = 0.0f;
            }
            powers = &smoothed_power;
        }
        std::vector<float>& weighted_out = outputs[0];
        std::vector<float>& raw_out = outputs[1];
        for (size_t i = 0; i < bins; ++i) {
            const float p = (*powers)[i];
            if (p < state_floor) {
                raw_out[i] = floor;
                weighted_out[i] = floor;
                continue;
            }
            const float db = std::log(p) * LN_TO_DB;
            raw_out[i] = rmax(db, floor);
            weighted_out[i] = rmax(db + weighting_db[i], floor);
        }
    }
};

struct SpectrumSnapshot {  // :33-37
    std::vector<float> frequency_bins;
    std::vector<float> traces[2][2];
};

class SpectrumProcessor {
public:
    explicit SpectrumProcessor(SpectrumConfig cfg) {  // :89-106
        cfg.normalize();
        config_ = cfg;
    }
    SpectrumConfig config() const { return config_; }

    void reset_audio() {  // :112-118
        if (prepared_) reset_level_buffers();
        pcm_[0].clear();
        pcm_[1].clear();
        pending_skip_ = 0;
    }
    void prepare() {  // :120-124
        if (!prepared_) rebuild_fft();
    }

    const SpectrumSnapshot* process_block(const AudioBlock& block) {  // :255-269
        if (block.is_empty()) return nullptr;
        if (block.sample_rate != config_.sample_rate) {
            config_.sample_rate = block.sample_rate;
            if (prepared_) reset_buffers();
        }
        prepare();
        push_sources(block);
        return process_ready_windows() ? &snapshot_ : nullptr;
    }

    void update_config(SpectrumConfig cfg) {  // :300-322
        const SpectrumConfig old = config_;
        cfg.normalize();
        config_ = cfg;
        if (!prepared_) return;
        const bool mode_changed = old.averaging_mode != cfg.averaging_mode;
        if (old.fft_size != cfg.fft_size || old.window != cfg.window) {
            rebuild_fft();
        } else if (old.sample_rate != cfg.sample_rate || old.hop_size != cfg.hop_size || old.source != cfg.source ||
                   old.secondary_source != cfg.secondary_source) {
            reset_buffers();
        } else if (mode_changed || std::fabs(old.floor_db - cfg.floor_db) > std::numeric_limits<float>::epsilon()) {
            reset_level_buffers();
        }
    }

    // test-only views
    bool prepared() const { return prepared_; }
    const std::deque<float>& pcm_buffer(int t) const { return pcm_[t]; }
    std::deque<float>& pcm_buffer_mut(int t) { return pcm_[t]; }
    SpectrumLevelBuffers& levels(int t) { return levels_[t]; }
    const SpectrumSnapshot& snapshot() const { return snapshot_; }

private:
    void rebuild_fft() {  // :126-136
        const size_t n = config_.fft_size;
        window_ = window_coefficients(config_.window, n);
        real_.assign(n, 0.0f);
        spectrum_.assign(n / 2 + 1, std::complex<float>(0, 0));
        prepared_ = true;
        bin_norm_ = compute_fft_bin_normalization(window_, n);
        reset_buffers();
    }
    void reset_buffers() {  // :138-150
        const size_t bins = config_.fft_size / 2 + 1;
        const float bin_hz = config_.sample_rate / (float)config_.fft_size;
        snapshot_.frequency_bins.resize(bins);
        a_weighting_db_.resize(bins);
        for (size_t b = 0; b < bins; ++b) {
            const float f = (float)b * bin_hz;
            snapshot_.frequency_bins[b] = f;
            a_weighting_db_[b] = a_weight(f);
        }
        reset_level_buffers();
        pcm_[0].clear();
        pcm_[1].clear();
        pending_skip_ = 0;
    }
    void reset_level_buffers() {  // :152-168
        const size_t bins = config_.fft_size / 2 + 1;
        const float floor = config_.floor_db;
        for (int t = 0; t < 2; ++t)
            for (int w = 0; w < 2; ++w) snapshot_.traces[t][w].assign(bins, floor);
        const float state_floor = smoothing_state_floor(a_weighting_db_, floor);
        bool active[2];
        active_traces(active);
        const bool smoothing = config_.averaging_mode != OMX_AVERAGING_NONE;
        for (int t = 0; t < 2; ++t) {
            if (active[t]) levels_[t].reset(bins, state_floor, smoothing);
            else levels_[t] = SpectrumLevelBuffers();
        }
    }
    void active_traces(bool out[2]) const {  // :174-177
        const uint32_t p = config_.source, s = config_.secondary_source;
        out[0] = p != OMX_CHANNEL_NONE;
        out[1] = s != OMX_CHANNEL_NONE && s != p;
    }

    bool process_ready_windows() {  // :179-213
        const size_t n = config_.fft_size, hop = config_.hop_size;
        const float floor = config_.floor_db;
        const float dt = (float)hop / config_.sample_rate;
        bool active[2];
        active_traces(active);
        bool produced = false;
        if (!active[0] && !active[1]) return false;
        for (;;) {
            bool all = true;
            for (int t = 0; t < 2; ++t) all = all && (!active[t] || pcm_[t].size() >= n);
            if (!all) break;
            for (int t = 0; t < 2; ++t)
                if (active[t]) process_trace_window(t, dt, floor);
            size_t drained = hop;
            for (int t = 0; t < 2; ++t) {
                if (!active[t]) continue;
                const size_t count = std::min(hop, pcm_[t].size());
                pcm_[t].erase(pcm_[t].begin(), pcm_[t].begin() + (std::ptrdiff_t)count);
                drained = std::min(drained, count);
            }
            pending_skip_ += hop - drained;
            produced = true;
        }
        return produced;
    }

    void process_trace_window(int trace, float dt, float floor) {  // :215-253
        copy_dc_removed_windowed(real_.data(), real_.size(), pcm_[trace], window_.data());
        rfft(real_.data(), real_.size(), spectrum_.data());
        SpectrumLevelBuffers& level = levels_[trace];
        for (size_t i = 0; i < spectrum_.size(); ++i) {
            const std::complex<float> c = spectrum_[i];
            level.scratch_power[i] = (c.real() * c.real() + c.imag() * c.imag()) * bin_norm_[i];
        }
        level.update_outputs(config_.averaging_mode, config_.averaging_param, snapshot_.traces[trace], a_weighting_db_,
                             dt, floor);
    }

    void push_sources(const AudioBlock& block) {  // :271-298
        const size_t frames = block.frame_count();
        const size_t skip = std::min(pending_skip_, frames);
        pending_skip_ -= skip;
        if (skip == frames) return;
        bool active[2];
        active_traces(active);
        for (size_t f = skip; f < frames; ++f) {
            float lr[2];
            block.stereo_frame(f, lr);
            if (active[0]) pcm_[0].push_back(project(config_.source, lr[0], lr[1]));
            if (active[1]) pcm_[1].push_back(project(config_.secondary_source, lr[0], lr[1]));
        }
    }

    SpectrumConfig config_;
    SpectrumSnapshot snapshot_;
    bool prepared_ = false;
    std::vector<float> window_, real_, bin_norm_, a_weighting_db_;
    std::vector<std::complex<float>> spectrum_;
    std::deque<float> pcm_[2];
    size_t pending_skip_ = 0;
    SpectrumLevelBuffers levels_[2];
};

}  // namespace omxo
