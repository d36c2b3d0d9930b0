// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of reference src/visuals/waveform/processor.rs:9-352 (min/max columns for L/R/M/S with
// fractional column phase, 12 dB/oct three-band split -> f32 WindowedMeans colour / RMS history).
#pragma once
#include <memory>
#include <optional>
#include <vector>

#include "primitives.hpp"

namespace omxo {

constexpr size_t WF_MAX_COLUMN_CAPACITY = 8192;          // :11
constexpr float WF_DEFAULT_SCROLL_SPEED = 300.0f;        // :13
constexpr float WF_MIN_RUNTIME_SCROLL_SPEED = 1.0f;      // :15
constexpr int WF_CHANNELS = 4;                           // L, R, Mid, Side (:16-18)
constexpr float WF_REFERENCE_SAMPLE_RATE = 44100.0f;     // :19
constexpr size_t WF_COLOR_WINDOW_AT_44K1 = 2048;         // :20
constexpr size_t WF_SLOW_WINDOW_AT_44K1 = 16384;         // :21
constexpr float WF_BAND_COLOR_GAINS[3] = {1.0f, 0.7f, 2.0f};  // :22
constexpr float WF_MAX_TRACKER_SAMPLE_RATE = 1000000.0f; // :24

struct WaveformConfig {  // :31-40
    float sample_rate = DEFAULT_SAMPLE_RATE;
    float scroll_speed = WF_DEFAULT_SCROLL_SPEED;
    size_t max_columns = WF_MAX_COLUMN_CAPACITY;
    bool analyze_bands = true;
    bool track_history = false;
    WaveformConfig normalized() const {  // :42-52
        WaveformConfig c = *this;
        c.sample_rate = sanitize_sample_rate(c.sample_rate);
        c.scroll_speed = (std::isfinite(c.scroll_speed) && c.scroll_speed > 0.0f) ? rmax(c.scroll_speed, WF_MIN_RUNTIME_SCROLL_SPEED)
                                                                                   : WF_DEFAULT_SCROLL_SPEED;
        c.max_columns = std::min(std::max<size_t>(c.max_columns, 1), WF_MAX_COLUMN_CAPACITY);
        c.track_history = c.track_history && c.analyze_bands;
        return c;
    }
};

inline size_t wf_window_len(size_t samples_at_reference_rate, float sample_rate) {  // :78-82
    sample_rate = rmin(sample_rate, WF_MAX_TRACKER_SAMPLE_RATE);
    return std::max<size_t>(f2usize((double)std::round((float)samples_at_reference_rate * sample_rate / WF_REFERENCE_SAMPLE_RATE)), 1);
}

using BandFilter = ThreeBand<1, 1, false>;  // ThreeBand<Biquad, false> (:86)

struct BandTracker {  // :92-121
    WindowedMeans<3, 1, float> color;
    std::unique_ptr<WindowedMeans<3, 2, float>> history;
    static size_t one(size_t a) { return a; }
    BandTracker(float sample_rate, bool track_history)
        : color(caps1(wf_window_len(WF_COLOR_WINDOW_AT_44K1, sample_rate))) {
        if (track_history) {
            const size_t caps[2] = {wf_window_len(WF_COLOR_WINDOW_AT_44K1, sample_rate), wf_window_len(WF_SLOW_WINDOW_AT_44K1, sample_rate)};
            history.reset(new WindowedMeans<3, 2, float>(caps));
        }
    }
    struct Caps1 { size_t v[1]; };
    static const size_t (&caps1(size_t c))[1] {
        static thread_local size_t buf[1];
        buf[0] = c;
        return buf;
    }
    void process(const float bands[3]) {
        std::array<float, 3> c;
        for (int b = 0; b < 3; ++b) {
            const float v = std::fabs(bands[b]) * WF_BAND_COLOR_GAINS[b];
            c[b] = std::isfinite(v) ? v : 0.0f;
        }
        color.push(c);
        if (history) {
            std::array<float, 3> p;
            for (int b = 0; b < 3; ++b) {
                const float pw = bands[b] * bands[b];
                p[b] = std::isfinite(pw) ? pw : 0.0f;
            }
            history->push(p);
        }
    }
};

struct WfCurrent {  // Option<(f32, f32, Option<f32>)>
    bool some = false;
    float min = 0, max = 0;
    bool has_last = false;
    float last = 0;
};

class WaveformProcessor {
public:
    explicit WaveformProcessor(WaveformConfig cfg) : config_(cfg.normalized()) {}  // :147-159
    WaveformConfig config() const { return config_; }
    void reset_audio() { rebuild(); }                                               // :165-167
    void prepare() {                                                                // :169-173
        if (config_.analyze_bands && !analysis_) make_analysis();
    }
    bool has_band_analysis() const { return analysis_ != nullptr; }
    double column_phase() const { return column_phase_; }

    struct Update {
        bool reset = false;
        std::vector<omx_wave_column> columns;  // [n][4]
        float preview_progress = 0.0f;
        bool preview_some = false;
        omx_wave_column preview[WF_CHANNELS];
    };

    bool process_block(const AudioBlock& block, Update& out) {  // :308-334
        if (block.is_empty()) return false;
        pending_.clear();
        if (block.channels != source_channels_ || config_.sample_rate != block.sample_rate) {
            source_channels_ = block.channels;
            config_.sample_rate = block.sample_rate;
            rebuild();
        }
        prepare();
        ingest_samples(block);
        if (analysis_)
            for (auto& f : analysis_->filters) f.flush_denormals();
        cap_pending();
        out.reset = reset_pending_;
        reset_pending_ = false;
        out.columns = pending_;
        const double p = std::min(std::max(column_phase_, 0.0), 1.0);  // preview (:300-306)
        out.preview_progress = (float)p;
        out.preview_some = out.preview_progress > 0.0f;
        if (out.preview_some)
            for (int ch = 0; ch < WF_CHANNELS; ++ch) out.preview[ch] = column_for(ch);
        return true;
    }

    void update_config(WaveformConfig cfg) {  // :336-352
        const WaveformConfig n = cfg.normalized();
        const bool rebuild_all = config_.sample_rate != n.sample_rate;
        const bool reset_analysis = config_.analyze_bands != n.analyze_bands || config_.track_history != n.track_history;
        config_ = n;
        if (rebuild_all) rebuild();
        else if (reset_analysis && analysis_) make_analysis();
    }

private:
    struct Analysis {
        BandFilter filters[2];
        std::vector<BandTracker> trackers;
    };
    void make_analysis() {  // band_analysis (:186-197): None when analyze_bands is off
        analysis_.reset();
        if (!config_.analyze_bands) return;
        analysis_.reset(new Analysis());
        for (auto& f : analysis_->filters) f = BandFilter(config_.sample_rate, BAND_SPLITS_HZ[0], BAND_SPLITS_HZ[1]);
        for (int c = 0; c < WF_CHANNELS; ++c) analysis_->trackers.emplace_back(config_.sample_rate, config_.track_history);
    }
    void rebuild() {  // :175-184
        column_phase_ = 0.0;
        for (auto& l : last_sample_) l.reset();
        pending_.clear();
        for (auto& c : current_) c = WfCurrent();
        if (analysis_) make_analysis();
        reset_pending_ = true;
    }
    omx_wave_column column_for(int channel) const {  // :213-235
        omx_wave_column col;
        col.min = 0.0f;
        col.max = 0.0f;
        for (int b = 0; b < 3; ++b) {
            col.color_bands[b] = 0.0f;
            col.rms_db[0][b] = DB_FLOOR;
            col.rms_db[1][b] = DB_FLOOR;
        }
        if (current_[channel].some) {
            float mn = current_[channel].min, mx = current_[channel].max;
            if (last_sample_[channel]) {
                mn = rmin(mn, *last_sample_[channel]);
                mx = rmax(mx, *last_sample_[channel]);
            }
            col.min = mn;
            col.max = mx;
        }
        if (analysis_) {
            const BandTracker& t = analysis_->trackers[channel];
            double m[3];
            t.color.mean(0, m);
            for (int b = 0; b < 3; ++b) col.color_bands[b] = (float)std::fmax(m[b], 0.0);
            if (t.history)
                for (int w = 0; w < 2; ++w) {
                    t.history->mean(w, m);
                    for (int b = 0; b < 3; ++b) col.rms_db[w][b] = power_to_db((float)std::fmax(m[b], 0.0), DB_FLOOR);
                }
        }
        return col;
    }
    void emit_column() {  // :237-250
        std::array<omx_wave_column, WF_CHANNELS> cols;
        for (int ch = 0; ch < WF_CHANNELS; ++ch) cols[ch] = column_for(ch);
        for (int ch = 0; ch < WF_CHANNELS; ++ch)
            if (current_[ch].some && current_[ch].has_last) last_sample_[ch] = current_[ch].last;
        for (int ch = 0; ch < WF_CHANNELS; ++ch) pending_.push_back(cols[ch]);
        if (pending_.size() / WF_CHANNELS >= config_.max_columns * 2) cap_pending();
        for (auto& c : current_) c = WfCurrent();
    }
    void cap_pending() {  // :293-298
        const size_t n = pending_.size() / WF_CHANNELS;
        if (n > config_.max_columns) pending_.erase(pending_.begin(), pending_.begin() + (std::ptrdiff_t)((n - config_.max_columns) * WF_CHANNELS));
    }
    void ingest_samples(const AudioBlock& block) {  // :252-273
        const double step = std::min(std::max((double)config_.scroll_speed / (double)config_.sample_rate, 0.0), 1.0);
        const size_t frames = block.frame_count();
        for (size_t f = 0; f < frames; ++f) {
            float lr[2];
            block.stereo_frame(f, lr);
            float derived[WF_CHANNELS];
            bool finite[WF_CHANNELS];
            const uint32_t chans[WF_CHANNELS] = {OMX_CHANNEL_LEFT, OMX_CHANNEL_RIGHT, OMX_CHANNEL_MID, OMX_CHANNEL_SIDE};
            for (int c = 0; c < WF_CHANNELS; ++c) {
                derived[c] = project(chans[c], lr[0], lr[1]);
                finite[c] = std::isfinite(derived[c]);
            }
            if (analysis_) {
                float left[3][1], right[3][1];
                const float in_l[1] = {finite[0] ? derived[0] : 0.0f}, in_r[1] = {finite[1] ? derived[1] : 0.0f};
                analysis_->filters[0].process(in_l, left);
                analysis_->filters[1].process(in_r, right);
                float bands[WF_CHANNELS][3];
                for (int b = 0; b < 3; ++b) {
                    bands[0][b] = left[b][0];
                    bands[1][b] = right[b][0];
                    bands[2][b] = (left[b][0] + right[b][0]) * 0.5f;
                    bands[3][b] = (left[b][0] - right[b][0]) * 0.5f;
                }
                const float zeros[3] = {0.0f, 0.0f, 0.0f};
                for (int c = 0; c < WF_CHANNELS; ++c) analysis_->trackers[c].process(finite[c] ? bands[c] : zeros);
            }
            // ingest_derived (:275-291)
            for (int c = 0; c < WF_CHANNELS; ++c) {
                if (finite[c]) {
                    const float s = derived[c];
                    WfCurrent& cur = current_[c];
                    if (cur.some) {
                        cur.min = rmin(cur.min, s);
                        cur.max = rmax(cur.max, s);
                    } else {
                        cur.some = true;
                        cur.min = cur.max = s;
                    }
                    cur.has_last = true;
                    cur.last = s;
                } else {
                    if (current_[c].some) current_[c].has_last = false;
                    last_sample_[c].reset();
                }
            }
            column_phase_ += step;
            if (column_phase_ >= 1.0) {
                emit_column();
                column_phase_ -= 1.0;
            }
        }
    }

    WaveformConfig config_;
    size_t source_channels_ = 2;
    std::unique_ptr<Analysis> analysis_;
    double column_phase_ = 0.0;
    WfCurrent current_[WF_CHANNELS];
    std::optional<float> last_sample_[WF_CHANNELS];
    std::vector<omx_wave_column> pending_;
    bool reset_pending_ = true;
};

}  // namespace omxo
