// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of reference src/visuals/stereometer/processor.rs:8-212 (L/R point cloud
// decimation, f64 EMA phase correlation, LR4 3-band split with per-band correlation).
#pragma once
#include <deque>
#include <utility>
#include <vector>

#include "primitives.hpp"

namespace omxo {

constexpr float BAND_DISPLAY_GAIN = 0.8f;  // :8
constexpr int BAND_COUNT = 3;              // :9

struct StereometerConfig {  // :11-21
    float sample_rate = DEFAULT_SAMPLE_RATE;
    float segment_duration = 0.02f;
    size_t target_sample_count = 2000;
    float correlation_window = 0.05f;
    bool analyze_bands = false;
    bool emit_band_points = false;
};

struct Correlator {  // :34-61
    double moments[3] = {0, 0, 0};
    void update(float left_f, float right_f, double alpha) {
        const double left = (double)left_f, right = (double)right_f;
        moments[0] += alpha * (left * right - moments[0]);
        moments[1] += alpha * (left * left - moments[1]);
        moments[2] += alpha * (right * right - moments[2]);
    }
    float value() const {
        const double denom = std::sqrt(moments[1] * moments[2]);
        if (denom <= 1e-12) return 0.0f;
        const double v = moments[0] / denom;
        if (!std::isfinite(v)) return 0.0f;
        return (float)std::min(std::max(v, -1.0), 1.0);
    }
    void flush_denormals() {
        for (double& m : moments) flush_denormal_f64(m);
    }
};

// :210-212
inline double ema_alpha(float sample_rate, float window) {
    return 1.0 - std::exp(-1.0 / std::fmax((double)sample_rate * (double)window, 1.0));
}

using BandSplitter = ThreeBand<2, 2, true>;  // :32  ThreeBand<[Cascade<Biquad,2>;2], true>

struct StereometerSnapshot {  // :23-26
    std::vector<std::pair<float, float>> points[BAND_COUNT + 1];
    float correlations[BAND_COUNT + 1] = {0, 0, 0, 0};
};

class StereometerProcessor {
public:
    explicit StereometerProcessor(StereometerConfig cfg) { init(cfg); }  // :75-86
    StereometerConfig config() const { return config_; }

    void reset_audio() {  // :92-97
        for (auto& h : histories_) h.clear();
        splitter_.clear();
        for (auto& c : correlators_) c = Correlator();
        for (auto& s : snapshot_) s.clear();
    }

    bool process_block(const AudioBlock& block, StereometerSnapshot& out) {  // :99-182
        const size_t channel_count = block.channels;
        if (block.is_empty()) return false;
        const float sample_rate = block.sample_rate;
        if (config_.sample_rate != sample_rate) {
            StereometerConfig c = config_;
            c.sample_rate = sample_rate;
            update_config(c);
        }
        if (history_channels_ != channel_count) {
            histories_[0].clear();
            history_channels_ = channel_count;
        }
        const bool analyze = config_.analyze_bands;
        const double alpha = alpha_;
        const size_t nframes = block.frame_count();
        for (size_t f = 0; f < nframes; ++f) {
            float lr[2];
            block.stereo_frame(f, lr);
            histories_[0].push_back({lr[0], lr[1]});
            correlators_[0].update(lr[0], lr[1], alpha);
            if (analyze) {
                float bands[3][2];
                splitter_.process(lr, bands);
                for (int b = 0; b < BAND_COUNT; ++b) {
                    correlators_[1 + b].update(bands[b][0], bands[b][1], alpha);
                    if (config_.emit_band_points) histories_[1 + b].push_back({bands[b][0], bands[b][1]});
                }
            }
        }
        correlators_[0].flush_denormals();
        if (analyze) {
            for (int b = 1; b <= BAND_COUNT; ++b) correlators_[b].flush_denormals();
            splitter_.flush_denormals();
        }
        const size_t frames = f2usize((double)rmax(std::round(config_.sample_rate * config_.segment_duration), 1.0f));
        const int history_count = config_.emit_band_points ? BAND_COUNT + 1 : 1;
        for (int h = 0; h < history_count; ++h) {
            auto& hist = histories_[h];
            const size_t drop = hist.size() > frames ? hist.size() - frames : 0;
            hist.erase(hist.begin(), hist.begin() + (std::ptrdiff_t)drop);
        }
        if (histories_[0].size() < frames) return false;
        const size_t target = std::min(std::max<size_t>(config_.target_sample_count, 1), frames);
        for (int band = 0; band < history_count; ++band) {
            auto& buf = snapshot_[band];
            auto& hist = histories_[band];
            buf.clear();
            if (hist.size() < frames) continue;
            for (size_t i = 0; i < target; ++i) {
                std::pair<float, float> p = hist[i * frames / target];
                if (band != 0) p = {p.first * BAND_DISPLAY_GAIN, p.second * BAND_DISPLAY_GAIN};
                buf.push_back(p);
            }
        }
        for (int band = 0; band <= BAND_COUNT; ++band) {
            out.points[band] = snapshot_[band];
            out.correlations[band] = (band == 0 || analyze) ? correlators_[band].value() : 0.0f;
        }
        return true;
    }

    void update_config(StereometerConfig cfg) {  // :183-207
        cfg.analyze_bands = cfg.analyze_bands || cfg.emit_band_points;
        const bool rate_changed = config_.sample_rate != cfg.sample_rate;
        const bool window_changed =
            std::fabs(config_.correlation_window - cfg.correlation_window) > std::numeric_limits<float>::epsilon();
        const bool bands_changed = config_.analyze_bands != cfg.analyze_bands;
        config_ = cfg;
        if (rate_changed) {
            init(config_);
        } else {
            if (window_changed) alpha_ = ema_alpha(cfg.sample_rate, cfg.correlation_window);
            if (bands_changed) {
                splitter_ = BandSplitter(cfg.sample_rate, BAND_SPLITS_HZ[0], BAND_SPLITS_HZ[1]);
                for (int b = 1; b <= BAND_COUNT; ++b) correlators_[b] = Correlator();
            }
        }
        if (!cfg.emit_band_points) {
            for (int b = 1; b <= BAND_COUNT; ++b) {
                histories_[b].clear();
                snapshot_[b].clear();
            }
        }
    }

private:
    void init(StereometerConfig cfg) {
        cfg.analyze_bands = cfg.analyze_bands || cfg.emit_band_points;
        for (auto& s : snapshot_) s.clear();
        for (auto& h : histories_) h.clear();
        history_channels_ = 0;
        splitter_ = BandSplitter(cfg.sample_rate, BAND_SPLITS_HZ[0], BAND_SPLITS_HZ[1]);
        for (auto& c : correlators_) c = Correlator();
        alpha_ = ema_alpha(cfg.sample_rate, cfg.correlation_window);
        config_ = cfg;
    }

    StereometerConfig config_;
    std::vector<std::pair<float, float>> snapshot_[BAND_COUNT + 1];
    std::deque<std::pair<float, float>> histories_[BAND_COUNT + 1];
    size_t history_channels_ = 0;
    BandSplitter splitter_;
    Correlator correlators_[BAND_COUNT + 1];
    double alpha_ = 0.0;
};

}  // namespace omxo
