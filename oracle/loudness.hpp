// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of reference src/visuals/loudness/processor.rs:10-311 (BS.1770 K-weighting,
// 3 s / 0.4 s LUFS, 0.3 s / 1 s K-RMS, 4x/2x polyphase true peak).
#pragma once
#include <memory>
#include <vector>

#include "primitives.hpp"

namespace omxo {

constexpr double LOUDNESS_OFFSET = -0.691;                 // :10
constexpr float LOUDNESS_DEFAULT_FLOOR_DB = -99.9f;        // :11
constexpr float LOUDNESS_WINDOWS[4] = {3.0f, 0.4f, 0.3f, 1.0f};  // :13
enum { WIN_SHORT_TERM = 0, WIN_MOMENTARY = 1, WIN_RMS_FAST = 2, WIN_RMS_SLOW = 3 };

struct KWeighting {
    double b[5], a[5];
};

// :22-55
inline KWeighting k_weighting_coefficients(double fs) {
    double f0 = 1681.974450955533, g = 3.999843853973347, q = 0.7071752369554196;
    double k = std::tan(M_PI * f0 / fs);
    const double vh = std::pow(10.0, g / 20.0);
    const double vb = std::pow(vh, 0.4996667741545416);
    double a0 = 1.0 + k / q + k * k;
    const double pb[3] = {(vh + vb * k / q + k * k) / a0, 2.0 * (k * k - vh) / a0, (vh - vb * k / q + k * k) / a0};
    const double pa[3] = {1.0, 2.0 * (k * k - 1.0) / a0, (1.0 - k / q + k * k) / a0};
    f0 = 38.13547087602444;
    q = 0.5003270373238773;
    k = std::tan(M_PI * f0 / fs);
    a0 = 1.0 + k / q + k * k;
    const double rb[3] = {1.0, -2.0, 1.0};
    const double ra[3] = {1.0, 2.0 * (k * k - 1.0) / a0, (1.0 - k / q + k * k) / a0};
    auto conv = [](const double p[3], const double r[3], double out[5]) {
        out[0] = p[0] * r[0];
        out[1] = p[0] * r[1] + p[1] * r[0];
        out[2] = p[0] * r[2] + p[1] * r[1] + p[2] * r[0];
        out[3] = p[1] * r[2] + p[2] * r[1];
        out[4] = p[2] * r[2];
    };
    KWeighting w;
    conv(pb, rb, w.b);
    conv(pa, ra, w.a);
    return w;
}

// :57-66
inline float mean_square_to_lufs(double mean_square, float floor) {
    if (mean_square > 0.0) return (float)std::fmax(std::fma(std::log10(mean_square), 10.0, LOUDNESS_OFFSET), (double)floor);
    return floor;
}

// :68-71
inline size_t window_length(float sample_rate, float window_secs) {
    const float len = sample_rate * window_secs;
    return len < 1.0f ? 1 : f2usize((double)len);
}

constexpr size_t TRUE_PEAK_TAPS = 48;                       // :75
constexpr size_t TRUE_PEAK_4X_DELAY = TRUE_PEAK_TAPS / 4;   // :76
constexpr size_t TRUE_PEAK_2X_DELAY = TRUE_PEAK_TAPS / 2;   // :77

// :79-84
inline float true_peak_coefficient(size_t j, size_t factor) {
    const double offset = (double)j - (double)TRUE_PEAK_TAPS * 0.5;
    const double window = 0.5 * (1.0 - std::cos(2.0 * M_PI * (double)j / (double)TRUE_PEAK_TAPS));
    const double x = offset * M_PI / (double)factor;
    return (float)(window * std::sin(x) / x);
}

struct TruePeakFirs {  // :86-97
    float fir4[TRUE_PEAK_4X_DELAY][3];
    float fir2[TRUE_PEAK_2X_DELAY];
    TruePeakFirs() {
        for (size_t tap = 0; tap < TRUE_PEAK_4X_DELAY; ++tap)
            for (size_t phase = 0; phase < 3; ++phase) fir4[tap][phase] = true_peak_coefficient(tap * 4 + phase + 1, 4);
        for (size_t tap = 0; tap < TRUE_PEAK_2X_DELAY; ++tap) fir2[tap] = true_peak_coefficient(tap * 2 + 1, 2);
    }
};
inline const TruePeakFirs& true_peak_firs() {
    static const TruePeakFirs firs;
    return firs;
}

struct TruePeakMeter {  // :99-151
    float delay[TRUE_PEAK_2X_DELAY * 2] = {};
    size_t write = 0, delay_len = 0;
    float peak = 0.0f;
    explicit TruePeakMeter(double sample_rate) {
        delay_len = sample_rate < 96000.0 ? TRUE_PEAK_4X_DELAY : (sample_rate < 192000.0 ? TRUE_PEAK_2X_DELAY : 0);
        write = delay_len;
    }
    void process(float sample) {
        peak = rmax(peak, std::fabs(sample));
        if (delay_len == 0) return;
        write = (write == 0 ? delay_len : write) - 1;
        const size_t pos = write;
        delay[pos] = sample;
        delay[pos + delay_len] = sample;
        const TruePeakFirs& firs = true_peak_firs();
        if (delay_len == TRUE_PEAK_4X_DELAY) {
            float output[3] = {0.0f, 0.0f, 0.0f};
            for (size_t i = 0; i < delay_len; ++i) {
                const float s = delay[pos + i];
                for (int phase = 0; phase < 3; ++phase) output[phase] += s * firs.fir4[i][phase];
            }
            for (int phase = 0; phase < 3; ++phase) peak = rmax(peak, std::fabs(output[phase]));
        } else {
            float output = 0.0f;
            for (size_t i = 0; i < delay_len; ++i) output += delay[pos + i] * firs.fir2[i];
            peak = rmax(peak, std::fabs(output));
        }
    }
};

// :153-162
inline float k_weighted(float sample, double state[4], const KWeighting& c) {
    const double x = (double)sample;
    const double y = c.b[0] * x + state[0];
    state[0] = c.b[1] * x + state[1] - c.a[1] * y;
    state[1] = c.b[2] * x + state[2] - c.a[2] * y;
    state[2] = c.b[3] * x + state[3] - c.a[3] * y;
    state[3] = c.b[4] * x - c.a[4] * y;
    return (float)y;
}

// :174-183
inline double channel_weight(uint8_t position) {
    switch (position) {
        case OMX_POS_LOW_FREQUENCY: return 0.0;
        case OMX_POS_REAR_LEFT:
        case OMX_POS_REAR_RIGHT:
        case OMX_POS_SIDE_LEFT:
        case OMX_POS_SIDE_RIGHT: return 1.41;
        default: return 1.0;
    }
}

struct LoudnessConfig {  // :210-216
    float sample_rate = DEFAULT_SAMPLE_RATE;
    float floor_db = LOUDNESS_DEFAULT_FLOOR_DB;
};

struct LoudnessActive {
    WindowedMeans<1, 4> windows;
    double filter[4] = {0, 0, 0, 0};
    TruePeakMeter true_peak;
    LoudnessActive(WindowedMeans<1, 4> w, double sr) : windows(std::move(w)), true_peak(sr) {}
};
struct LoudnessChannelState {  // :166-170
    std::unique_ptr<LoudnessActive> active;
    size_t silent_frames = 0;
};

class LoudnessProcessor {
public:
    explicit LoudnessProcessor(LoudnessConfig cfg) : config_(cfg) {  // :225-232
        weighting_ = k_weighting_coefficients((double)sanitize_sample_rate(cfg.sample_rate));
    }
    void reset_audio() {  // :234-236
        for (auto& c : channels_) c = LoudnessChannelState();
    }
    void ensure_state(size_t requested_channels, float sample_rate) {  // :238-251
        const size_t channels = std::min<size_t>(std::max<size_t>(requested_channels, 1), MAX_CH);
        sample_rate = sanitize_sample_rate(sample_rate);
        const bool rate_changed = config_.sample_rate != sample_rate;
        if (rate_changed) {
            config_.sample_rate = sample_rate;
            weighting_ = k_weighting_coefficients((double)sample_rate);
        }
        if (rate_changed || channels_.size() != channels) {
            channels_.clear();
            channels_.resize(channels);
        }
    }
    void force_active_for_test() {  // the "eager" arm of the reference test :405-416
        size_t caps[4];
        capacities(caps);
        for (auto& c : channels_)
            c.active.reset(new LoudnessActive(WindowedMeans<1, 4>(caps), (double)config_.sample_rate));
    }

    bool process_block(const AudioBlock& block, omx_loudness_snapshot& snapshot) {  // :253-311
        if (block.is_empty()) return false;
        ensure_state(block.channels, block.sample_rate);
        size_t caps[4];
        capacities(caps);
        const double sample_rate = (double)config_.sample_rate;
        const size_t frames = block.len / block.channels;
        for (size_t f = 0; f < frames; ++f) {
            const float* frame = block.samples + f * block.channels;
            const size_t n = std::min(channels_.size(), block.channels);
            for (size_t ch = 0; ch < n; ++ch) {
                LoudnessChannelState& channel = channels_[ch];
                const float sample = frame[ch];
                if (!channel.active) {
                    if (f32_bits(sample) == 0) {
                        channel.silent_frames += 1;
                        continue;
                    }
                    channel.active.reset(new LoudnessActive(
                        WindowedMeans<1, 4>::with_leading_zeros(caps, channel.silent_frames), sample_rate));
                }
                LoudnessActive& a = *channel.active;
                const double filtered = (double)k_weighted(sample, a.filter, weighting_);
                a.windows.push({filtered * filtered});
                a.true_peak.process(sample);
            }
        }
        for (auto& channel : channels_)
            if (channel.active)
                for (double& s : channel.active->filter) flush_denormal_f64(s);

        const float floor = config_.floor_db;
        snapshot.short_term_loudness = floor;  // with_floor :197-207
        snapshot.momentary_loudness = floor;
        for (int i = 0; i < MAX_CH; ++i) {
            snapshot.rms_fast_db[i] = floor;
            snapshot.rms_slow_db[i] = floor;
            snapshot.true_peak_db[i] = floor;
            snapshot.positions[i] = OMX_POS_UNKNOWN;
        }
        snapshot.channel_count = 0;
        snapshot._pad = 0;
        double weighted_short_term = 0.0, weighted_momentary = 0.0;
        for (size_t ci = 0; ci < channels_.size(); ++ci) {
            if (!channels_[ci].active) continue;
            LoudnessActive& a = *channels_[ci].active;
            const double weight = channel_weight(block.positions[ci]);
            double m[1];
            a.windows.mean(WIN_SHORT_TERM, m);
            weighted_short_term += m[0] * weight;
            a.windows.mean(WIN_MOMENTARY, m);
            weighted_momentary += m[0] * weight;
            a.windows.mean(WIN_RMS_FAST, m);
            snapshot.rms_fast_db[ci] = power_to_db((float)m[0], floor);
            a.windows.mean(WIN_RMS_SLOW, m);
            snapshot.rms_slow_db[ci] = power_to_db((float)m[0], floor);
            const float peak = a.true_peak.peak;
            a.true_peak.peak = 0.0f;
            snapshot.true_peak_db[ci] = power_to_db(peak * peak, floor);
        }
        snapshot.short_term_loudness = mean_square_to_lufs(weighted_short_term, floor);
        snapshot.momentary_loudness = mean_square_to_lufs(weighted_momentary, floor);
        snapshot.channel_count = (uint32_t)channels_.size();
        for (int i = 0; i < MAX_CH; ++i) snapshot.positions[i] = block.positions[i];
        return true;
    }

private:
    void capacities(size_t caps[4]) const {
        for (int w = 0; w < 4; ++w) caps[w] = window_length(config_.sample_rate, LOUDNESS_WINDOWS[w]);
    }
    LoudnessConfig config_;
    std::vector<LoudnessChannelState> channels_;
    KWeighting weighting_;
};

}  // namespace omxo
