// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the shared DSP primitives of the reference:
//   src/dsp.rs:8-504            ChannelPosition, AudioBlock, WindowedMeans, Biquad, ThreeBand
//   src/util/audio/window.rs    cosine-sum windows, DC-removed windowing, bin normalisation
//   src/util/audio/level.rs     power_to_db / db_to_power / denormal flush
//   src/util/audio/channel.rs   Channel::project
//   src/util/audio/rate.rs      sanitize_sample_rate
// Arithmetic order and precision (f32 vs f64) follow the reference statement by statement;
// build with -ffp-contract=off so no mul+add is fused (Rust never contracts).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <vector>

#include "../include/omx.h"

namespace omxo {

constexpr int MAX_CH = OMX_MAX_CHANNELS;
constexpr float DEFAULT_SAMPLE_RATE = 48000.0f;   // rate.rs:6
constexpr float MAX_SAMPLE_RATE = 768000.0f;      // rate.rs:7
constexpr float DB_FLOOR = -140.0f;               // level.rs:4
constexpr float LN_TO_DB = 4.3429448f;            // level.rs:5
constexpr float TAU_F = 6.28318530717958647692f;  // core::f32::consts::TAU
constexpr float FRAC_1_SQRT_2_F = 0.707106781186547524400844362104849039f;

inline uint32_t f32_bits(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return u;
}

// Rust `as usize` on a float: saturating, NaN -> 0.
inline size_t f2usize(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551615.0) return SIZE_MAX;
    return (size_t)x;
}

// Rust f32::max / f32::min (IEEE maxNum / minNum: a NaN operand is ignored).
inline float rmax(float a, float b) { return std::fmax(a, b); }
inline float rmin(float a, float b) { return std::fmin(a, b); }
// Rust f32::clamp (NaN stays NaN).
inline float rclamp(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

// rate.rs:9-13 + util.rs:10-12
inline float sanitize_sample_rate(float r) {
    float v = (std::isfinite(r) && r > 0.0f) ? r : DEFAULT_SAMPLE_RATE;
    return rclamp(v, 1.0f, MAX_SAMPLE_RATE);
}

// level.rs:8-18
inline void flush_denormal_f32(float& v) {
    if (std::fabs(v) < 1.0e-20f) v = 0.0f;
}
inline void flush_denormal_f64(double& v) {
    if (std::fabs(v) < 1.0e-30) v = 0.0;
}
// level.rs:20-26
inline float sanitize_negative_db(float db, float def) { return (std::isfinite(db) && db < 0.0f) ? db : def; }
// level.rs:28-34
inline float power_to_db(float power, float floor) {
    if (power > 0.0f) return rmax(std::log(power) * LN_TO_DB, floor);
    return floor;
}
// level.rs:36-39
inline float db_to_power(float db) {
    const float DB_TO_LOG2 = 0.1f * 3.32192809488736234787f;  // 0.1 * LOG2_10 (f32 const arithmetic)
    return std::exp2(db * DB_TO_LOG2);
}

// ---------------------------------------------------------------- channel positions (dsp.rs:8-77)
using Positions = std::array<uint8_t, MAX_CH>;

inline Positions surround_positions() {
    return Positions{OMX_POS_FRONT_LEFT, OMX_POS_FRONT_RIGHT, OMX_POS_FRONT_CENTER, OMX_POS_LOW_FREQUENCY,
                     OMX_POS_REAR_LEFT,  OMX_POS_REAR_RIGHT,  OMX_POS_SIDE_LEFT,    OMX_POS_SIDE_RIGHT};
}

// dsp.rs:36-47
inline Positions positions_fallback(size_t channels) {
    channels = std::min<size_t>(channels, MAX_CH);
    Positions p;
    p.fill(OMX_POS_UNKNOWN);
    const Positions s = surround_positions();
    for (size_t i = 0; i < channels; ++i) p[i] = s[i];
    if (channels == 1) p[0] = OMX_POS_MONO;
    if (channels == 4) { p[2] = OMX_POS_REAR_LEFT; p[3] = OMX_POS_REAR_RIGHT; }
    if (channels == 5) { p[3] = OMX_POS_REAR_LEFT; p[4] = OMX_POS_REAR_RIGHT; }
    return p;
}

// dsp.rs:49-76
inline Positions positions_normalize(size_t channels, Positions p) {
    channels = std::min<size_t>(channels, MAX_CH);
    for (size_t i = channels; i < MAX_CH; ++i) p[i] = OMX_POS_UNKNOWN;
    for (size_t i = 0; i < channels; ++i) {
        bool dup = false;
        for (size_t j = 0; j < i; ++j) dup |= p[j] == p[i];
        if (p[i] == OMX_POS_UNKNOWN || dup) p[i] = OMX_POS_UNKNOWN;
    }
    const Positions fb = positions_fallback(channels);
    const Positions sur = surround_positions();
    for (size_t i = 0; i < channels; ++i) {
        if (p[i] != OMX_POS_UNKNOWN) continue;
        std::vector<uint8_t> cand;
        cand.push_back(fb[i]);
        for (auto c : fb) cand.push_back(c);
        for (auto c : sur) cand.push_back(c);
        for (int a = 0; a < MAX_CH; ++a) cand.push_back((uint8_t)(OMX_POS_AUX0 + a));
        for (auto c : cand) {
            if (c == OMX_POS_UNKNOWN) continue;
            bool used = false;
            for (size_t j = 0; j < channels; ++j) used |= p[j] == c;
            if (!used) { p[i] = c; break; }
        }
    }
    return p;
}

// ---------------------------------------------------------------- Channel::project (channel.rs:13-21)
inline float project(uint32_t channel, float left, float right) {
    switch (channel) {
        case OMX_CHANNEL_LEFT: return left;
        case OMX_CHANNEL_RIGHT: return right;
        case OMX_CHANNEL_MID: return (left + right) * 0.5f;
        case OMX_CHANNEL_SIDE: return (left - right) * 0.5f;
        default: return 0.0f;
    }
}

// ---------------------------------------------------------------- AudioBlock (dsp.rs:108-262)
struct AudioBlock {
    const float* samples = nullptr;
    size_t len = 0;
    size_t channels = 1;
    float sample_rate = DEFAULT_SAMPLE_RATE;
    Positions positions{};
    float stereo[MAX_CH][2] = {};
    size_t stereo_channels = 1;

    // dsp.rs:117-133
    static void stereo_indices(size_t channels, const Positions& pos, size_t out[2]) {
        auto find = [&](uint8_t want) -> int {
            for (size_t i = 0; i < channels; ++i)
                if (pos[i] == want) return (int)i;
            return -1;
        };
        const int explicit_right = find(OMX_POS_FRONT_RIGHT);
        int left = find(OMX_POS_FRONT_LEFT);
        if (left < 0) left = find(OMX_POS_MONO);
        if (left < 0) {
            for (size_t i = 0; i < channels; ++i)
                if ((int)i != explicit_right) { left = (int)i; break; }
        }
        if (left < 0) left = 0;
        int right = (explicit_right >= 0 && explicit_right != left) ? explicit_right : -1;
        if (right < 0) {
            for (size_t i = 0; i < channels; ++i)
                if ((int)i != left) { right = (int)i; break; }
        }
        if (right < 0) right = left;
        out[0] = (size_t)left;
        out[1] = (size_t)right;
    }

    // dsp.rs:135-176
    static void stereo_matrix(size_t channels, const Positions& pos, float m[MAX_CH][2]) {
        channels = std::min<size_t>(std::max<size_t>(channels, 1), MAX_CH);
        const float s = FRAC_1_SQRT_2_F;
        for (int i = 0; i < MAX_CH; ++i) m[i][0] = m[i][1] = 0.0f;
        for (size_t i = 0; i < channels; ++i) {
            switch (pos[i]) {
                case OMX_POS_FRONT_LEFT: m[i][0] = 1.0f; m[i][1] = 0.0f; break;
                case OMX_POS_FRONT_RIGHT: m[i][0] = 0.0f; m[i][1] = 1.0f; break;
                case OMX_POS_FRONT_CENTER: m[i][0] = s; m[i][1] = s; break;
                case OMX_POS_REAR_LEFT:
                case OMX_POS_SIDE_LEFT: m[i][0] = s; m[i][1] = 0.0f; break;
                case OMX_POS_REAR_RIGHT:
                case OMX_POS_SIDE_RIGHT: m[i][0] = 0.0f; m[i][1] = s; break;
                case OMX_POS_MONO: m[i][0] = 1.0f; m[i][1] = 1.0f; break;
                default: break;  // LFE / Aux / Unknown -> [0,0]
            }
        }
        auto populated = [&](int side) {
            for (size_t i = 0; i < channels; ++i)
                if (m[i][side] != 0.0f) return true;
            return false;
        };
        const bool l = populated(0), r = populated(1);
        if (!l && !r) {
            size_t idx[2];
            stereo_indices(channels, pos, idx);
            m[idx[0]][0] = 1.0f;
            m[idx[1]][1] = 1.0f;
        } else if (!l && r) {
            for (int i = 0; i < MAX_CH; ++i) m[i][0] = m[i][1];
        } else if (l && !r) {
            for (int i = 0; i < MAX_CH; ++i) m[i][1] = m[i][0];
        }
    }

    // dsp.rs:190-213
    static AudioBlock with_positions(const float* samples, size_t len, size_t channels, float sample_rate,
                                     const Positions& positions) {
        AudioBlock b;
        channels = std::min<size_t>(std::max<size_t>(channels, 1), MAX_CH);
        b.samples = samples;
        b.len = len;
        b.channels = channels;
        b.sample_rate = sanitize_sample_rate(sample_rate);
        b.positions = positions;
        stereo_matrix(channels, positions, b.stereo);
        size_t sc = std::min<size_t>(channels, 2);
        const size_t hi = std::min(channels, len);
        for (size_t ch = hi; ch-- > 2;) {
            bool any = false;
            for (size_t i = ch; i < len; i += channels)
                if (f32_bits(samples[i]) != 0) { any = true; break; }
            if (any) { sc = ch + 1; break; }
        }
        b.stereo_channels = sc;
        return b;
    }

    // dsp.rs:180-188 (test constructor)
    static AudioBlock make(const float* samples, size_t len, size_t channels, float sample_rate) {
        channels = std::min<size_t>(std::max<size_t>(channels, 1), MAX_CH);
        return with_positions(samples, len, channels, sample_rate, positions_fallback(channels));
    }

    size_t frame_count() const { return len / std::max<size_t>(channels, 1); }       // :219-221
    bool is_empty() const { return len < std::max<size_t>(channels, 1); }            // :259-261

    // dsp.rs:223-249 — one frame of stereo_frames()
    void stereo_frame(size_t frame, float out[2]) const {
        const float* f = samples + frame * channels;
        if (stereo_channels == 1) {
            const float s = f[0];
            out[0] = 0.0f + s * stereo[0][0];
            out[1] = 0.0f + s * stereo[0][1];
        } else if (stereo_channels == 2) {
            float s = f[0];
            float left = 0.0f + s * stereo[0][0], right = 0.0f + s * stereo[0][1];
            s = f[1];
            out[0] = left + s * stereo[1][0];
            out[1] = right + s * stereo[1][1];
        } else {
            float left = 0.0f, right = 0.0f;
            for (size_t c = 0; c < stereo_channels; ++c) {
                left = left + f[c] * stereo[c][0];
                right = right + f[c] * stereo[c][1];
            }
            out[0] = left;
            out[1] = right;
        }
    }
    float projected(size_t frame, uint32_t channel) const {  // :251-257
        float lr[2];
        stereo_frame(frame, lr);
        return project(channel, lr[0], lr[1]);
    }
};

// ---------------------------------------------------------------- windows (window.rs:20-109)
// Rust `iter().sum::<f32>()` folds from -0.0 (identity of float addition since Rust 1.83).
inline float rust_sum_f32(const float* p, size_t n) {
    float acc = -0.0f;
    for (size_t i = 0; i < n; ++i) acc = acc + p[i];
    return acc;
}

// window.rs:20-43
inline std::vector<float> window_coefficients(uint32_t kind, size_t len) {
    if (len <= 1) return std::vector<float>(len, 1.0f);
    const float hann[] = {0.5f, -0.5f};
    const float hamming[] = {25.0f / 46.0f, -21.0f / 46.0f};
    const float blackman[] = {0.42f, -0.5f, 0.08f};
    const float bh[] = {0.35875f, -0.48829f, 0.14128f, -0.01168f};
    const float* c = nullptr;
    size_t nc = 0;
    switch (kind) {
        case OMX_WINDOW_HANN: c = hann; nc = 2; break;
        case OMX_WINDOW_HAMMING: c = hamming; nc = 2; break;
        case OMX_WINDOW_BLACKMAN: c = blackman; nc = 3; break;
        case OMX_WINDOW_BLACKMAN_HARRIS: c = bh; nc = 4; break;
        default: return std::vector<float>(len, 1.0f);
    }
    const float step = TAU_F / (float)len;
    std::vector<float> w(len);
    for (size_t n = 0; n < len; ++n) {
        const float phi = (float)n * step;
        float sum = 0.0f;
        for (size_t k = 0; k < nc; ++k) sum = sum + c[k] * std::cos(phi * (float)k);
        w[n] = sum;
    }
    return w;
}

// window.rs:66-88 — front `dst.size()` samples of the deque, DC removed, windowed.
inline void copy_dc_removed_windowed(float* dst, size_t len, const std::deque<float>& src, const float* window) {
    if (len == 0) return;
    float sum = -0.0f;
    for (size_t i = 0; i < len; ++i) {
        dst[i] = src[i];
        sum = sum + src[i];
    }
    const float mean = sum / (float)len;
    for (size_t i = 0; i < len; ++i) dst[i] = (dst[i] - mean) * window[i];
}

// window.rs:90-109
inline std::vector<float> compute_fft_bin_normalization(const std::vector<float>& window, size_t fft_size) {
    const size_t bins = fft_size / 2 + 1;
    const float window_sum = rust_sum_f32(window.data(), window.size());
    float inv_sum;
    if (std::fabs(window_sum) > std::numeric_limits<float>::epsilon()) inv_sum = 1.0f / window_sum;
    else if (fft_size > 0) inv_sum = 1.0f / (float)fft_size;
    else inv_sum = 0.0f;
    const float dc = inv_sum * inv_sum;
    const float ac = 4.0f * dc;
    std::vector<float> norms(bins, ac);
    norms[0] = dc;
    if (fft_size % 2 == 0 && bins > 1) norms[bins - 1] = dc;
    return norms;
}

// ---------------------------------------------------------------- WindowedMeans (dsp.rs:264-371)
struct CompensatedPair {
    double sums[2] = {0.0, 0.0};
    double corrections[2] = {0.0, 0.0};
    // Kahan-Babuska-Neumaier (dsp.rs:277-285)
    void add(int index, double value) {
        const double next = sums[index] + value;
        if (std::fabs(sums[index]) >= std::fabs(value)) corrections[index] += (sums[index] - next) + value;
        else corrections[index] += (value - next) + sums[index];
        sums[index] = next;
    }
    void refresh() {  // :287-289
        sums[0] = sums[1]; sums[1] = 0.0;
        corrections[0] = corrections[1]; corrections[1] = 0.0;
    }
    double value() const { return sums[0] + corrections[0]; }  // :291-293
};

template <int VALUES, int WINDOWS, class T = double>
struct WindowedMeans {
    std::vector<std::array<T, VALUES>> buffer;
    size_t capacities[WINDOWS];
    CompensatedPair sums[WINDOWS][VALUES];
    size_t refresh_counts[WINDOWS];
    size_t head = 0, count = 0;

    explicit WindowedMeans(const size_t (&caps)[WINDOWS]) {  // :311-322
        size_t len = 1;
        for (int w = 0; w < WINDOWS; ++w) {
            capacities[w] = std::max<size_t>(caps[w], 1);
            refresh_counts[w] = 0;
            len = (w == 0) ? capacities[w] : std::max(len, capacities[w]);
        }
        std::array<T, VALUES> zero;
        zero.fill((T)0.0f);
        buffer.assign(len, zero);
    }

    // :359-365
    static WindowedMeans with_leading_zeros(const size_t (&caps)[WINDOWS], size_t n) {
        WindowedMeans m(caps);
        m.head = n % m.buffer.size();
        m.count = std::min(n, m.buffer.size());
        for (int w = 0; w < WINDOWS; ++w) m.refresh_counts[w] = n % m.capacities[w];
        return m;
    }

    void push(std::array<T, VALUES> values) {  // :324-357
        double mapped[VALUES];
        for (int v = 0; v < VALUES; ++v) {
            const double x = (double)values[v];
            if (std::isfinite(x)) mapped[v] = x;
            else { values[v] = (T)0.0f; mapped[v] = 0.0; }
        }
        const size_t len = buffer.size();
        for (int w = 0; w < WINDOWS; ++w) {
            const size_t cap = capacities[w];
            const bool has_old = count >= cap;
            std::array<T, VALUES> old{};
            if (has_old) old = buffer[(head + len - cap) % len];
            for (int v = 0; v < VALUES; ++v) {
                sums[w][v].add(0, mapped[v]);
                sums[w][v].add(1, mapped[v]);
                if (has_old) sums[w][v].add(0, -(double)old[v]);
            }
            refresh_counts[w] += 1;
            if (refresh_counts[w] == cap) {
                for (int v = 0; v < VALUES; ++v) sums[w][v].refresh();
                refresh_counts[w] = 0;
            }
        }
        buffer[head] = values;
        head = (head + 1) % len;
        count = std::min(count + 1, len);
    }

    void mean(int window, double out[VALUES]) const {  // :367-370
        const size_t c = std::max<size_t>(std::min(count, capacities[window]), 1);
        for (int v = 0; v < VALUES; ++v) out[v] = sums[window][v].value() / (double)c;
    }
};

// ---------------------------------------------------------------- Biquad / Cascade / ThreeBand (dsp.rs:373-504)
enum class FilterKind { LowPass, HighPass };

struct Biquad {
    float b[3] = {0, 0, 0};
    float a[2] = {0, 0};
    float z[2] = {0, 0};
    Biquad() = default;
    Biquad(FilterKind kind, float sample_rate, float frequency) {  // :402-420
        const float ratio = rclamp(frequency / sample_rate, 1.0e-6f, 0.49f);
        const float ang = TAU_F * ratio;
        const float sn = std::sin(ang), cs = std::cos(ang);
        const float alpha = sn * FRAC_1_SQRT_2_F;
        float gain, sign;
        if (kind == FilterKind::LowPass) { gain = 1.0f - cs; sign = 1.0f; }
        else { gain = 1.0f + cs; sign = -1.0f; }
        const float inv_a0 = 1.0f / (1.0f + alpha);
        b[0] = gain * 0.5f * inv_a0;
        b[1] = gain * inv_a0 * sign;
        b[2] = gain * 0.5f * inv_a0;
        a[0] = -2.0f * cs * inv_a0;
        a[1] = (1.0f - alpha) * inv_a0;
    }
    float process(float sample) {  // :422-432
        const float output = b[0] * sample + z[0];
        z[0] = b[1] * sample - a[0] * output + z[1];
        z[1] = b[2] * sample - a[1] * output;
        if (std::isfinite(output)) return output;
        z[0] = z[1] = 0.0f;
        return 0.0f;
    }
    void flush_denormals() { flush_denormal_f32(z[0]); flush_denormal_f32(z[1]); }  // :391-393
    void clear() { z[0] = z[1] = 0.0f; }                                             // :394-396
};

template <int N>
struct Cascade {  // :439-457
    Biquad f[N];
    Cascade() = default;
    Cascade(FilterKind kind, float sr, float freq) {
        for (int i = 0; i < N; ++i) f[i] = Biquad(kind, sr, freq);
    }
    float process(float s) {
        for (int i = 0; i < N; ++i) s = f[i].process(s);
        return s;
    }
    void flush_denormals() { for (auto& x : f) x.flush_denormals(); }
    void clear() { for (auto& x : f) x.clear(); }
};

// ThreeBand<[F;LANES], CASCADE_HIGH> with F = Cascade<ORDER> (:459-504).  Stereometer uses
// LANES=2, ORDER=2, CASCADE_HIGH=true; waveform uses ORDER=1, CASCADE_HIGH=false.
template <int LANES, int ORDER, bool CASCADE_HIGH>
struct ThreeBand {
    Cascade<ORDER> filters[4][LANES];
    ThreeBand() = default;
    ThreeBand(float sample_rate, float low, float high) {
        for (int l = 0; l < LANES; ++l) {
            filters[0][l] = Cascade<ORDER>(FilterKind::LowPass, sample_rate, low);
            filters[1][l] = Cascade<ORDER>(FilterKind::HighPass, sample_rate, low);
            filters[2][l] = Cascade<ORDER>(FilterKind::LowPass, sample_rate, high);
            filters[3][l] = Cascade<ORDER>(FilterKind::HighPass, sample_rate, high);
        }
    }
    // out[band][lane]  (:489-495; array filter processes lane by lane, dsp.rs:464-466)
    void process(const float in[LANES], float out[3][LANES]) {
        float above_low[LANES];
        for (int l = 0; l < LANES; ++l) out[0][l] = filters[0][l].process(in[l]);
        for (int l = 0; l < LANES; ++l) above_low[l] = filters[1][l].process(in[l]);
        for (int l = 0; l < LANES; ++l) out[1][l] = filters[2][l].process(above_low[l]);
        for (int l = 0; l < LANES; ++l) out[2][l] = filters[3][l].process(CASCADE_HIGH ? above_low[l] : in[l]);
    }
    void flush_denormals() {
        for (auto& row : filters) for (auto& x : row) x.flush_denormals();
    }
    void clear() {
        for (auto& row : filters) for (auto& x : row) x.clear();
    }
};

constexpr float BAND_SPLITS_HZ[2] = {200.0f, 2000.0f};  // util/audio.rs:26

}  // namespace omxo
