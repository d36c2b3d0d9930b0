// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of reference src/visuals/spectrogram/processor.rs:45-608
// (SpectrogramProcessor: classic STFT and time-frequency reassignment).
#pragma once
#include <complex>
#include <deque>
#include <vector>

#include "fft.hpp"
#include "primitives.hpp"

namespace omxo {

using cf = std::complex<float>;

constexpr size_t DEFAULT_SPECTROGRAM_FFT_SIZE = 2048;               // :58
constexpr size_t DEFAULT_SPECTROGRAM_HOP_SIZE = 64;                 // :59
constexpr size_t MAX_SPECTROGRAM_HISTORY_COLUMNS = 8192;            // :60
constexpr size_t SPECTROGRAM_HISTORY_BYTE_BUDGET = 128u * 1024 * 1024;  // :61
constexpr float CLASSIC_DB_STORE_LO = -144.0f;                      // :66
constexpr float CLASSIC_DB_STORE_HI = 12.0f;                        // :67
constexpr float CLASSIC_DB_STORE_RANGE = CLASSIC_DB_STORE_HI - CLASSIC_DB_STORE_LO;
constexpr float ANALYSIS_FLOOR_POWER = 1e-14f;                      // :69

struct SpectrogramConfig {  // :45-56
    float sample_rate = DEFAULT_SAMPLE_RATE;
    size_t fft_size = DEFAULT_SPECTROGRAM_FFT_SIZE;
    size_t hop_size = DEFAULT_SPECTROGRAM_HOP_SIZE;
    uint32_t window = OMX_WINDOW_HANN;
    size_t history_length = 0;
    bool use_reassignment = true;
    size_t zero_padding_factor = 1;

    void normalize() {  // :71-82
        sample_rate = sanitize_sample_rate(sample_rate);
        if (fft_size == 0) fft_size = DEFAULT_SPECTROGRAM_FFT_SIZE;
        if (hop_size == 0) hop_size = std::max<size_t>(std::min(DEFAULT_SPECTROGRAM_HOP_SIZE, fft_size), 1);
        zero_padding_factor = std::max<size_t>(zero_padding_factor, 1);
    }
};

// :103-108
inline uint16_t pack_classic_db(float db) {
    const float SCALE = 65535.0f / CLASSIC_DB_STORE_RANGE;
    float v = std::round((db - CLASSIC_DB_STORE_LO) * SCALE);  // f32::round: half away from zero
    v = rclamp(v, 0.0f, 65535.0f);
    if (!(v == v)) return 0;  // NaN as u16 -> 0
    return (uint16_t)v;
}

// :111-117
inline float reassigned_power_scale(const std::vector<float>& window, size_t fft_size) {
    double sum = 0.0, squares = 0.0;
    for (float xf : window) {
        const double x = (double)xf;
        sum = sum + x;
        squares = squares + x * x;
    }
    return (float)(sum * sum / ((double)fft_size * squares));
}

// :144-151
inline uint64_t col_byte_stride(uint32_t kind, uint32_t points) {
    if (kind == OMX_COLUMN_REASSIGNED) return (uint64_t)points * 12u;
    return (((uint64_t)points + 1) / 2) * 4;
}
// :153-158
inline size_t history_columns(uint32_t kind, uint32_t points, size_t requested) {
    const size_t clamped = std::min(std::max<size_t>(requested, 1), MAX_SPECTROGRAM_HISTORY_COLUMNS);
    const size_t budget = SPECTROGRAM_HISTORY_BYTE_BUDGET * (1 + (kind == OMX_COLUMN_REASSIGNED ? 1 : 0)) /
                          (size_t)std::max<uint64_t>(col_byte_stride(kind, points), 1);
    return std::min(clamped, budget);
}

// :546-557
inline void hilbert_transform(cf* analytic, size_t n) {
    fft_inplace(analytic, n, false);
    analytic[0] = cf(0, 0);
    for (size_t i = n / 2 + 1; i < n; ++i) analytic[i] = cf(0, 0);
    fft_inplace(analytic, n, true);
}

// :559-567
inline void apply_complex_window(const cf* analytic, const std::vector<float>& window, cf* out, size_t out_len) {
    const size_t n = std::min(window.size(), out_len);
    for (size_t i = 0; i < n; ++i) out[i] = cf(analytic[i].real() * window[i], analytic[i].imag() * window[i]);
    for (size_t i = window.size(); i < out_len; ++i) out[i] = cf(0, 0);
}

// :569-599
inline std::vector<float> compute_derivative_spectral(const std::vector<float>& window) {
    const size_t n = window.size();
    if (n <= 1) return std::vector<float>(n, 0.0f);
    std::vector<cf> buf(n);
    for (size_t i = 0; i < n; ++i) buf[i] = cf(window[i], 0.0f);
    fft_inplace(buf.data(), n, false);
    const float scale = TAU_F / (float)n;
    const size_t half = n / 2;
    buf[0] = cf(0, 0);
    if (n % 2 == 0) buf[half] = cf(0, 0);
    for (size_t k = 1; k < n; ++k) {
        const float omega = scale * ((float)k - (k > half ? (float)n : 0.0f));
        buf[k] = cf(-omega * buf[k].imag(), omega * buf[k].real());
    }
    fft_inplace(buf.data(), n, true);
    const float inv_n = 1.0f / (float)n;
    std::vector<float> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = buf[i].real() * inv_n;
    return out;
}

// :601-608
inline std::vector<float> compute_time_weighted(const std::vector<float>& window) {
    const float center = (float)(window.empty() ? 0 : window.size() - 1) * 0.5f;
    std::vector<float> out(window.size());
    for (size_t i = 0; i < window.size(); ++i) out[i] = ((float)i - center) * window[i];
    return out;
}

struct SpectrogramColumn {
    uint32_t kind = OMX_COLUMN_REASSIGNED;
    std::vector<omx_spectrogram_point> points;
    std::vector<uint16_t> codes;
};

struct SpectrogramUpdate {  // :160-168
    size_t fft_size = 0, hop_size = 0;
    float sample_rate = 0;
    size_t history_length = 0;
    bool reset = false;
    float reassigned_power_scale = 1.0f;
    std::vector<SpectrogramColumn> new_columns;
};

class SpectrogramProcessor {
public:
    explicit SpectrogramProcessor(SpectrogramConfig cfg) {  // :188-206
        cfg.normalize();
        config_ = cfg;
    }
    SpectrogramConfig config() const { return config_; }  // :208-210

    void reset_audio() {  // :212-217
        audio_.clear();
        pending_skip_ = 0;
        has_nonzero_ = false;
        reset_ = true;
    }
    void prepare() {  // :219-223
        if (!prepared_) rebuild_fft();
    }
    static size_t hilbert_len_for(size_t window_size) {  // :225-227
        size_t v = window_size * 2, p = 1;
        while (p < v) p <<= 1;
        return std::max<size_t>(p, 2);
    }

    bool process_block(const AudioBlock& block, SpectrogramUpdate& out) {  // :490-516
        if (block.is_empty()) return false;
        const float sample_rate = block.sample_rate;
        if (config_.sample_rate != sample_rate) {
            config_.sample_rate = sample_rate;
            rebuild_fft();
            audio_.clear();
            has_nonzero_ = false;
            reset_ = true;
        }
        prepare();
        push_audio(block);
        std::vector<SpectrogramColumn> cols = process_ready_windows();
        if (cols.empty()) return false;
        out.fft_size = fft_size_;
        out.hop_size = config_.hop_size;
        out.sample_rate = config_.sample_rate;
        out.history_length = config_.history_length;
        out.reset = reset_;
        reset_ = false;
        out.reassigned_power_scale = power_scale_;
        out.new_columns = std::move(cols);
        return true;
    }

    void update_config(SpectrogramConfig cfg) {  // :518-543
        cfg.normalize();
        const SpectrogramConfig prev = config_;
        const bool prepared = prepared_;
        config_ = cfg;
        const bool rate_changed = prev.sample_rate != cfg.sample_rate;
        const bool rebuild = prev.fft_size != cfg.fft_size || prev.zero_padding_factor != cfg.zero_padding_factor ||
                             prev.window != cfg.window || prev.use_reassignment != cfg.use_reassignment || rate_changed;
        if (rebuild && prepared) {
            rebuild_fft();
            if (rate_changed) {
                audio_.clear();
                has_nonzero_ = false;
            }
        }
        const bool hop_changed = prev.hop_size != cfg.hop_size;
        if (hop_changed) pending_skip_ = 0;
        reset_ = reset_ || rebuild || hop_changed;
    }

    // test-only views (reference tests reach into private fields)
    const std::deque<float>& audio_buffer() const { return audio_; }
    bool prepared() const { return prepared_; }
    size_t padded_fft_size() const { return fft_size_; }
    void push_audio_for_test(const AudioBlock& b) { push_audio(b); }
    const std::vector<float>& derivative_window() const { return derivative_window_; }
    const std::vector<float>& time_weighted_window() const { return time_weighted_window_; }
    const std::vector<float>& window() const { return window_; }
    const std::vector<float>& bin_norm() const { return bin_norm_; }

    void set_capture_inputs(bool on) {
        capture_inputs_ = on;
        captured_.clear();
    }
    const std::vector<std::vector<float>>& captured_inputs() const { return captured_; }

private:
    void rebuild_fft() {  // :229-279
        const size_t window_size = config_.fft_size;
        fft_size_ = window_size * config_.zero_padding_factor;
        const size_t hilbert_len = hilbert_len_for(window_size);
        const bool reassign = config_.use_reassignment;
        const size_t active_len = reassign ? hilbert_len : fft_size_;
        window_ = window_coefficients(config_.window, window_size);
        const size_t bin_count = fft_size_ / 2 + 1;
        real_.assign(reassign ? 0 : fft_size_, 0.0f);
        complex_.assign(reassign ? hilbert_len : bin_count, cf(0, 0));
        prepared_ = true;
        bin_norm_ = compute_fft_bin_normalization(window_, fft_size_);
        if (reassign) {
            const float inv_h = 1.0f / (float)hilbert_len;
            for (float& n : bin_norm_) n *= inv_h * inv_h;
            derivative_window_ = compute_derivative_spectral(window_);
            time_weighted_window_ = compute_time_weighted(window_);
            spectra_.assign(fft_size_ * 3, cf(0, 0));
            power_scale_ = reassigned_power_scale(window_, fft_size_);
        } else {
            derivative_window_.clear();
            time_weighted_window_.clear();
            spectra_.clear();
            power_scale_ = 1.0f;
        }
        const size_t buffered_len = active_len * 2;
        drain_audio(audio_.size() > buffered_len ? audio_.size() - buffered_len : 0);
        pending_skip_ = 0;
    }

    std::vector<SpectrogramColumn> process_ready_windows() {  // :281-388
        const size_t window_size = config_.fft_size;
        const size_t hop = config_.hop_size;
        const float sample_rate = config_.sample_rate;
        const bool reassign = config_.use_reassignment;
        const size_t bin_count = fft_size_ / 2 + 1;
        size_t read_len, center_offset;
        if (reassign) {
            const size_t h = hilbert_len_for(window_size);
            read_len = h;
            center_offset = (h - window_size) / 2;
        } else {
            read_len = window_size;
            center_offset = 0;
        }
        const size_t pending = audio_.size();
        const size_t ready = pending >= read_len ? (pending - read_len) / hop + 1 : 0;
        const uint32_t kind = reassign ? OMX_COLUMN_REASSIGNED : OMX_COLUMN_CLASSIC;
        const size_t retained = history_columns(kind, (uint32_t)bin_count, config_.history_length);
        const size_t skip = ready > retained ? ready - retained : 0;
        std::vector<SpectrogramColumn> output;
        output.reserve(std::min(ready, retained));
        advance_audio(skip * hop);
        if (capture_inputs_) captured_.clear();

        for (size_t it = skip; it < ready; ++it) {
            SpectrogramColumn col;
            col.kind = kind;
            if (capture_inputs_) captured_.emplace_back(audio_.begin(), audio_.begin() + (std::ptrdiff_t)std::min(read_len, audio_.size()));
            if (!has_nonzero_) {  // :307-316 silent fast path
                if (!reassign) col.codes.assign(bin_count, pack_classic_db(DB_FLOOR));
                output.push_back(std::move(col));
                advance_audio(hop);
                continue;
            }
            if (reassign) {  // :318-348
                for (size_t i = 0; i < complex_.size() && i < audio_.size(); ++i) complex_[i] = cf(audio_[i], 0.0f);
                hilbert_transform(complex_.data(), complex_.size());
                const cf* analytic = complex_.data() + center_offset;
                apply_complex_window(analytic, window_, spectra_.data(), fft_size_);
                apply_complex_window(analytic, derivative_window_, spectra_.data() + fft_size_, fft_size_);
                apply_complex_window(analytic, time_weighted_window_, spectra_.data() + 2 * fft_size_, fft_size_);
                fft_chunks(spectra_.data(), spectra_.size(), fft_size_, false);
                col.points = reassigned_points(sample_rate, hop, center_offset, bin_count);
            } else {  // :350-380
                copy_dc_removed_windowed(real_.data(), window_size, audio_, window_.data());
                for (size_t i = window_size; i < real_.size(); ++i) real_[i] = 0.0f;
                rfft(real_.data(), real_.size(), complex_.data());
                col.codes.resize(bin_count);
                for (size_t i = 0; i < bin_count; ++i) {
                    const cf c = complex_[i];
                    col.codes[i] = pack_classic_db(
                        power_to_db((c.real() * c.real() + c.imag() * c.imag()) * bin_norm_[i], DB_FLOOR));
                }
            }
            output.push_back(std::move(col));
            advance_audio(hop);
        }
        return output;
    }

    void drain_audio(size_t count) {  // :397-404
        count = std::min(count, audio_.size());
        if (count == 0) return;
        audio_.erase(audio_.begin(), audio_.begin() + (std::ptrdiff_t)count);
        if (has_nonzero_) {
            if (last_nonzero_ >= count) last_nonzero_ -= count;
            else has_nonzero_ = false;
        }
    }
    void advance_audio(size_t count) {  // :406-410
        const size_t missing = count > audio_.size() ? count - audio_.size() : 0;
        drain_audio(count);
        pending_skip_ += missing;
    }
    void push_audio(const AudioBlock& block) {  // :412-437
        const size_t frames = block.frame_count();
        const size_t skip = std::min(pending_skip_, frames);
        pending_skip_ -= skip;
        if (skip == frames) return;
        if (block.channels == 1) {
            const size_t base = audio_.size();
            for (size_t i = frames; i-- > skip;) {
                if (block.samples[i] != 0.0f) {
                    has_nonzero_ = true;
                    last_nonzero_ = base + (i - skip);
                    break;
                }
            }
            for (size_t i = skip; i < frames; ++i) audio_.push_back(block.samples[i]);
            return;
        }
        for (size_t f = skip; f < frames; ++f) {
            const float s = block.projected(f, OMX_CHANNEL_MID);
            if (s != 0.0f) {
                has_nonzero_ = true;
                last_nonzero_ = audio_.size();
            }
            audio_.push_back(s);
        }
    }

    std::vector<omx_spectrogram_point> reassigned_points(float sample_rate, size_t hop_size, size_t latency_samples,
                                                         size_t bin_count) const {  // :439-488
        const float bin_hz = sample_rate / (float)fft_size_;
        const float max_hz = sample_rate * 0.5f;
        const float inv_2pi = sample_rate / TAU_F;
        const float inv_hop = 1.0f / (float)hop_size;
        const float latency_hops = (float)latency_samples * inv_hop;
        std::vector<omx_spectrogram_point> points;
        const cf* spectrum = spectra_.data();
        const cf* dspec = spectra_.data() + fft_size_;
        const cf* tspec = spectra_.data() + 2 * fft_size_;
        for (size_t i = 0; i < bin_count; ++i) {
            const cf base = spectrum[i];
            const float pow = base.real() * base.real() + base.imag() * base.imag();
            const float scaled_power = pow * bin_norm_[i];
            if (scaled_power < ANALYSIS_FLOOR_POWER) continue;
            const cf d = dspec[i];
            const cf t = tspec[i];
            const float inv_pow = 1.0f / pow;
            const float d_omega = -(d.imag() * base.real() - d.real() * base.imag()) * inv_pow;
            const float freq_hz = (float)i * bin_hz + d_omega * inv_2pi;
            if (!(freq_hz > 0.0f && max_hz - freq_hz > 0.0f)) continue;
            omx_spectrogram_point p;
            p.time_offset = (t.real() * base.real() + t.imag() * base.imag()) * inv_pow * inv_hop - latency_hops;
            p.freq_hz = freq_hz;
            p.power = scaled_power;
            points.push_back(p);
        }
        return points;
    }

    // test hook (parity arbitration against exact f64, tests/parity.py): the samples every column of the last update was computed from
    // — the front `read_len` samples of the pending buffer at the moment the column was taken (:323-325, window.rs:76-79)
    bool capture_inputs_ = false;
    std::vector<std::vector<float>> captured_;

    SpectrogramConfig config_;
    bool prepared_ = false;
    size_t fft_size_ = 0;
    std::vector<float> window_, real_, derivative_window_, time_weighted_window_, bin_norm_;
    std::vector<cf> complex_, spectra_;
    float power_scale_ = 1.0f;
    std::deque<float> audio_;
    size_t pending_skip_ = 0;
    bool has_nonzero_ = false;   // audio_last_nonzero: Option<usize>
    size_t last_nonzero_ = 0;
    bool reset_ = true;
};

}  // namespace omxo
