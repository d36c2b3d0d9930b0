// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of reference src/visuals/oscilloscope/processor.rs:12-803 (NSDF period
// estimate via FFT autocorrelation, stateful template-correlation trigger, zero-crossing
// trigger, linear-interpolated trace resampling).
#pragma once
#include <complex>
#include <deque>
#include <optional>
#include <vector>

#include "fft.hpp"
#include "primitives.hpp"

namespace omxo {

constexpr float F32_EPS = std::numeric_limits<float>::epsilon();

// :14-19
inline float parabolic_refine(float y_prev, float y_curr, float y_next, size_t tau) {
    const float denom = y_prev - 2.0f * y_curr + y_next;
    if (std::fabs(denom) < F32_EPS) return (float)tau;
    const float delta = 0.5f * (y_prev - y_next) / denom;
    return rmax((float)tau + rclamp(delta, -1.0f, 1.0f), 1.0f);
}

struct OscilloscopeConfig {  // :33-43
    float sample_rate = DEFAULT_SAMPLE_RATE;
    float segment_duration = 0.02f;
    uint32_t trigger_mode = OMX_TRIGGER_STABLE;  // TriggerMode::default() = Stable{2} (:27-31)
    size_t num_cycles = 2;
    uint32_t trigger_source = OMX_CHANNEL_MID;
    uint32_t channel_1 = OMX_CHANNEL_MID;
    uint32_t channel_2 = OMX_CHANNEL_NONE;
    bool operator==(const OscilloscopeConfig& o) const {
        const bool mode_eq = trigger_mode == o.trigger_mode &&
                             (trigger_mode != OMX_TRIGGER_STABLE || num_cycles == o.num_cycles);
        return sample_rate == o.sample_rate && segment_duration == o.segment_duration && mode_eq &&
               trigger_source == o.trigger_source && channel_1 == o.channel_1 && channel_2 == o.channel_2;
    }
};

struct PeriodEstimate {
    float period;
    float confidence;
};

// Rust `usize::next_power_of_two`
inline size_t next_pow2(size_t v) {
    size_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct PeriodEstimator {  // :77-182
    static constexpr float MIN_HZ = 20.0f;
    static constexpr float MAX_HZ = 8000.0f;
    static constexpr float PROBE_SECONDS = 0.1f;
    static constexpr float MIN_SIGNAL_PEAK = 0.001f;
    static constexpr float MIN_PERIODICITY = 0.5f;
    static constexpr float PEAK_CUTOFF = 0.93f;

    std::vector<float> periodicity, energy_prefix;
    float last_peak = 0.0f;
    std::vector<float> fft_input;
    std::vector<std::complex<float>> fft_spectrum;

    std::optional<PeriodEstimate> estimate_period(const float* samples, size_t n, float rate) {  // :93-131
        last_peak = 0.0f;
        if (n < 3) return std::nullopt;
        const float mean = rust_sum_f32(samples, n) / (float)n;
        float peak = 0.0f;
        for (size_t i = 0; i < n; ++i) peak = rmax(peak, std::fabs(samples[i] - mean));
        last_peak = peak;
        if (last_peak < MIN_SIGNAL_PEAK) return std::nullopt;
        const size_t min_period = f2usize((double)rmax(std::round(rate / MAX_HZ), 2.0f));
        const size_t max_period = std::min(f2usize((double)std::round(rate / MIN_HZ)), n / 2);
        if (max_period <= min_period + 1) return std::nullopt;
        if (!compute_periodicity(samples, n, mean, max_period)) return std::nullopt;

        const float* nsdf = periodicity.data();
        size_t zero_crossing = 0;
        for (size_t tau = 1; tau <= max_period; ++tau)
            if (nsdf[tau] <= 0.0f) { zero_crossing = tau; break; }
        if (zero_crossing == 0) return std::nullopt;
        const size_t first_tau = std::max(min_period, zero_crossing);
        if (first_tau >= max_period) return std::nullopt;
        auto is_candidate = [&](size_t tau) {
            return nsdf[tau] >= MIN_PERIODICITY && nsdf[tau] >= nsdf[tau - 1] && nsdf[tau] >= nsdf[tau + 1];
        };
        // Iterator::max_by returns the LAST maximum under total_cmp.
        bool have_best = false;
        size_t best = 0;
        for (size_t tau = first_tau; tau < max_period; ++tau) {
            if (!is_candidate(tau)) continue;
            if (!have_best || total_cmp(nsdf[tau], nsdf[best]) >= 0) { best = tau; have_best = true; }
        }
        if (!have_best) return std::nullopt;
        const float cutoff = nsdf[best] * PEAK_CUTOFF;
        size_t pk = best;
        for (size_t tau = first_tau; tau <= best; ++tau)
            if (is_candidate(tau) && nsdf[tau] >= cutoff) { pk = tau; break; }
        PeriodEstimate e;
        e.period = parabolic_refine(nsdf[pk - 1], nsdf[pk], nsdf[pk + 1], pk);
        e.confidence = rclamp(nsdf[pk], 0.0f, 1.0f);
        return e;
    }

    static int total_cmp(float a, float b) {  // f32::total_cmp
        int32_t x, y;
        std::memcpy(&x, &a, 4);
        std::memcpy(&y, &b, 4);
        x ^= (int32_t)((uint32_t)(x >> 31) >> 1);
        y ^= (int32_t)((uint32_t)(y >> 31) >> 1);
        return (x > y) - (x < y);
    }

    bool compute_periodicity(const float* samples, size_t n, float mean, size_t max_lag) {  // :133-181
        const size_t fft_size = next_pow2(n + max_lag);
        if (fft_input.size() != fft_size) {
            fft_input.assign(fft_size, 0.0f);
            fft_spectrum.assign(fft_size / 2 + 1, std::complex<float>(0, 0));
        }
        energy_prefix.resize(n + 1, 0.0f);
        energy_prefix[0] = 0.0f;
        for (size_t i = 0; i < n; ++i) {
            const float centered = samples[i] - mean;
            fft_input[i] = centered;
            energy_prefix[i + 1] = centered * centered + energy_prefix[i];
        }
        for (size_t i = n; i < fft_size; ++i) fft_input[i] = 0.0f;
        rfft(fft_input.data(), fft_size, fft_spectrum.data());
        for (auto& bin : fft_spectrum)
            bin = std::complex<float>(bin.real() * bin.real() + bin.imag() * bin.imag(), 0.0f);
        if (!irfft(fft_spectrum.data(), fft_size, fft_input.data())) return false;
        const float norm = 1.0f / (float)fft_size;
        periodicity.resize(max_lag + 1, 0.0f);
        const float total_energy = energy_prefix[n];
        if (total_energy <= F32_EPS) return false;
        for (size_t tau = 0; tau <= max_lag; ++tau) {
            const float left_energy = energy_prefix[n - tau];
            const float right_energy = total_energy - energy_prefix[tau];
            const float denom = left_energy + right_energy;
            periodicity[tau] = denom > F32_EPS ? 2.0f * fft_input[tau] * norm / denom : 0.0f;
        }
        return true;
    }
};

struct Capture {  // :265-270
    float span;
    size_t start;
    float frac_offset;
};

namespace trig {
constexpr float WINDOW_SECONDS = 0.04f;
constexpr float MIN_CYCLES = 2.0f;
constexpr float SEARCH_PERIODS = 1.5f;
constexpr float NORMALIZE_FLOOR = 0.01f;
constexpr float MEAN_RESPONSIVENESS = 0.25f;
constexpr float EDGE_STRENGTH = 1.0f;
constexpr float BUFFER_RESPONSIVENESS = 0.5f;
constexpr float BUFFER_FALLOFF_PERIODS = 0.5f;
constexpr float BUFFER_RETUNE_SEMITONES = 1.0f;
constexpr float SLOPE_WIDTH_PERIODS = 0.25f;
constexpr float RESET_BELOW_MATCH = 0.3f;
constexpr uint8_t MAX_MISSED_PERIODS = 4;
}  // namespace trig

// :184-189
inline size_t trigger_kernel_len(float period, float rate) {
    return f2usize((double)rmax(std::round(rmax(rate * trig::WINDOW_SECONDS, period * trig::MIN_CYCLES)), 2.0f));
}
// :191-197
inline void normalize_peak(std::vector<float>& data) {
    float peak = 0.0f;
    for (float s : data) peak = rmax(peak, std::fabs(s));
    const float scale = 1.0f / rmax(peak, trig::NORMALIZE_FLOOR);
    for (float& s : data) s *= scale;
}
// :199-204
inline float gaussian(size_t len, size_t index, float std_) {
    if (len <= 1 || std_ <= F32_EPS) return 0.0f;
    const float center = (float)(len - 1) * 0.5f;
    const float x = (float)index - center;
    const float r = x / std_;
    return std::exp(-0.5f * (r * r));  // powi(2) == r*r
}
// :206-208
inline void correlation_stats(const std::vector<float>& y, float out[2]) {
    float sum = 0.0f, squares = 0.0f;
    for (float v : y) {
        sum = sum + v;
        squares = squares + v * v;
    }
    out[0] = sum;
    out[1] = squares;
}
// :210-236 — 4-lane accumulation order preserved
inline float normalized_correlation(const float* x, const float* y, size_t n, const float stats[2]) {
    float sums[3][4] = {};
    const size_t chunks = n / 4;
    for (size_t c = 0; c < chunks; ++c) {
        for (int lane = 0; lane < 4; ++lane) {
            const float xv = x[c * 4 + lane], yv = y[c * 4 + lane];
            sums[0][lane] += xv;
            sums[1][lane] += xv * xv;
            sums[2][lane] += xv * yv;
        }
    }
    float sum_x = rust_sum_f32(sums[0], 4), sum_xx = rust_sum_f32(sums[1], 4), sum_xy = rust_sum_f32(sums[2], 4);
    for (size_t i = chunks * 4; i < n; ++i) {
        sum_x += x[i];
        sum_xx += x[i] * x[i];
        sum_xy += x[i] * y[i];
    }
    if (n == 0) return 0.0f;
    const float nf = (float)n;
    const float dot = sum_xy - sum_x * stats[0] / nf;
    const float energy_x = rmax(sum_xx - sum_x * sum_x / nf, 0.0f);
    const float energy_y = rmax(stats[1] - stats[0] * stats[0] / nf, 0.0f);
    const float denom = std::sqrt(energy_x * energy_y);
    return denom > F32_EPS ? rclamp(dot / denom, -1.0f, 1.0f) : 0.0f;
}
// :238-247 (+ util.rs:18-20 lerp)
inline float sample_linear_zero(const float* data, size_t n, float pos) {
    if (n == 0 || pos < 0.0f || pos > (float)(n - 1)) return 0.0f;
    const size_t idx = f2usize((double)pos);
    const float frac = pos - (float)idx;
    if (frac > F32_EPS && idx + 1 < n) return data[idx] + (data[idx + 1] - data[idx]) * frac;
    return data[idx];
}
// :249-263
inline std::vector<float> retune_reference_fn(const std::vector<float>& reference, float old_period, float new_period,
                                              size_t len) {
    const float ratio = new_period / old_period;
    if (!std::isfinite(ratio) || ratio <= F32_EPS) return std::vector<float>(len, 0.0f);
    const float old_center = (float)(reference.empty() ? 0 : reference.size() - 1) * 0.5f;
    const float new_center = (float)(len == 0 ? 0 : len - 1) * 0.5f;
    std::vector<float> out(len);
    for (size_t i = 0; i < len; ++i) {
        const float pos = old_center + ((float)i - new_center) / ratio;
        out[i] = sample_linear_zero(reference.data(), reference.size(), pos);
    }
    return out;
}

struct StableTrigger {  // :272-528
    PeriodEstimator estimator;
    std::optional<float> period;
    uint8_t missed_periods = 0;
    std::vector<float> reference;
    float reference_period = 0.0f;
    std::vector<float> work, candidate;
    float mean = 0.0f;

    void unlock() {  // :298-304
        period.reset();
        missed_periods = 0;
        reference.clear();
        reference_period = 0.0f;
        mean = 0.0f;
    }

    Capture capture(const float* trace, size_t n, float sample_rate, size_t probe_frames, size_t fallback_frames,
                    size_t cycles) {  // :306-334
        const size_t probe_len = std::min(probe_frames, n);
        std::optional<PeriodEstimate> detected;
        if (probe_len >= 3) detected = estimator.estimate_period(trace + (n - probe_len), probe_len, sample_rate);
        else estimator.last_peak = 0.0f;
        if (probe_len > 0 && estimator.last_peak < PeriodEstimator::MIN_SIGNAL_PEAK) unlock();
        std::optional<PeriodEstimate> est = stabilize(detected);
        if (est) {
            std::optional<Capture> c = locate(trace, n, *est, cycles, sample_rate);
            if (c) return *c;
        }
        Capture c;
        c.span = (float)std::max<size_t>(fallback_frames > 0 ? fallback_frames - 1 : 0, 1);
        c.start = n > fallback_frames ? n - fallback_frames : 0;
        c.frac_offset = 0.0f;
        return c;
    }

    std::optional<PeriodEstimate> stabilize(std::optional<PeriodEstimate> detected) {  // :336-356
        if (!detected) {
            if (!period) return std::nullopt;
            const float p = *period;
            missed_periods = missed_periods == 255 ? 255 : (uint8_t)(missed_periods + 1);
            if (missed_periods > trig::MAX_MISSED_PERIODS) {
                unlock();
                return std::nullopt;
            }
            return PeriodEstimate{p, 0.0f};
        }
        PeriodEstimate estimate = *detected;
        missed_periods = 0;
        if (period) {
            const float prev = *period;
            const float r = estimate.period / prev;
            if (r >= 0.9f && r <= 1.1f) estimate.period = prev + 0.35f * (estimate.period - prev);
        }
        period = estimate.period;
        return estimate;
    }

    std::optional<Capture> locate(const float* trace, size_t n, PeriodEstimate estimate, size_t cycles, float rate) {  // :358-411
        const float per = rmax(estimate.period, 1.0f);
        const float span = per * (float)std::max<size_t>(cycles, 1);
        const size_t frames = f2usize((double)std::ceil(span)) + 1;
        const size_t len = trigger_kernel_len(per, rate);
        const size_t before = len / 2;
        const size_t after = len - before;
        const size_t tail = std::max(frames, after);
        if (n < tail) return std::nullopt;
        const size_t right = n - tail;
        if (right < before) return std::nullopt;
        size_t search = std::max<size_t>(f2usize((double)std::round(per * trig::SEARCH_PERIODS)), 1);
        search = std::min(search, len / 2);
        search = std::min(search, right - before);
        const size_t left = right - search;
        prepare(trace + (left - before), (right + after) - (left - before), len, per);

        bool use_reference = false;
        for (float s : reference)
            if (std::fabs(s) > 1.0e-3f) { use_reference = true; break; }
        prepare_template(per, use_reference);
        auto best = find_best(search, per);
        size_t offset = best.first;
        float frac_offset = best.second;
        const bool confident = estimate.confidence >= PeriodEstimator::MIN_PERIODICITY;
        auto segment = [&](size_t off) { return trace + (left + off - before); };
        const bool reset = confident && use_reference && write_candidate(segment(offset), len, per) < trig::RESET_BELOW_MATCH;
        if (reset) {
            std::fill(reference.begin(), reference.end(), 0.0f);
            prepare_template(per, false);
            best = find_best(search, per);
            offset = best.first;
            frac_offset = best.second;
        }
        if (confident) {
            if (!use_reference || reset) write_candidate(segment(offset), len, per);
            update_reference(per);
        }
        size_t start = left + offset;
        if (frac_offset < 0.0f && start > 0) {
            start -= 1;
            frac_offset += 1.0f;
        }
        return Capture{span, start, frac_offset};
    }

    void prepare(const float* data, size_t n, size_t len, float per) {  // :413-420
        retune_reference(len, per);
        const float m = rust_sum_f32(data, n) / (float)std::max<size_t>(n, 1);
        mean += trig::MEAN_RESPONSIVENESS * (m - mean);
        work.resize(n);
        for (size_t i = 0; i < n; ++i) work[i] = data[i] - mean;
    }

    void prepare_template(float per, bool use_reference) {  // :422-439
        const size_t len = reference.size();
        candidate.resize(len, 0.0f);
        const size_t midpoint = len / 2;
        const float max_width = rmax((float)std::max<size_t>(midpoint, 1) / 3.0f, 1.0f);
        const float width = rclamp(trig::SLOPE_WIDTH_PERIODS * per, 1.0f, max_width);
        for (size_t i = 0; i < (len + 1) / 2; ++i) {
            const size_t mirror = len - 1 - i;
            const float weight = gaussian(len, i, width);
            candidate[i] = -0.5f * trig::EDGE_STRENGTH * 2.0f * weight;
            candidate[mirror] = 0.5f * trig::EDGE_STRENGTH * 2.0f * weight;
        }
        if (use_reference)
            for (size_t i = 0; i < std::min(candidate.size(), reference.size()); ++i) candidate[i] += reference[i];
    }

    std::pair<size_t, float> find_best(size_t search, float per) {  // :441-484
        const std::vector<float>& tmpl = candidate;
        float stats[2];
        correlation_stats(tmpl, stats);
        std::vector<float>& scores = estimator.periodicity;
        const float NEG_INF = -std::numeric_limits<float>::infinity();
        scores.assign(search + 1, NEG_INF);
        auto score_at = [&](size_t offset) {
            if (scores[offset] == NEG_INF)
                scores[offset] = normalized_correlation(work.data() + offset, tmpl.data(), tmpl.size(), stats);
            return scores[offset];
        };
        size_t stride = f2usize((double)std::round(per / 16.0f));
        stride = std::min(std::max<size_t>(stride, 1), (size_t)128);
        stride = std::min(stride, std::max<size_t>(search, 1));
        size_t best_off = search / 2;
        float best_score = NEG_INF;
        // (0..=search).rev().step_by(stride).chain([0])
        for (size_t k = 0;; ++k) {
            if (k * stride > search) break;
            const size_t offset = search - k * stride;
            const float score = score_at(offset);
            if (score > best_score) { best_off = offset; best_score = score; }
        }
        {
            const float score = score_at(0);
            if (score > best_score) { best_off = 0; best_score = score; }
        }
        size_t step = stride;
        while (step > 1) {
            const size_t next = std::max<size_t>(step / 4, 1);
            const size_t lo = best_off > step ? best_off - step : 0;
            const size_t hi = std::min(best_off + step, search);
            // (lo..=hi).rev().step_by(next); the range is fixed before iteration
            for (size_t k = 0;; ++k) {
                if (lo + k * next > hi) break;
                const size_t offset = hi - k * next;
                const float score = score_at(offset);
                if (score > best_score) { best_off = offset; best_score = score; }
            }
            step = next;
        }
        float frac_offset = 0.0f;
        if (best_off > 0 && best_off < search) {
            const float prev = score_at(best_off - 1), next = score_at(best_off + 1);
            frac_offset = rclamp(parabolic_refine(prev, best_score, next, best_off) - (float)best_off, -0.5f, 0.5f);
        }
        return {best_off, frac_offset};
    }

    void retune_reference(size_t len, float per) {  // :486-498
        if (reference.empty()) {
            reference.assign(len, 0.0f);
            reference_period = per;
            return;
        }
        const float semitones = std::log2(per / reference_period) * 12.0f;
        if (reference.size() != len || std::fabs(semitones) >= trig::BUFFER_RETUNE_SEMITONES) {
            reference = retune_reference_fn(reference, reference_period, per, len);
            reference_period = per;
        }
    }

    void update_reference(float per) {  // :500-507
        normalize_peak(reference);
        for (size_t i = 0; i < std::min(reference.size(), candidate.size()); ++i)
            reference[i] += trig::BUFFER_RESPONSIVENESS * (candidate[i] - reference[i]);
        reference_period += trig::BUFFER_RESPONSIVENESS * (per - reference_period);
    }

    float write_candidate(const float* segment, size_t n, float per) {  // :509-527
        const float m = rust_sum_f32(segment, n) / (float)std::max<size_t>(n, 1);
        candidate.resize(n);
        for (size_t i = 0; i < n; ++i) candidate[i] = segment[i] - m;
        normalize_peak(candidate);
        const float std_ = rmax(per * trig::BUFFER_FALLOFF_PERIODS, 1.0f);
        const size_t len = candidate.size();
        for (size_t i = 0; i < (len + 1) / 2; ++i) {
            const size_t mirror = len - 1 - i;
            const float weight = gaussian(len, i, std_);
            candidate[i] *= weight;
            if (mirror != i) candidate[mirror] *= weight;
        }
        float stats[2];
        correlation_stats(candidate, stats);
        // zip(reference, candidate): debug_assert equal lengths; use the shorter
        const size_t m_len = std::min(reference.size(), candidate.size());
        return normalized_correlation(reference.data(), candidate.data(), m_len, stats);
    }
};

// :530-551 — `frames` given as (first, last, ascending?) inclusive
inline std::optional<size_t> find_rising_zero_crossing(const float* samples, size_t n, size_t lo, size_t hi, bool reversed) {
    if (lo > hi) return std::nullopt;
    const size_t count = hi - lo + 1;
    auto at = [&](size_t k) { return reversed ? hi - k : lo + k; };
    const size_t first = at(0);
    if (first >= n) return std::nullopt;
    float prev_val = samples[first];
    size_t prev_idx = first;
    for (size_t k = 1; k < count; ++k) {
        const size_t f = at(k);
        if (f >= n) return std::nullopt;
        const float cur = samples[f];
        float lo_val, hi_val;
        size_t hi_idx;
        if (f > prev_idx) { lo_val = prev_val; hi_idx = f; hi_val = cur; }
        else { lo_val = cur; hi_idx = prev_idx; hi_val = prev_val; }
        if (hi_val > 0.0f && lo_val <= 0.0f) return hi_idx;
        prev_val = cur;
        prev_idx = f;
    }
    return std::nullopt;
}

// :761-767
inline size_t stable_history_frames(size_t max_period, size_t cycles, float sample_rate) {
    const float max_period_f = (float)max_period;
    const size_t max_kernel = trigger_kernel_len(max_period_f, sample_rate);
    const size_t max_tail = std::max(max_period * std::max<size_t>(cycles, 1) + 1, (max_kernel + 1) / 2);
    const size_t max_search = f2usize((double)std::ceil(max_period_f * trig::SEARCH_PERIODS));
    return max_kernel / 2 + max_tail + max_search + 2;
}

// :769-786
inline std::optional<Capture> zero_crossing_capture(const float* samples, size_t n, size_t frames, size_t search_range) {
    frames = std::min(frames, n);
    if (frames == 0) return std::nullopt;
    const size_t end = n > 0 ? n - 1 : 0;
    const size_t right_lo = end > search_range ? end - search_range : 0;
    const size_t right = find_rising_zero_crossing(samples, n, right_lo, end, true).value_or(end);
    const size_t left_lo = right > frames ? right - frames : 0;
    const size_t left_hi = std::min(left_lo + search_range, right > 2 ? right - 2 : 0);
    const size_t left = find_rising_zero_crossing(samples, n, left_lo, left_hi, false).value_or(left_lo);
    Capture c;
    c.span = (float)std::max<size_t>(right > left ? right - left : 0, 1);
    c.start = left;
    c.frac_offset = 0.0f;
    return c;
}

// :788-803
inline bool downsample_trace(std::vector<float>& output, const float* data, size_t n, Capture capture, size_t target) {
    if (target < 2) return false;
    const size_t start = std::min(capture.start, n);
    data += start;
    n -= start;
    if (n < 2) return false;
    const float last = (float)(n - 1);
    const float start_offset = rclamp(capture.frac_offset, 0.0f, last);
    const float span = rmin(capture.span, last - start_offset);
    if (!(std::isfinite(span) && span > 0.0f)) return false;
    const float step = span / (float)(target - 1);
    for (size_t i = 0; i < target; ++i) output.push_back(sample_linear_zero(data, n, start_offset + (float)i * step));
    return true;
}

struct OscilloscopeSnapshot {  // :553-560
    uint64_t epoch = 0;
    size_t channels = 0;
    size_t slots[2] = {0, 0};
    std::vector<float> samples;
    size_t samples_per_channel = 0;
};

struct TraceState {  // :564-568
    std::vector<float> buffer;  // VecDeque<f32>, always used through make_contiguous()
    StableTrigger trigger;
};

class OscilloscopeProcessor {
public:
    explicit OscilloscopeProcessor(OscilloscopeConfig cfg) : config_(cfg) {}  // :579-587
    OscilloscopeConfig config() const { return config_; }

    void reset_audio() {  // :593-600
        clear_history();
        const uint64_t epoch = snapshot_.epoch;
        snapshot_ = OscilloscopeSnapshot();
        snapshot_.epoch = epoch;
    }

    const std::optional<Capture>& last_capture() const { return last_capture_; }
    std::optional<float> last_cycle_rate() const {  // :602-609
        std::optional<float> p = source_.trigger.period;
        if (!p)
            for (const auto& t : traces_)
                if (t.trigger.period) { p = t.trigger.period; break; }
        if (!p) return std::nullopt;
        return config_.sample_rate / *p;
    }

    bool process_block(const AudioBlock& block, OscilloscopeSnapshot& out) {  // :611-712
        if (block.is_empty()) return false;
        if (config_.sample_rate != block.sample_rate) {
            OscilloscopeConfig c = config_;
            c.sample_rate = block.sample_rate;
            update_config(c);
        }
        const size_t channel_count = block.channels;
        if (has_history_channels_ && history_channels_ != channel_count) clear_history();
        has_history_channels_ = true;
        history_channels_ = channel_count;

        const size_t base_frames = f2usize((double)rmax(std::round(config_.sample_rate * config_.segment_duration), 1.0f));
        const size_t max_period = f2usize((double)std::ceil(config_.sample_rate / PeriodEstimator::MIN_HZ));
        const size_t probe_frames =
            std::max(f2usize((double)std::round(config_.sample_rate * PeriodEstimator::PROBE_SECONDS)), max_period * 2);
        const size_t trigger_frames = config_.trigger_mode == OMX_TRIGGER_ZERO_CROSSING
                                          ? base_frames + max_period
                                          : stable_history_frames(max_period, config_.num_cycles, config_.sample_rate);
        const uint32_t trace_channels[2] = {config_.channel_1, config_.channel_2};
        const uint32_t trigger_source = config_.trigger_source;
        const size_t history_frames = std::max(std::max(probe_frames, base_frames), trigger_frames);
        const float sample_rate = config_.sample_rate;
        auto capture = [&](const std::vector<float>& trace, StableTrigger& trigger) -> std::optional<Capture> {
            if (config_.trigger_mode == OMX_TRIGGER_ZERO_CROSSING)
                return zero_crossing_capture(trace.data(), trace.size(), base_frames, max_period);
            if (trace.size() >= base_frames)
                return trigger.capture(trace.data(), trace.size(), sample_rate, probe_frames, base_frames, config_.num_cycles);
            return std::nullopt;
        };
        const bool active[2] = {trace_channels[0] != OMX_CHANNEL_NONE, trace_channels[1] != OMX_CHANNEL_NONE};
        int matching_trace = -1;
        for (int s = 0; s < 2; ++s)
            if (trace_channels[s] == trigger_source) { matching_trace = s; break; }
        if (matching_trace >= 0 && !active[matching_trace]) matching_trace = -1;
        const bool separate_source = matching_trace < 0 && trigger_source != OMX_CHANNEL_NONE;
        if (trigger_source == OMX_CHANNEL_NONE) source_.buffer.clear();

        if (active[0] || active[1] || separate_source) {
            const size_t nframes = block.frame_count();
            for (size_t f = 0; f < nframes; ++f) {
                float lr[2];
                block.stereo_frame(f, lr);
                for (int s = 0; s < 2; ++s)
                    if (trace_channels[s] != OMX_CHANNEL_NONE) traces_[s].buffer.push_back(project(trace_channels[s], lr[0], lr[1]));
                if (separate_source) source_.buffer.push_back(project(trigger_source, lr[0], lr[1]));
            }
        }
        auto trim = [](std::vector<float>& b, size_t keep) {
            if (b.size() > keep) b.erase(b.begin(), b.begin() + (std::ptrdiff_t)(b.size() - keep));
        };
        for (int s = 0; s < 2; ++s) trim(traces_[s].buffer, active[s] ? history_frames : 0);
        if (separate_source) trim(source_.buffer, history_frames);

        std::optional<Capture> linked;
        if (matching_trace >= 0) linked = capture(traces_[matching_trace].buffer, source_.trigger);
        else if (separate_source) linked = capture(source_.buffer, source_.trigger);

        std::optional<Capture> captures[2];
        for (int s = 0; s < 2; ++s) {
            if (!active[s]) continue;
            captures[s] = linked ? linked : capture(traces_[s].buffer, traces_[s].trigger);
        }
        if (!captures[0] && !captures[1]) return false;
        last_capture_ = captures[0] ? captures[0] : captures[1];  // test-only view (omx_oscilloscope_last_capture)
        write_snapshot(captures);
        out = snapshot_;
        return true;
    }

    void update_config(OscilloscopeConfig cfg) {  // :752-758
        if (!(config_ == cfg)) {
            const uint64_t epoch = snapshot_.epoch + 1;
            *this = OscilloscopeProcessor(cfg);
            snapshot_.epoch = epoch;
        }
    }

    // test-only views
    const std::vector<float>& trace_buffer(int slot) const { return traces_[slot].buffer; }

private:
    void clear_history() {  // :714-723
        snapshot_.epoch += 1;
        has_history_channels_ = false;
        for (auto& t : traces_) {
            t.buffer.clear();
            t.trigger.unlock();
        }
        source_.buffer.clear();
        source_.trigger.unlock();
    }

    void write_snapshot(const std::optional<Capture> captures[2]) {  // :725-750
        const size_t TARGET = 4096;
        size_t target = 0;
        bool any = false;
        for (int s = 0; s < 2; ++s) {
            if (!captures[s]) continue;
            const size_t t = f2usize((double)rmax(std::round(captures[s]->span), 1.0f)) + 1;
            target = any ? std::max(target, t) : t;
            any = true;
        }
        if (!any) target = 2;
        target = std::min(std::max<size_t>(target, 2), TARGET);
        snapshot_.samples.clear();
        snapshot_.channels = 0;
        for (int slot = 0; slot < 2; ++slot) {
            if (!captures[slot]) continue;
            if (downsample_trace(snapshot_.samples, traces_[slot].buffer.data(), traces_[slot].buffer.size(), *captures[slot],
                                 target)) {
                snapshot_.slots[snapshot_.channels] = (size_t)slot;
                snapshot_.channels += 1;
            }
        }
        snapshot_.samples_per_channel = snapshot_.channels == 0 ? 0 : target;
    }

    OscilloscopeConfig config_;
    OscilloscopeSnapshot snapshot_;
    std::optional<Capture> last_capture_;
    bool has_history_channels_ = false;
    size_t history_channels_ = 0;
    TraceState traces_[2];
    TraceState source_;
};

}  // namespace omxo
