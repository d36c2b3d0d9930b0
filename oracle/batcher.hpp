// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of reference src/meter.rs:15-80 (DspBatcher) and :145-166 (ingest_silence).
#pragma once
#include <cmath>
#include <cstring>
#include <functional>
#include <optional>
#include <vector>

#include "primitives.hpp"

namespace omxo {

struct AudioFormat {  // dsp.rs:79-85
    size_t channels = 2;
    float sample_rate = DEFAULT_SAMPLE_RATE;
    uint64_t generation = 0;
    Positions positions{};
    bool operator==(const AudioFormat& o) const {
        return channels == o.channels && sample_rate == o.sample_rate && generation == o.generation && positions == o.positions;
    }
};

class DspBatcher {
public:
    using Ingest = std::function<void(const float*, size_t, const AudioFormat&)>;
    static constexpr size_t BATCH_FRAMES_AT_48K = 256, MAX_INGEST_FRAMES_AT_48K = 1024, SILENCE_CHUNK_FRAMES = 4096;
    static constexpr uint64_t MAX_SILENCE_SECONDS = 2;

    static size_t scaled_samples(size_t frames_at_48k, const AudioFormat& f) {  // :20-25
        const double scaled = std::round((double)frames_at_48k * (double)f.sample_rate / (double)DEFAULT_SAMPLE_RATE);
        return f2usize(std::fmax(scaled, 1.0)) * std::max<size_t>(f.channels, 1);
    }

    size_t push(const float* data, size_t n, const AudioFormat& format, const Ingest& ingest) {  // :40-69
        if (format_ && !(*format_ == format)) samples_.clear();
        format_ = format;
        const size_t batch = scaled_samples(BATCH_FRAMES_AT_48K, format);
        size_t count = 0;
        if (!samples_.empty()) {
            const size_t take = std::min(batch - samples_.size(), n);
            samples_.insert(samples_.end(), data, data + take);
            data += take;
            n -= take;
            if (samples_.size() == batch) {
                ingest(samples_.data(), samples_.size(), format);
                samples_.clear();
                count += 1;
            }
        }
        const size_t ready = n / batch * batch;
        const size_t step = scaled_samples(MAX_INGEST_FRAMES_AT_48K, format);
        for (size_t at = 0; at < ready; at += step) {
            ingest(data + at, std::min(step, ready - at), format);
            count += 1;
        }
        samples_.insert(samples_.end(), data + ready, data + n);
        return count;
    }
    void clear() {  // :76-79
        samples_.clear();
        format_.reset();
    }
    // ingest_silence (:145-166); returns false when the gap was too long and the caller must reset the visuals
    bool push_silence(uint64_t frames, const AudioFormat& format, const Ingest& ingest, size_t* count_out) {
        const double lim = std::fmax(std::round((double)MAX_SILENCE_SECONDS * (double)format.sample_rate), 1.0);
        if ((double)frames > lim) {
            clear();
            if (count_out) *count_out = 0;
            return false;
        }
        std::vector<float> scratch(SILENCE_CHUNK_FRAMES * MAX_CH, 0.0f);
        const size_t capacity = scratch.size() / std::max<size_t>(format.channels, 1);
        size_t count = 0;
        uint64_t remaining = frames;
        while (remaining > 0) {
            const size_t chunk = (size_t)std::min<uint64_t>(remaining, capacity);
            count += push(scratch.data(), chunk * format.channels, format, ingest);
            remaining -= chunk;
        }
        if (count_out) *count_out = count;
        return true;
    }
    const std::vector<float>& pending() const { return samples_; }
    const std::optional<AudioFormat>& format() const { return format_; }

private:
    std::vector<float> samples_;
    std::optional<AudioFormat> format_;
};

}  // namespace omxo
