// DspBatcher (reference src/meter.rs:15-80) and ingest_silence (:145-166): the block-partition policy of the
// processors' caller, as host-side integer logic behind the C ABI.
#include <vector>

#include "common.hpp"

namespace {
constexpr size_t SILENCE_CHUNK_FRAMES = 4096;          // :15
constexpr size_t DSP_BATCH_FRAMES_AT_48K = 256;        // :16
constexpr size_t MAX_DSP_INGEST_FRAMES_AT_48K = 1024;  // :17
constexpr uint64_t MAX_SILENCE_SECONDS = 2;            // :18

size_t scaled_samples(size_t frames_at_48k, const omx_audio_format& f) {  // :20-25
    const double v = std::round((double)frames_at_48k * (double)f.sample_rate / (double)omx::kDefaultSampleRate);
    return omx::f2usize(std::fmax(v, 1.0)) * std::max<size_t>(f.channels, 1);
}
bool same_format(const omx_audio_format& a, const omx_audio_format& b) {  // derive(PartialEq) on AudioFormat
    return a.generation == b.generation && a.sample_rate == b.sample_rate && a.channels == b.channels &&
           std::memcmp(a.positions, b.positions, sizeof(a.positions)) == 0;
}
}  // namespace

struct omx_batcher {
    std::vector<float> samples;
    bool has_format = false;
    omx_audio_format format{};
    std::vector<float> silence;  // MeterEngine::silence scratch (:96)
};

extern "C" {

int omx_batcher_create(omx_batcher** out) {
    if (!out) return OMX_ERR_INVALID;
    *out = new omx_batcher();
    (*out)->samples.reserve(DSP_BATCH_FRAMES_AT_48K * OMX_MAX_CHANNELS);
    return OMX_NONE;
}
void omx_batcher_destroy(omx_batcher* b) { delete b; }

uint64_t omx_batcher_push(omx_batcher* b, const float* samples, uint64_t n, const omx_audio_format* format, omx_ingest_fn ingest,
                          void* user) {
    if (!b || !format || (!samples && n)) return 0;
    if (b->has_format && !same_format(b->format, *format)) b->samples.clear();
    b->has_format = true;
    b->format = *format;
    const size_t batch = scaled_samples(DSP_BATCH_FRAMES_AT_48K, *format);
    uint64_t count = 0;
    if (!b->samples.empty()) {
        const size_t take = std::min<uint64_t>(batch - b->samples.size(), n);
        b->samples.insert(b->samples.end(), samples, samples + take);
        samples += take;
        n -= take;
        if (b->samples.size() == batch) {
            if (ingest) ingest(user, b->samples.data(), b->samples.size(), format);
            b->samples.clear();
            ++count;
        }
    }
    const uint64_t ready = n / batch * batch;
    const size_t chunk = scaled_samples(MAX_DSP_INGEST_FRAMES_AT_48K, *format);
    for (uint64_t off = 0; off < ready; off += chunk) {
        const uint64_t len = std::min<uint64_t>(chunk, ready - off);
        if (ingest) ingest(user, samples + off, len, format);
        ++count;
    }
    b->samples.insert(b->samples.end(), samples + ready, samples + n);
    return count;
}

void omx_batcher_clear(omx_batcher* b) {
    if (!b) return;
    b->samples.clear();
    b->has_format = false;
}
void omx_batcher_reset(omx_batcher* b, omx_reset_fn reset, void* user) {
    omx_batcher_clear(b);
    if (reset) reset(user);
}

uint64_t omx_batcher_push_silence(omx_batcher* b, uint64_t frames, const omx_audio_format* format, omx_ingest_fn ingest,
                                  omx_reset_fn reset, void* user) {
    if (!b || !format) return 0;
    const double lim = std::fmax(std::round((double)MAX_SILENCE_SECONDS * (double)format->sample_rate), 1.0);
    const uint64_t limit = lim >= 18446744073709551615.0 ? UINT64_MAX : (uint64_t)lim;
    if (frames > limit) {
        omx_batcher_reset(b, reset, user);
        return 0;
    }
    if (b->silence.empty()) b->silence.assign(SILENCE_CHUNK_FRAMES * OMX_MAX_CHANNELS, 0.0f);
    const size_t channels = std::max<size_t>(format->channels, 1);
    const size_t capacity = b->silence.size() / channels;
    uint64_t remaining = frames, count = 0;
    while (remaining > 0) {
        const size_t chunk = (size_t)std::min<uint64_t>(remaining, capacity);
        count += omx_batcher_push(b, b->silence.data(), (uint64_t)chunk * format->channels, format, ingest, user);
        remaining -= chunk;
    }
    return count;
}

uint64_t omx_batcher_pending(const omx_batcher* b, float* dst, uint64_t cap) {
    if (!b) return 0;
    if (dst)
        for (size_t i = 0; i < b->samples.size() && i < cap; ++i) dst[i] = b->samples[i];
    return b->samples.size();
}
int omx_batcher_format(const omx_batcher* b, omx_audio_format* out) {
    if (!b || !b->has_format) return 0;
    if (out) *out = b->format;
    return 1;
}

}  // extern "C"
