// extern "C" surface of libomx_hip.so — exactly the entry points declared in include/omx.h.
// Every call is wrapped so that no C++ exception crosses the ABI; backend failures come back as
// negative omx_status values with the text available from omx_last_error().
#include "common.hpp"
#include "spectrogram.hpp"
#include "spectrum.hpp"
#include "loudness.hpp"
#include "stereometer.hpp"
#include "capture_group.hpp"
#include "oscilloscope.hpp"
#include "waveform.hpp"

namespace omx {
const std::string& last_error();
}
using namespace omx;

#define REQUIRE_DEVICE()                       \
    do {                                       \
        const int _rc = device_ready();        \
        if (_rc != OMX_NONE) return _rc;       \
    } while (0)

struct omx_spectrogram {
    SpectrogramSingle impl;
    explicit omx_spectrogram(const omx_spectrogram_config& c) : impl(c) {}
};
struct omx_spectrogram_bank {
    SpectrogramBank impl;
    omx_spectrogram_bank(const omx_spectrogram_config& c, uint32_t n) : impl(c, n) {}
};

struct omx_spectrum {
    SpectrumSingle impl;
    explicit omx_spectrum(const omx_spectrum_config& c) : impl(c) {}
};
struct omx_spectrum_bank {
    SpectrumBank impl;
    omx_spectrum_bank(const omx_spectrum_config& c, uint32_t n, bool all) : impl(c, n, all) {}
};

struct omx_loudness {
    LoudnessBank bank;
    explicit omx_loudness(const omx_loudness_config& c) : bank(c, 1) { bank.host_outputs(true); }
};
struct omx_loudness_bank {
    LoudnessBank impl;
    omx_loudness_bank(const omx_loudness_config& c, uint32_t n) : impl(c, n) {}
};
struct omx_stereometer {
    StereometerBank bank;
    std::vector<float> points[4];
    explicit omx_stereometer(const omx_stereometer_config& c) : bank(c, 1) { bank.host_outputs(true); }
};
struct omx_stereometer_bank {
    StereometerBank impl;
    omx_stereometer_bank(const omx_stereometer_config& c, uint32_t n) : impl(c, n) {}
};
struct omx_oscilloscope {
    OscilloscopeBank bank;
    std::vector<float> samples;
    ScopeBlockHeader last{};
    bool have_last = false;
    explicit omx_oscilloscope(const omx_oscilloscope_config& c) : bank(c, 1) { bank.host_outputs(true); }
};
struct omx_oscilloscope_bank {
    OscilloscopeBank impl;
    omx_oscilloscope_bank(const omx_oscilloscope_config& c, uint32_t n) : impl(c, n) {}
};

struct omx_waveform {
    WaveformBank bank;
    std::vector<omx_wave_column> columns;
    explicit omx_waveform(const omx_waveform_config& c) : bank(c, 1) { bank.host_outputs(true); }
};
struct omx_waveform_bank {
    WaveformBank impl;
    omx_waveform_bank(const omx_waveform_config& c, uint32_t n) : impl(c, n) {}
};

struct omx_capture_group {
    CaptureGroup impl;
    explicit omx_capture_group(const omx_capture_group_config& c) : impl(c) {}
};

extern "C" {

const char* omx_last_error(void) { return last_error().c_str(); }
int omx_device_available(void) { return device_ready() == OMX_NONE ? 1 : 0; }
int omx_device_count(void) { return device_count(); }
int omx_set_device(int index) { return select_device(index); }
const char* omx_version(void) { return "openmeters_amd 0.6 (HIP, gfx950)"; }
int omx_abi_version(void) { return OMX_ABI_VERSION; }
void omx_positions_fallback(uint32_t channels, uint8_t out[OMX_MAX_CHANNELS]) { positions_fallback(channels, out); }
void omx_positions_normalize(uint32_t channels, const uint8_t in[OMX_MAX_CHANNELS], uint8_t out[OMX_MAX_CHANNELS]) {
    positions_normalize(channels, in, out);
}

// ------------------------------------------------------------------ spectrogram
void omx_spectrogram_config_default(omx_spectrogram_config* out) {
    if (out) spectrogram_config_default(out);
}
int omx_spectrogram_create(const omx_spectrogram_config* cfg, omx_spectrogram** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_spectrogram(*cfg);
        return (int)OMX_NONE;
    });
}
void omx_spectrogram_destroy(omx_spectrogram* h) { delete h; }
int omx_spectrogram_get_config(const omx_spectrogram* h, omx_spectrogram_config* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    *out = h->impl.bank.config();
    return OMX_NONE;
}
int omx_spectrogram_update_config(omx_spectrogram* h, const omx_spectrogram_config* cfg) {
    if (!h || !cfg) return OMX_ERR_INVALID;
    return guarded([&] {
        h->impl.bank.update_config(*cfg, nullptr);
        return (int)OMX_NONE;
    });
}
int omx_spectrogram_reset_audio(omx_spectrogram* h) {
    if (!h) return OMX_ERR_INVALID;
    return guarded([&] {
        h->impl.bank.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_spectrogram_prepare(omx_spectrogram* h) {
    if (!h) return OMX_ERR_INVALID;
    return guarded([&] {
        h->impl.bank.prepare(nullptr);
        return (int)OMX_NONE;
    });
}
int omx_spectrogram_process_block(omx_spectrogram* h, const omx_block* block, omx_spectrogram_update* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    return guarded([&] { return h->impl.process_block(block, out); });
}
uint16_t omx_pack_classic_db(float db) { return pack_classic_db_host(db); }
uint64_t omx_spectrogram_history_columns(uint32_t kind, uint32_t points, uint64_t requested) {
    return history_columns(kind, points, requested);
}

int omx_spectrogram_bank_create(const omx_spectrogram_config* cfg, uint32_t n_streams, omx_spectrogram_bank** out) {
    if (!cfg || !out || n_streams == 0) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_spectrogram_bank(*cfg, n_streams);
        return (int)OMX_NONE;
    });
}
void omx_spectrogram_bank_destroy(omx_spectrogram_bank* b) { delete b; }
int omx_spectrogram_bank_update_config(omx_spectrogram_bank* b, const omx_spectrogram_config* cfg) {
    if (!b || !cfg) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.update_config(*cfg, b->impl.last_stream());
        return (int)OMX_NONE;
    });
}
int omx_spectrogram_bank_reset_audio(omx_spectrogram_bank* b) {
    if (!b) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_spectrogram_bank_process(omx_spectrogram_bank* b, const float* pcm, int pcm_on_device, uint64_t frames,
                                 uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS],
                                 void* stream, omx_spectrogram_bank_update* out) {
    if (!b || !pcm || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process(pcm, pcm_on_device != 0, frames, channels, sample_rate, positions,
                               static_cast<hipStream_t>(stream), out);
    });
}
int omx_spectrogram_bank_fetch_column(omx_spectrogram_bank* b, uint64_t stream_index, uint64_t column, void* dst,
                                      uint64_t dst_capacity_elems, uint64_t* n_out) {
    if (!b || !dst) return OMX_ERR_INVALID;
    return guarded([&] { return b->impl.fetch_column(stream_index, column, dst, dst_capacity_elems, n_out, b->impl.last_stream()); });
}
int omx_spectrogram_bank_kernel_time(omx_spectrogram_bank* b, double* avg_ms, uint64_t* launches) {
    if (!b || !avg_ms) return OMX_ERR_INVALID;
    return guarded([&] {
        *avg_ms = b->impl.timer().collect(launches);
        return (int)OMX_NONE;
    });
}
int omx_debug_scope_phase_cycles(uint64_t* out, uint32_t n, int reset) {
    if (!out || n < (uint32_t)SCOPE_PHASES) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        unsigned long long c[SCOPE_PHASES], w[SCOPE_PHASES];
        scope_phase_cycles(c, reset != 0);      // single-pass kernel (other sample rates)
        scope_fast_phase_cycles(w, reset != 0);  // wide form's trigger kernel: only one of the two ran
        for (int i = 0; i < SCOPE_PHASES; ++i) out[i] = c[i] + w[i];
        return (int)SCOPE_PHASES;
    });
}
int omx_debug_scope_find_best(const float* work, const float* tmpl, uint32_t len, uint32_t search, float period, uint32_t* best_off,
                              float* frac_offset, float* best_score, float* scores) {
    if (!work || !tmpl || !best_off || !frac_offset || !best_score || !scores || len < 8 || search == 0) return OMX_ERR_INVALID;
    if ((uint64_t)(len + search + 16) * sizeof(float) + (uint64_t)(len + 16) * sizeof(float) > 150 * 1024) return OMX_ERR_UNSUPPORTED;
    REQUIRE_DEVICE();
    return guarded([&] {
        DeviceBuffer<float> d_work, d_tmpl, d_out;
        DeviceBuffer<uint32_t> d_off;
        d_work.reserve((size_t)len + search);
        d_tmpl.reserve(len);
        d_out.reserve((size_t)search + 3);
        d_off.reserve(1);
        OMX_HIP(hipMemcpy(d_work.ptr, work, ((size_t)len + search) * sizeof(float), hipMemcpyHostToDevice));
        OMX_HIP(hipMemcpy(d_tmpl.ptr, tmpl, (size_t)len * sizeof(float), hipMemcpyHostToDevice));
        launch_scope_find_best_debug(d_work.ptr, d_tmpl.ptr, len, search, period, d_off.ptr, d_out.ptr, d_out.ptr + 1, d_out.ptr + 2, nullptr);
        OMX_HIP(hipGetLastError());
        OMX_HIP(hipDeviceSynchronize());
        std::vector<float> h((size_t)search + 3);
        OMX_HIP(hipMemcpy(h.data(), d_out.ptr, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        OMX_HIP(hipMemcpy(best_off, d_off.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost));
        *frac_offset = h[0];
        *best_score = h[1];
        std::memcpy(scores, h.data() + 2, ((size_t)search + 1) * sizeof(float));
        return (int)OMX_PRODUCED;
    });
}
int omx_debug_window_sums(const float* samples, uint32_t n_streams, uint64_t cap, uint64_t tail, uint32_t hop, uint32_t window, uint32_t n_hops,
                          float* sums) {
    if (!samples || !sums || n_streams == 0 || n_hops == 0 || hop == 0 || window == 0 || cap == 0 || (cap & (cap - 1)) != 0 || cap > (uint64_t(1) << 30))
        return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        DeviceBuffer<float> d_ring, d_sums;
        d_ring.reserve((size_t)(n_streams * cap));
        d_sums.reserve((size_t)n_streams * n_hops);
        OMX_HIP(hipMemcpy(d_ring.ptr, samples, (size_t)(n_streams * cap) * sizeof(float), hipMemcpyHostToDevice));
        WindowSumArgs w{};
        w.ring[0] = d_ring.ptr;
        w.n_rings = 1;
        w.cap = cap;
        w.tail = tail;
        w.hop = hop;
        w.window = window;
        w.n_hops = n_hops;
        w.n_streams = n_streams;
        w.sums = d_sums.ptr;
        launch_window_sums(w, nullptr);
        OMX_HIP(hipGetLastError());
        OMX_HIP(hipDeviceSynchronize());
        OMX_HIP(hipMemcpy(sums, d_sums.ptr, (size_t)n_streams * n_hops * sizeof(float), hipMemcpyDeviceToHost));
        return (int)OMX_PRODUCED;
    });
}
int omx_debug_transforms_per_frame(void) { return stft_reassigned_4096_transforms_per_frame(); }
int omx_debug_k_weighting_transition(double sample_rate, uint64_t frames, double* out) {
    if (!out || frames == 0 || !(sample_rate > 0.0)) return OMX_ERR_INVALID;
    k_weighting_transition_debug(sample_rate, frames, out);
    return OMX_NONE;
}
int omx_debug_k2_phase_cycles(uint64_t* out, uint32_t n, int reset) {
    if (!out || n < (uint32_t)K2_PHASES) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
#ifndef OMX_TUNING
    (void)reset;
    set_last_error("omx_debug_k2_phase_cycles: phase-timing kernels exist only in the tuning build (make TUNING=1)");
    return OMX_ERR_UNSUPPORTED;
#else
    return guarded([&] {
        unsigned long long c[K2_PHASES];
        k2_phase_cycles(c, reset != 0);
        for (int i = 0; i < K2_PHASES; ++i) out[i] = c[i];
        return (int)K2_PHASES;
    });
#endif
}
int omx_spectrogram_bank_process_ragged(omx_spectrogram_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                        const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                        const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_spectrogram_ragged_update* out) {
    if (!b || !pcm || !frames || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process_ragged(pcm, frames_capacity, frames, reset_mask, channels, sample_rate, positions, static_cast<hipStream_t>(stream), out);
    });
}
int omx_spectrogram_bank_set_option(omx_spectrogram_bank* b, uint32_t option, uint64_t value) {
    if (!b) return OMX_ERR_INVALID;
    switch (option) {
        case OMX_OPT_KERNEL_TIMING: b->impl.timer().enabled = value != 0; return OMX_NONE;
        case OMX_OPT_FORCE_GENERIC: b->impl.force_generic(value != 0); return OMX_NONE;
        case OMX_OPT_KERNEL_FORM:
            if (value != 0 && value != 1 && value != 2 && value != 30 && value != 31) return OMX_ERR_INVALID;
#ifndef OMX_TUNING
            if (value == 1 || value == 2) {  // the round-1 kernel and the round-2 pair kernel: superseded, compiled into the tuning library only
                set_last_error("OMX_OPT_KERNEL_FORM 1 / 2: superseded kernels, built by `make TUNING=1` (libomx_hip_tuning.so) only");
                return OMX_ERR_UNSUPPORTED;
            }
#endif
            b->impl.kernel_form((int)value);
            return OMX_NONE;
        default: return OMX_ERR_INVALID;
    }
}

// ------------------------------------------------------------------ spectrum
void omx_spectrum_config_default(omx_spectrum_config* out) {
    if (out) spectrum_config_default(out);
}
int omx_spectrum_create(const omx_spectrum_config* cfg, omx_spectrum** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_spectrum(*cfg);
        return (int)OMX_NONE;
    });
}
void omx_spectrum_destroy(omx_spectrum* h) { delete h; }
int omx_spectrum_get_config(const omx_spectrum* h, omx_spectrum_config* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    *out = h->impl.bank.config();
    return OMX_NONE;
}
int omx_spectrum_update_config(omx_spectrum* h, const omx_spectrum_config* cfg) {
    if (!h || !cfg) return OMX_ERR_INVALID;
    return guarded([&] {
        h->impl.bank.update_config(*cfg, nullptr);
        return (int)OMX_NONE;
    });
}
int omx_spectrum_reset_audio(omx_spectrum* h) {
    if (!h) return OMX_ERR_INVALID;
    return guarded([&] {
        h->impl.bank.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_spectrum_prepare(omx_spectrum* h) {
    if (!h) return OMX_ERR_INVALID;
    return guarded([&] {
        h->impl.bank.prepare(nullptr);
        return (int)OMX_NONE;
    });
}
int omx_spectrum_process_block(omx_spectrum* h, const omx_block* block, omx_spectrum_snapshot* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    return guarded([&] { return h->impl.process_block(block, out); });
}
float omx_a_weight(float freq_hz) { return a_weight_host(freq_hz); }

int omx_spectrum_bank_create(const omx_spectrum_config* cfg, uint32_t n_streams, int emit_all_hops, omx_spectrum_bank** out) {
    if (!cfg || !out || n_streams == 0) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_spectrum_bank(*cfg, n_streams, emit_all_hops != 0);
        return (int)OMX_NONE;
    });
}
void omx_spectrum_bank_destroy(omx_spectrum_bank* b) { delete b; }
int omx_spectrum_bank_update_config(omx_spectrum_bank* b, const omx_spectrum_config* cfg) {
    if (!b || !cfg) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.update_config(*cfg, nullptr);
        return (int)OMX_NONE;
    });
}
int omx_spectrum_bank_reset_audio(omx_spectrum_bank* b) {
    if (!b) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_spectrum_bank_process(omx_spectrum_bank* b, const float* pcm, int pcm_on_device, uint64_t frames, uint32_t channels,
                              float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], void* stream,
                              omx_spectrum_bank_update* out) {
    if (!b || !pcm || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process(pcm, pcm_on_device != 0, frames, channels, sample_rate, positions,
                               static_cast<hipStream_t>(stream), out);
    });
}
int omx_spectrum_bank_process_ragged(omx_spectrum_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                     const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                     const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_spectrum_ragged_update* out) {
    if (!b || !pcm || !frames || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process_ragged(pcm, frames_capacity, frames, reset_mask, channels, sample_rate, positions, static_cast<hipStream_t>(stream), out);
    });
}
int omx_spectrum_bank_fetch(omx_spectrum_bank* b, uint64_t stream_index, uint64_t hop, float* dst) {
    if (!b || !dst) return OMX_ERR_INVALID;
    return guarded([&] { return b->impl.fetch(stream_index, hop, dst, b->impl.last_stream()); });
}
int omx_spectrum_bank_set_option(omx_spectrum_bank* b, uint32_t option, uint64_t value) {
    if (!b) return OMX_ERR_INVALID;
    switch (option) {
        case OMX_OPT_KERNEL_TIMING: b->impl.timer().enabled = value != 0; return OMX_NONE;
        case OMX_OPT_FORCE_GENERIC: b->impl.force_generic(value != 0); return OMX_NONE;
        default: return OMX_ERR_INVALID;
    }
}

// ------------------------------------------------------------------ loudness
void omx_loudness_config_default(omx_loudness_config* out) {
    if (out) loudness_config_default(out);
}
int omx_loudness_create(const omx_loudness_config* cfg, omx_loudness** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_loudness(*cfg);
        return (int)OMX_NONE;
    });
}
void omx_loudness_destroy(omx_loudness* h) { delete h; }
int omx_loudness_reset_audio(omx_loudness* h) {
    if (!h) return OMX_ERR_INVALID;
    return guarded([&] {
        h->bank.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_loudness_process_block(omx_loudness* h, const omx_block* block, omx_loudness_snapshot* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    return guarded([&] {
        const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(block->channels, 1), OMX_MAX_CHANNELS);
        if (block->n_samples < channels) return (int)OMX_NONE;
        const int rc = h->bank.process(block->samples, false, block->n_samples / channels, 1, channels, block->sample_rate,
                                       block->positions, nullptr, nullptr);
        if (rc != OMX_PRODUCED) return rc;
        const int frc = h->bank.fetch(0, 0, out, nullptr);
        return frc < 0 ? frc : (int)OMX_PRODUCED;
    });
}
void omx_k_weighting_coefficients(double fs, double b[5], double a[5]) { k_weighting_coefficients(fs, b, a); }

int omx_loudness_bank_create(const omx_loudness_config* cfg, uint32_t n_streams, uint32_t channels, omx_loudness_bank** out) {
    (void)channels;  // the channel count is taken from each call, like LoudnessProcessor::ensure_state
    if (!cfg || !out || n_streams == 0) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_loudness_bank(*cfg, n_streams);
        return (int)OMX_NONE;
    });
}
void omx_loudness_bank_destroy(omx_loudness_bank* b) { delete b; }
int omx_loudness_bank_reset_audio(omx_loudness_bank* b) {
    if (!b) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_loudness_bank_process(omx_loudness_bank* b, const float* pcm, int pcm_on_device, uint64_t block_frames, uint64_t n_blocks,
                              uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], void* stream,
                              const omx_loudness_snapshot** d_snapshots) {
    if (!b || !pcm || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process(pcm, pcm_on_device != 0, block_frames, n_blocks, channels, sample_rate, positions,
                               static_cast<hipStream_t>(stream), d_snapshots);
    });
}
int omx_loudness_bank_process_ragged(omx_loudness_bank* b, const float* pcm, uint64_t block_frames, uint64_t max_blocks,
                                     const uint32_t* n_blocks, const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                     const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_loudness_ragged_update* out) {
    if (!b || !pcm || !n_blocks || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process_ragged(pcm, block_frames, max_blocks, n_blocks, reset_mask, channels, sample_rate, positions,
                                      static_cast<hipStream_t>(stream), out);
    });
}
int omx_loudness_bank_process_chunks(omx_loudness_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                         const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                         const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_loudness_ragged_update* out) {
    if (!b || !pcm || !frames || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process_chunks(pcm, frames_capacity, frames, reset_mask, channels, sample_rate, positions,
                                      static_cast<hipStream_t>(stream), out);
    });
}
int omx_loudness_bank_fetch(omx_loudness_bank* b, uint64_t stream_index, uint64_t block, omx_loudness_snapshot* dst) {
    if (!b || !dst) return OMX_ERR_INVALID;
    return guarded([&] { return b->impl.fetch(stream_index, block, dst, b->impl.last_stream()); });
}
int omx_loudness_bank_kernel_time(omx_loudness_bank* b, double* avg_ms, uint64_t* launches) {
    if (!b || !avg_ms) return OMX_ERR_INVALID;
    return guarded([&] {
        *avg_ms = b->impl.timer().collect(launches);
        return (int)OMX_NONE;
    });
}
int omx_debug_loudness_bank_last_form(const omx_loudness_bank* b) { return b ? b->impl.last_form() : OMX_ERR_INVALID; }
int omx_loudness_bank_set_option(omx_loudness_bank* b, uint32_t option, uint64_t value) {
    if (!b) return OMX_ERR_INVALID;
    if (option == OMX_OPT_KERNEL_TIMING) {
        b->impl.timer().enabled = value != 0;
        return OMX_NONE;
    }
    if (option == OMX_OPT_KERNEL_FORM && value <= 2) {  // 0 = by call shape, 1 = sequential kernels, 2 = chunk-parallel when the shape allows
        b->impl.chunked_mode(value == 0 ? -1 : (value == 1 ? 0 : 1));
        return OMX_NONE;
    }
    if (option == OMX_OPT_LOUDNESS_REBASE_FRAMES) {
        b->impl.rebase_frames(value);
        return OMX_NONE;
    }
    return OMX_ERR_INVALID;
}

// ------------------------------------------------------------------ stereometer
void omx_stereometer_config_default(omx_stereometer_config* out) {
    if (out) stereometer_config_default(out);
}
int omx_stereometer_create(const omx_stereometer_config* cfg, omx_stereometer** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_stereometer(*cfg);
        return (int)OMX_NONE;
    });
}
void omx_stereometer_destroy(omx_stereometer* h) { delete h; }
int omx_stereometer_get_config(const omx_stereometer* h, omx_stereometer_config* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    *out = h->bank.config();
    return OMX_NONE;
}
int omx_stereometer_update_config(omx_stereometer* h, const omx_stereometer_config* cfg) {
    if (!h || !cfg) return OMX_ERR_INVALID;
    return guarded([&] {
        h->bank.update_config(*cfg);
        return (int)OMX_NONE;
    });
}
int omx_stereometer_reset_audio(omx_stereometer* h) {
    if (!h) return OMX_ERR_INVALID;
    return guarded([&] {
        h->bank.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_stereometer_process_block(omx_stereometer* h, const omx_block* block, omx_stereometer_snapshot* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    return guarded([&] {
        const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(block->channels, 1), OMX_MAX_CHANNELS);
        if (block->n_samples < channels) return (int)OMX_NONE;
        omx_stereometer_bank_update bu;
        const int rc = h->bank.process(block->samples, false, block->n_samples / channels, 1, channels, block->sample_rate,
                                       block->positions, nullptr, &bu);
        if (rc != OMX_PRODUCED) return rc;
        uint32_t produced = 0;
        int frc = h->bank.fetch(0, 0, out->correlations, &produced, nullptr);
        if (frc < 0) return frc;
        for (uint32_t band = 0; band < 4; ++band) {
            h->points[band].resize((size_t)h->bank.target() * 2);
            uint64_t n = 0;
            frc = h->bank.fetch_points(0, band, h->points[band].data(), &n, nullptr);
            if (frc < 0) return frc;
            out->points[band] = h->points[band].data();
            out->n_points[band] = n;
        }
        return (int)OMX_PRODUCED;
    });
}

int omx_stereometer_bank_create(const omx_stereometer_config* cfg, uint32_t n_streams, omx_stereometer_bank** out) {
    if (!cfg || !out || n_streams == 0) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_stereometer_bank(*cfg, n_streams);
        return (int)OMX_NONE;
    });
}
void omx_stereometer_bank_destroy(omx_stereometer_bank* b) { delete b; }
int omx_stereometer_bank_update_config(omx_stereometer_bank* b, const omx_stereometer_config* cfg) {
    if (!b || !cfg) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.update_config(*cfg);
        return (int)OMX_NONE;
    });
}
int omx_stereometer_bank_reset_audio(omx_stereometer_bank* b) {
    if (!b) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_stereometer_bank_process_ragged(omx_stereometer_bank* b, const float* pcm, uint64_t block_frames, uint64_t max_blocks,
                                        const uint32_t* n_blocks, const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                        const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_stereometer_ragged_update* out) {
    if (!b || !pcm || !n_blocks || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process_ragged(pcm, block_frames, max_blocks, n_blocks, reset_mask, channels, sample_rate, positions,
                                      static_cast<hipStream_t>(stream), out);
    });
}
int omx_stereometer_bank_process(omx_stereometer_bank* b, const float* pcm, int pcm_on_device, uint64_t block_frames,
                                 uint64_t n_blocks, uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS],
                                 void* stream, omx_stereometer_bank_update* out) {
    if (!b || !pcm || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process(pcm, pcm_on_device != 0, block_frames, n_blocks, channels, sample_rate, positions,
                               static_cast<hipStream_t>(stream), out);
    });
}
int omx_stereometer_bank_process_chunks(omx_stereometer_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                            const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                            const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_stereometer_ragged_update* out) {
    if (!b || !pcm || !frames || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process_chunks(pcm, frames_capacity, frames, reset_mask, channels, sample_rate, positions,
                                      static_cast<hipStream_t>(stream), out);
    });
}
int omx_stereometer_bank_fetch(omx_stereometer_bank* b, uint64_t stream_index, uint64_t block, float correlations[4],
                               uint32_t* produced) {
    if (!b || !correlations) return OMX_ERR_INVALID;
    return guarded([&] { return b->impl.fetch(stream_index, block, correlations, produced, b->impl.last_stream()); });
}

int omx_stereometer_bank_fetch_points(omx_stereometer_bank* b, uint64_t stream_index, uint32_t band, float* dst, uint64_t cap_pairs,
                                      uint64_t* n_pairs) {
    if (!b || !dst || !n_pairs) return OMX_ERR_INVALID;
    if (cap_pairs < b->impl.target()) return OMX_ERR_INVALID;
    return guarded([&] { return b->impl.fetch_points(stream_index, band, dst, n_pairs, b->impl.last_stream()); });
}
int omx_stereometer_bank_set_option(omx_stereometer_bank* b, uint32_t option, uint64_t value) {
    if (!b || option != OMX_OPT_KERNEL_FORM || value > 2) return OMX_ERR_INVALID;
    b->impl.chunked_mode(value == 0 ? -1 : (value == 1 ? 0 : 1));
    return OMX_NONE;
}

// ------------------------------------------------------------------ oscilloscope
void omx_oscilloscope_config_default(omx_oscilloscope_config* out) {
    if (out) oscilloscope_config_default(out);
}
int omx_oscilloscope_create(const omx_oscilloscope_config* cfg, omx_oscilloscope** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_oscilloscope(*cfg);
        return (int)OMX_NONE;
    });
}
void omx_oscilloscope_destroy(omx_oscilloscope* h) { delete h; }
int omx_oscilloscope_get_config(const omx_oscilloscope* h, omx_oscilloscope_config* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    *out = h->bank.config();
    return OMX_NONE;
}
int omx_oscilloscope_update_config(omx_oscilloscope* h, const omx_oscilloscope_config* cfg) {
    if (!h || !cfg) return OMX_ERR_INVALID;
    return guarded([&] {
        const uint64_t before = h->bank.epoch();
        h->bank.update_config(*cfg);
        if (h->bank.epoch() != before) h->have_last = false;  // rebuilt: every trigger starts unlocked
        return (int)OMX_NONE;
    });
}
int omx_oscilloscope_reset_audio(omx_oscilloscope* h) {
    if (!h) return OMX_ERR_INVALID;
    return guarded([&] {
        h->bank.reset_audio();
        h->have_last = false;
        return (int)OMX_NONE;
    });
}
int omx_oscilloscope_process_block(omx_oscilloscope* h, const omx_block* block, omx_oscilloscope_snapshot* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    return guarded([&] {
        const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(block->channels, 1), OMX_MAX_CHANNELS);
        if (block->n_samples < channels) return (int)OMX_NONE;
        const int rc = h->bank.process(block->samples, false, block->n_samples / channels, 1, channels, block->sample_rate,
                                       block->positions, nullptr);
        if (rc != OMX_PRODUCED) return rc;
        int frc = h->bank.fetch_header(0, 0, &h->last, nullptr);
        if (frc < 0) return frc;
        h->have_last = true;
        if (!h->last.produced) return (int)OMX_NONE;
        const uint64_t spc = h->last.samples_per_channel, ch = h->last.channels;
        std::vector<float> raw(2 * (size_t)kScopeTarget);
        frc = h->bank.fetch_samples(0, raw.data(), raw.size(), nullptr);
        if (frc < 0) return frc;
        h->samples.resize((size_t)(ch * spc));
        for (uint64_t c = 0; c < ch; ++c)
            std::copy(raw.begin() + c * kScopeTarget, raw.begin() + c * kScopeTarget + spc, h->samples.begin() + c * spc);
        out->epoch = h->bank.epoch();
        out->channels = ch;
        out->slots[0] = h->last.slots[0];
        out->slots[1] = h->last.slots[1];
        out->samples_per_channel = spc;
        out->n_samples = ch * spc;
        out->samples = h->samples.data();
        return (int)OMX_PRODUCED;
    });
}
int omx_oscilloscope_last_capture(const omx_oscilloscope* h, uint32_t* start, float* frac_offset) {
    if (!h || !h->have_last || !h->last.produced) return 0;
    if (start) *start = h->last.capture_start;
    if (frac_offset) *frac_offset = h->last.capture_frac;
    return 1;
}
int omx_oscilloscope_last_cycle_rate(const omx_oscilloscope* h, float* hz) {
    if (!h || !hz || !h->have_last || !h->last.locked) return 0;
    *hz = h->bank.config().sample_rate / h->last.period;
    return 1;
}

int omx_oscilloscope_bank_create(const omx_oscilloscope_config* cfg, uint32_t n_streams, omx_oscilloscope_bank** out) {
    if (!cfg || !out || n_streams == 0) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_oscilloscope_bank(*cfg, n_streams);
        return (int)OMX_NONE;
    });
}
void omx_oscilloscope_bank_destroy(omx_oscilloscope_bank* b) { delete b; }
int omx_oscilloscope_bank_update_config(omx_oscilloscope_bank* b, const omx_oscilloscope_config* cfg) {
    if (!b || !cfg) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.update_config(*cfg);
        return (int)OMX_NONE;
    });
}
int omx_oscilloscope_bank_reset_audio(omx_oscilloscope_bank* b) {
    if (!b) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_oscilloscope_bank_process_ragged(omx_oscilloscope_bank* b, const float* pcm, uint64_t block_frames, uint64_t max_blocks,
                                         const uint32_t* n_blocks, const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                         const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_oscilloscope_ragged_update* out) {
    if (!b || !pcm || !n_blocks || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process_ragged(pcm, block_frames, max_blocks, n_blocks, reset_mask, channels, sample_rate, positions,
                                      static_cast<hipStream_t>(stream), out);
    });
}
int omx_oscilloscope_bank_process(omx_oscilloscope_bank* b, const float* pcm, int pcm_on_device, uint64_t block_frames,
                                  uint64_t n_blocks, uint32_t channels, float sample_rate,
                                  const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_oscilloscope_bank_update* out) {
    if (!b || !pcm || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        const int rc = b->impl.process(pcm, pcm_on_device != 0, block_frames, n_blocks, channels, sample_rate, positions,
                                       static_cast<hipStream_t>(stream));
        if (rc == OMX_PRODUCED && out) {
            out->n_streams = b->impl.n_streams();
            out->n_blocks = n_blocks;
            out->epoch = b->impl.epoch();
            out->d_headers = reinterpret_cast<const omx_oscilloscope_block_header*>(b->impl.d_headers());
            out->d_samples = b->impl.d_samples();
            out->sample_stride = kScopeTarget;
        }
        return rc;
    });
}
int omx_oscilloscope_bank_process_chunks(omx_oscilloscope_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                             const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                             const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_oscilloscope_ragged_update* out) {
    if (!b || !pcm || !frames || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process_chunks(pcm, frames_capacity, frames, reset_mask, channels, sample_rate, positions,
                                      static_cast<hipStream_t>(stream), out);
    });
}
int omx_oscilloscope_bank_fetch(omx_oscilloscope_bank* b, uint64_t stream_index, uint64_t block,
                                omx_oscilloscope_block_header* header, float* samples /* [2][4096] or NULL */) {
    if (!b || !header) return OMX_ERR_INVALID;
    return guarded([&] {
        static_assert(sizeof(omx_oscilloscope_block_header) == sizeof(ScopeBlockHeader), "header layout");
        int rc = b->impl.fetch_header(stream_index, block, reinterpret_cast<ScopeBlockHeader*>(header), b->impl.last_stream());
        if (rc < 0) return rc;
        if (samples) rc = b->impl.fetch_samples(stream_index, samples, 2 * kScopeTarget, b->impl.last_stream());
        return rc;
    });
}

// ------------------------------------------------------------------ waveform
void omx_waveform_config_default(omx_waveform_config* out) {
    if (out) waveform_config_default(out);
}
int omx_waveform_create(const omx_waveform_config* cfg, omx_waveform** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_waveform(*cfg);
        return (int)OMX_NONE;
    });
}
void omx_waveform_destroy(omx_waveform* h) { delete h; }
int omx_waveform_get_config(const omx_waveform* h, omx_waveform_config* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    *out = h->bank.config();
    return OMX_NONE;
}
int omx_waveform_update_config(omx_waveform* h, const omx_waveform_config* cfg) {
    if (!h || !cfg) return OMX_ERR_INVALID;
    return guarded([&] {
        h->bank.update_config(*cfg);
        return (int)OMX_NONE;
    });
}
int omx_waveform_reset_audio(omx_waveform* h) {
    if (!h) return OMX_ERR_INVALID;
    return guarded([&] {
        h->bank.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_waveform_prepare(omx_waveform* h) {
    if (!h) return OMX_ERR_INVALID;
    return guarded([&] {
        h->bank.prepare(nullptr);
        return (int)OMX_NONE;
    });
}
int omx_waveform_process_block(omx_waveform* h, const omx_block* block, omx_waveform_update* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    return guarded([&] {
        const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(block->channels, 1), OMX_MAX_CHANNELS);
        if (block->n_samples < channels) return (int)OMX_NONE;
        omx_waveform_bank_update bu;
        const int rc = h->bank.process(block->samples, false, block->n_samples / channels, channels, block->sample_rate,
                                       block->positions, nullptr, &bu);
        if (rc != OMX_PRODUCED) return rc;
        h->columns.resize((size_t)bu.n_columns * 4);
        std::memset(out, 0, sizeof(*out));
        const int frc = h->bank.fetch(0, h->columns.data(), out->preview, nullptr);
        if (frc < 0) return frc;
        out->n_columns = bu.n_columns;
        out->columns = h->columns.data();
        out->reset = bu.reset;
        out->preview_some = bu.preview_some;
        out->preview_progress = bu.preview_progress;
        return (int)OMX_PRODUCED;
    });
}
int omx_waveform_bank_create(const omx_waveform_config* cfg, uint32_t n_streams, omx_waveform_bank** out) {
    if (!cfg || !out || n_streams == 0) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_waveform_bank(*cfg, n_streams);
        return (int)OMX_NONE;
    });
}
void omx_waveform_bank_destroy(omx_waveform_bank* b) { delete b; }
int omx_waveform_bank_update_config(omx_waveform_bank* b, const omx_waveform_config* cfg) {
    if (!b || !cfg) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.update_config(*cfg);
        return (int)OMX_NONE;
    });
}
int omx_waveform_bank_reset_audio(omx_waveform_bank* b) {
    if (!b) return OMX_ERR_INVALID;
    return guarded([&] {
        b->impl.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_waveform_bank_process_ragged(omx_waveform_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                     const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                     const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_waveform_ragged_update* out) {
    if (!b || !pcm || !frames || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process_ragged(pcm, frames_capacity, frames, reset_mask, channels, sample_rate, positions,
                                      static_cast<hipStream_t>(stream), out);
    });
}
int omx_waveform_bank_process(omx_waveform_bank* b, const float* pcm, int pcm_on_device, uint64_t frames, uint32_t channels,
                              float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], void* stream,
                              omx_waveform_bank_update* out) {
    if (!b || !pcm || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return b->impl.process(pcm, pcm_on_device != 0, frames, channels, sample_rate, positions, static_cast<hipStream_t>(stream), out);
    });
}
int omx_waveform_bank_set_option(omx_waveform_bank* b, uint32_t option, uint64_t value) {
    if (!b || option != OMX_OPT_KERNEL_FORM || value > 2) return OMX_ERR_INVALID;
    b->impl.set_form((uint32_t)value);
    return OMX_NONE;
}
int omx_waveform_bank_fetch(omx_waveform_bank* b, uint64_t stream_index, omx_wave_column* columns, omx_wave_column* preview) {
    if (!b) return OMX_ERR_INVALID;
    return guarded([&] { return b->impl.fetch(stream_index, columns, preview, b->impl.last_stream()); });
}

// ------------------------------------------------------------------ capture group (VisualManager fan-out, registry.rs:396-418)
void omx_capture_group_config_default(omx_capture_group_config* out) {
    if (out) capture_group_config_default(out);
}
int omx_capture_group_create(const omx_capture_group_config* cfg, omx_capture_group** out) {
    if (!cfg || !out || cfg->n_streams == 0) return OMX_ERR_INVALID;
    REQUIRE_DEVICE();
    return guarded([&] {
        *out = new omx_capture_group(*cfg);
        return (int)OMX_NONE;
    });
}
void omx_capture_group_destroy(omx_capture_group* g) { delete g; }
int omx_capture_group_reset_audio(omx_capture_group* g) {
    if (!g) return OMX_ERR_INVALID;
    return guarded([&] {
        g->impl.reset_audio();
        return (int)OMX_NONE;
    });
}
int omx_capture_group_set_option(omx_capture_group* g, uint32_t option, uint64_t value) {
    if (!g) return OMX_ERR_INVALID;
    switch (option) {
        case OMX_OPT_GROUP_STATS: g->impl.set_stats(value != 0); return OMX_NONE;
        case OMX_OPT_GROUP_SHARED_INGEST: g->impl.set_shared_ingest(value != 0); return OMX_NONE;
        case OMX_OPT_KERNEL_TIMING: g->impl.set_timing(value != 0); return OMX_NONE;
        default: return OMX_ERR_INVALID;
    }
}
int omx_capture_group_ingest(omx_capture_group* g, const float* pcm, uint64_t frames, uint32_t channels, float sample_rate,
                             const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_capture_group_update* out) {
    if (!g || (!pcm && frames) || !positions) return OMX_ERR_INVALID;
    return guarded([&] { return g->impl.ingest(pcm, frames, channels, sample_rate, positions, static_cast<hipStream_t>(stream), out); });
}
int omx_capture_group_set_enabled(omx_capture_group* g, uint32_t visual, int enabled) {
    if (!g) return OMX_ERR_INVALID;
    return guarded([&] { return g->impl.set_enabled(visual, enabled != 0); });
}
int omx_capture_group_enabled(const omx_capture_group* g) { return g ? (int)g->impl.enabled() : OMX_ERR_INVALID; }
int omx_capture_group_update_config(omx_capture_group* g, uint32_t visual, const void* config, void* stream) {
    if (!g || !config) return OMX_ERR_INVALID;
    return guarded([&] { return g->impl.update_config(visual, config, static_cast<hipStream_t>(stream)); });
}
int omx_capture_group_note_format(omx_capture_group* g, uint64_t generation) {
    if (!g) return OMX_ERR_INVALID;
    return guarded([&] { return g->impl.note_format_generation(generation) ? 1 : 0; });
}
int omx_capture_group_ingest_ragged(omx_capture_group* g, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                    const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                    const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_capture_group_ragged_update* out) {
    if (!g || !pcm || !frames || !positions) return OMX_ERR_INVALID;
    return guarded([&] {
        return g->impl.ingest_ragged(pcm, frames_capacity, frames, reset_mask, channels, sample_rate, positions, static_cast<hipStream_t>(stream), out);
    });
}
int omx_capture_group_kernel_time(omx_capture_group* g, double* avg_ms, uint64_t* launches) {
    if (!g || !avg_ms || !g->impl.spectrogram()) return OMX_ERR_INVALID;
    return guarded([&] {
        *avg_ms = g->impl.spectrogram()->timer().collect(launches);
        return (int)OMX_NONE;
    });
}

}  // extern "C"
int omx_debug_stereometer_bank_last_form(const omx_stereometer_bank* b) { return b ? b->impl.last_form() : OMX_ERR_INVALID; }
long long omx_debug_oscilloscope_bank_resume_block(const omx_oscilloscope_bank* b, uint32_t stream_index) {
    if (!b) return -1;
    long long v = -1;
    (void)guarded([&] {
        v = b->impl.debug_resume_block(stream_index);
        return (int)OMX_NONE;
    });
    return v;
}
int omx_debug_waveform_bank_last_form(const omx_waveform_bank* b) { return b ? (int)b->impl.last_form() : OMX_ERR_INVALID; }
