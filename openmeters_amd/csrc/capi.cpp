// extern "C" surface of libomx_hip.so — exactly the entry points declared in include/omx.h.
// Every call is wrapped so that no C++ exception crosses the ABI; backend failures come back as
// negative omx_status values with the text available from omx_last_error().
#include "common.hpp"
#include "spectrogram.hpp"
#include "spectrum.hpp"

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

extern "C" {

const char* omx_last_error(void) { return last_error().c_str(); }
int omx_device_available(void) { return device_ready() == OMX_NONE ? 1 : 0; }
const char* omx_version(void) { return "openmeters_amd 0.1 (HIP, gfx950)"; }
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
int omx_spectrogram_bank_set_option(omx_spectrogram_bank* b, uint32_t option, uint64_t value) {
    if (!b) return OMX_ERR_INVALID;
    switch (option) {
        case OMX_OPT_KERNEL_TIMING: b->impl.timer().enabled = value != 0; return OMX_NONE;
        case OMX_OPT_FORCE_GENERIC: b->impl.force_generic(value != 0); return OMX_NONE;
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

}  // extern "C"
