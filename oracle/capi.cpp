// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// C entry points of the CPU oracle.  They mirror include/omx.h one for one with an `omxo_`
// prefix (same structs, same return convention) so tests/ can drive the oracle and the HIP
// product through identical Python wrappers.  Extra `omxo_kat_*` exports expose the shared
// primitives for the known-answer tests ported from the reference's in-file unit tests.
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "batcher.hpp"
#include "loudness.hpp"
#include "oscilloscope.hpp"
#include "primitives.hpp"
#include "spectrogram.hpp"
#include "spectrum.hpp"
#include "stereometer.hpp"
#include "splat.hpp"
#include "summary.hpp"
#include "waveform.hpp"

using namespace omxo;

namespace {
AudioBlock to_block(const omx_block* b) {
    Positions p;
    for (int i = 0; i < MAX_CH; ++i) p[i] = b->positions[i];
    return AudioBlock::with_positions(b->samples, (size_t)b->n_samples, b->channels, b->sample_rate, p);
}
SpectrogramConfig from_c(const omx_spectrogram_config& c) {
    SpectrogramConfig o;
    o.sample_rate = c.sample_rate;
    o.fft_size = (size_t)c.fft_size;
    o.hop_size = (size_t)c.hop_size;
    o.window = c.window;
    o.history_length = (size_t)c.history_length;
    o.use_reassignment = c.use_reassignment != 0;
    o.zero_padding_factor = (size_t)c.zero_padding_factor;
    return o;
}
void to_c(const SpectrogramConfig& c, omx_spectrogram_config* o) {
    std::memset(o, 0, sizeof(*o));
    o->sample_rate = c.sample_rate;
    o->fft_size = c.fft_size;
    o->hop_size = c.hop_size;
    o->window = c.window;
    o->history_length = c.history_length;
    o->use_reassignment = c.use_reassignment ? 1 : 0;
    o->zero_padding_factor = c.zero_padding_factor;
}
SpectrumConfig from_c(const omx_spectrum_config& c) {
    SpectrumConfig o;
    o.sample_rate = c.sample_rate;
    o.fft_size = (size_t)c.fft_size;
    o.hop_size = (size_t)c.hop_size;
    o.window = c.window;
    o.averaging_mode = c.averaging_mode;
    o.averaging_param = c.averaging_param;
    o.source = c.source;
    o.secondary_source = c.secondary_source;
    o.floor_db = c.floor_db;
    return o;
}
void to_c(const SpectrumConfig& c, omx_spectrum_config* o) {
    std::memset(o, 0, sizeof(*o));
    o->sample_rate = c.sample_rate;
    o->fft_size = c.fft_size;
    o->hop_size = c.hop_size;
    o->window = c.window;
    o->averaging_mode = c.averaging_mode;
    o->averaging_param = c.averaging_param;
    o->source = c.source;
    o->secondary_source = c.secondary_source;
    o->floor_db = c.floor_db;
}
StereometerConfig from_c(const omx_stereometer_config& c) {
    StereometerConfig o;
    o.sample_rate = c.sample_rate;
    o.segment_duration = c.segment_duration;
    o.target_sample_count = (size_t)c.target_sample_count;
    o.correlation_window = c.correlation_window;
    o.analyze_bands = c.analyze_bands != 0;
    o.emit_band_points = c.emit_band_points != 0;
    return o;
}
void to_c(const StereometerConfig& c, omx_stereometer_config* o) {
    std::memset(o, 0, sizeof(*o));
    o->sample_rate = c.sample_rate;
    o->segment_duration = c.segment_duration;
    o->target_sample_count = c.target_sample_count;
    o->correlation_window = c.correlation_window;
    o->analyze_bands = c.analyze_bands;
    o->emit_band_points = c.emit_band_points;
}
OscilloscopeConfig from_c(const omx_oscilloscope_config& c) {
    OscilloscopeConfig o;
    o.sample_rate = c.sample_rate;
    o.segment_duration = c.segment_duration;
    o.trigger_mode = c.trigger_mode;
    o.num_cycles = (size_t)c.num_cycles;
    o.trigger_source = c.trigger_source;
    o.channel_1 = c.channel_1;
    o.channel_2 = c.channel_2;
    return o;
}
void to_c(const OscilloscopeConfig& c, omx_oscilloscope_config* o) {
    std::memset(o, 0, sizeof(*o));
    o->sample_rate = c.sample_rate;
    o->segment_duration = c.segment_duration;
    o->trigger_mode = c.trigger_mode;
    o->num_cycles = c.num_cycles;
    o->trigger_source = c.trigger_source;
    o->channel_1 = c.channel_1;
    o->channel_2 = c.channel_2;
}
}  // namespace

struct omxo_spectrogram {
    SpectrogramProcessor p;
    SpectrogramUpdate last;
    std::vector<uint64_t> offsets;
    std::vector<omx_spectrogram_point> points;
    std::vector<uint16_t> codes;
    explicit omxo_spectrogram(SpectrogramConfig c) : p(c) {}
};
struct omxo_spectrum {
    SpectrumProcessor p;
    explicit omxo_spectrum(SpectrumConfig c) : p(c) {}
};
struct omxo_loudness {
    LoudnessProcessor p;
    explicit omxo_loudness(LoudnessConfig c) : p(c) {}
};
struct omxo_stereometer {
    StereometerProcessor p;
    StereometerSnapshot last;
    std::vector<float> flat[4];
    explicit omxo_stereometer(StereometerConfig c) : p(c) {}
};
struct omxo_oscilloscope {
    OscilloscopeProcessor p;
    OscilloscopeSnapshot last;
    explicit omxo_oscilloscope(OscilloscopeConfig c) : p(c) {}
};

extern "C" {

int omxo_abi_version(void) { return OMX_ABI_VERSION; }
const char* omxo_version(void) { return "omx-oracle 0.1 (CPU restatement; parity unpinned through the FFT boundary)"; }

void omxo_positions_fallback(uint32_t channels, uint8_t out[8]) {
    const Positions p = positions_fallback(channels);
    for (int i = 0; i < 8; ++i) out[i] = p[i];
}
void omxo_positions_normalize(uint32_t channels, const uint8_t in[8], uint8_t out[8]) {
    Positions p;
    for (int i = 0; i < 8; ++i) p[i] = in[i];
    p = positions_normalize(channels, p);
    for (int i = 0; i < 8; ++i) out[i] = p[i];
}

// ------------------------------------------------------------------ spectrogram
void omxo_spectrogram_config_default(omx_spectrogram_config* out) { to_c(SpectrogramConfig(), out); }
int omxo_spectrogram_create(const omx_spectrogram_config* cfg, omxo_spectrogram** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    *out = new omxo_spectrogram(from_c(*cfg));
    return OMX_NONE;
}
void omxo_spectrogram_destroy(omxo_spectrogram* h) { delete h; }
int omxo_spectrogram_get_config(const omxo_spectrogram* h, omx_spectrogram_config* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    to_c(h->p.config(), out);
    return OMX_NONE;
}
int omxo_spectrogram_update_config(omxo_spectrogram* h, const omx_spectrogram_config* cfg) {
    if (!h || !cfg) return OMX_ERR_INVALID;
    h->p.update_config(from_c(*cfg));
    return OMX_NONE;
}
int omxo_spectrogram_reset_audio(omxo_spectrogram* h) {
    if (!h) return OMX_ERR_INVALID;
    h->p.reset_audio();
    return OMX_NONE;
}
int omxo_spectrogram_prepare(omxo_spectrogram* h) {
    if (!h) return OMX_ERR_INVALID;
    h->p.prepare();
    return OMX_NONE;
}
int omxo_spectrogram_process_block(omxo_spectrogram* h, const omx_block* block, omx_spectrogram_update* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    const AudioBlock b = to_block(block);
    if (!h->p.process_block(b, h->last)) return OMX_NONE;
    const SpectrogramUpdate& u = h->last;
    h->offsets.assign(1, 0);
    h->points.clear();
    h->codes.clear();
    uint32_t kind = OMX_COLUMN_REASSIGNED;
    for (const auto& col : u.new_columns) {
        kind = col.kind;
        if (col.kind == OMX_COLUMN_REASSIGNED) {
            h->points.insert(h->points.end(), col.points.begin(), col.points.end());
            h->offsets.push_back(h->points.size());
        } else {
            h->codes.insert(h->codes.end(), col.codes.begin(), col.codes.end());
            h->offsets.push_back(h->codes.size());
        }
    }
    std::memset(out, 0, sizeof(*out));
    out->fft_size = u.fft_size;
    out->hop_size = u.hop_size;
    out->history_length = u.history_length;
    out->n_columns = u.new_columns.size();
    out->column_offsets = h->offsets.data();
    out->points = h->points.data();
    out->codes = h->codes.data();
    out->sample_rate = u.sample_rate;
    out->reassigned_power_scale = u.reassigned_power_scale;
    out->reset = u.reset ? 1 : 0;
    out->kind = kind;
    return OMX_PRODUCED;
}
uint16_t omxo_pack_classic_db(float db) { return pack_classic_db(db); }
uint64_t omxo_spectrogram_history_columns(uint32_t kind, uint32_t points, uint64_t requested) {
    return history_columns(kind, points, (size_t)requested);
}
// test-only views of private state the reference's own tests inspect
// test hook: record (enable != 0) the input samples of every column of the following updates; omxo_debug_spectrogram_captured copies
// the samples column `index` of the LAST update was computed from (returns their count, 0 when there is no such column)
void omxo_debug_spectrogram_capture(omxo_spectrogram* h, int enable) {
    h->p.set_capture_inputs(enable != 0);
}
uint64_t omxo_debug_spectrogram_captured(const omxo_spectrogram* h, uint64_t index, float* dst, uint64_t cap) {
    if (index >= h->p.captured_inputs().size()) return 0;
    const std::vector<float>& v = h->p.captured_inputs()[index];
    if (dst) std::memcpy(dst, v.data(), std::min<uint64_t>(cap, v.size()) * sizeof(float));
    return v.size();
}
uint64_t omxo_spectrogram_pending(const omxo_spectrogram* h, float* dst, uint64_t cap) {
    const auto& a = h->p.audio_buffer();
    for (size_t i = 0; i < a.size() && i < cap; ++i) dst[i] = a[i];
    return a.size();
}
int omxo_spectrogram_push_audio(omxo_spectrogram* h, const omx_block* block) {
    h->p.push_audio_for_test(to_block(block));
    return OMX_NONE;
}
uint64_t omxo_spectrogram_tables(const omxo_spectrogram* h, int which, float* dst, uint64_t cap) {
    const std::vector<float>* v = nullptr;
    switch (which) {
        case 0: v = &h->p.window(); break;
        case 1: v = &h->p.derivative_window(); break;
        case 2: v = &h->p.time_weighted_window(); break;
        default: v = &h->p.bin_norm(); break;
    }
    for (size_t i = 0; i < v->size() && i < cap; ++i) dst[i] = (*v)[i];
    return v->size();
}

// ------------------------------------------------------------------ spectrum
void omxo_spectrum_config_default(omx_spectrum_config* out) { to_c(SpectrumConfig(), out); }
int omxo_spectrum_create(const omx_spectrum_config* cfg, omxo_spectrum** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    *out = new omxo_spectrum(from_c(*cfg));
    return OMX_NONE;
}
void omxo_spectrum_destroy(omxo_spectrum* h) { delete h; }
int omxo_spectrum_get_config(const omxo_spectrum* h, omx_spectrum_config* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    to_c(h->p.config(), out);
    return OMX_NONE;
}
int omxo_spectrum_update_config(omxo_spectrum* h, const omx_spectrum_config* cfg) {
    if (!h || !cfg) return OMX_ERR_INVALID;
    h->p.update_config(from_c(*cfg));
    return OMX_NONE;
}
int omxo_spectrum_reset_audio(omxo_spectrum* h) {
    if (!h) return OMX_ERR_INVALID;
    h->p.reset_audio();
    return OMX_NONE;
}
int omxo_spectrum_prepare(omxo_spectrum* h) {
    if (!h) return OMX_ERR_INVALID;
    h->p.prepare();
    return OMX_NONE;
}
static void fill_spectrum_snapshot(const SpectrumSnapshot& s, omx_spectrum_snapshot* out) {
    out->bins = s.frequency_bins.size();
    out->frequency_bins = s.frequency_bins.data();
    for (int t = 0; t < 2; ++t)
        for (int w = 0; w < 2; ++w) out->traces[t][w] = s.traces[t][w].data();
}
int omxo_spectrum_process_block(omxo_spectrum* h, const omx_block* block, omx_spectrum_snapshot* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    const SpectrumSnapshot* s = h->p.process_block(to_block(block));
    if (!s) return OMX_NONE;
    fill_spectrum_snapshot(*s, out);
    return OMX_PRODUCED;
}
float omxo_a_weight(float freq_hz) { return a_weight(freq_hz); }
// test-only views
int omxo_spectrum_peek_snapshot(omxo_spectrum* h, omx_spectrum_snapshot* out) {
    fill_spectrum_snapshot(h->p.snapshot(), out);
    return OMX_NONE;
}
uint64_t omxo_spectrum_pending(omxo_spectrum* h, int trace, float* dst, uint64_t cap) {
    const auto& a = h->p.pcm_buffer(trace);
    for (size_t i = 0; i < a.size() && i < cap; ++i) dst[i] = a[i];
    return a.size();
}
void omxo_spectrum_extend_pending(omxo_spectrum* h, int trace, const float* src, uint64_t n) {
    auto& a = h->p.pcm_buffer_mut(trace);
    for (uint64_t i = 0; i < n; ++i) a.push_back(src[i]);
}
// levels: which 0 = smoothed_power, 1 = scratch_power; returns length
uint64_t omxo_spectrum_levels(omxo_spectrum* h, int trace, int which, float* dst, uint64_t cap) {
    auto& l = h->p.levels(trace);
    const auto& v = which == 0 ? l.smoothed_power : l.scratch_power;
    for (size_t i = 0; i < v.size() && i < cap; ++i) dst[i] = v[i];
    return v.size();
}
void omxo_spectrum_fill_smoothed(omxo_spectrum* h, int trace, float value) {
    for (float& v : h->p.levels(trace).smoothed_power) v = value;
}
float omxo_kat_smoothing_state_floor(const float* weighting, uint64_t n, float floor) {
    return smoothing_state_floor(std::vector<float>(weighting, weighting + n), floor);
}
// One SpectrumLevelBuffers::update_outputs step on a 1-bin buffer (reference tests :613-651).
void omxo_kat_level_update(float state_floor, float smoothed_init, float scratch_power, uint32_t mode, float param,
                           float weighting_db, float dt, float floor, float out[3]) {
    SpectrumLevelBuffers b;
    b.reset(1, state_floor, true);
    b.smoothed_power[0] = smoothed_init;
    b.scratch_power[0] = scratch_power;
    std::vector<float> outputs[2];
    b.update_outputs(mode, param, outputs, std::vector<float>{weighting_db}, dt, floor);
    out[0] = outputs[0][0];
    out[1] = outputs[1][0];
    out[2] = b.smoothed_power[0];
}

// ------------------------------------------------------------------ loudness
void omxo_loudness_config_default(omx_loudness_config* out) {
    out->sample_rate = DEFAULT_SAMPLE_RATE;
    out->floor_db = LOUDNESS_DEFAULT_FLOOR_DB;
}
int omxo_loudness_create(const omx_loudness_config* cfg, omxo_loudness** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    LoudnessConfig c;
    c.sample_rate = cfg->sample_rate;
    c.floor_db = cfg->floor_db;
    *out = new omxo_loudness(c);
    return OMX_NONE;
}
void omxo_loudness_destroy(omxo_loudness* h) { delete h; }
int omxo_loudness_reset_audio(omxo_loudness* h) {
    if (!h) return OMX_ERR_INVALID;
    h->p.reset_audio();
    return OMX_NONE;
}
int omxo_loudness_process_block(omxo_loudness* h, const omx_block* block, omx_loudness_snapshot* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    return h->p.process_block(to_block(block), *out) ? OMX_PRODUCED : OMX_NONE;
}
void omxo_k_weighting_coefficients(double fs, double b[5], double a[5]) {
    const KWeighting w = k_weighting_coefficients(fs);
    for (int i = 0; i < 5; ++i) {
        b[i] = w.b[i];
        a[i] = w.a[i];
    }
}
void omxo_loudness_force_active(omxo_loudness* h, uint32_t channels, float sample_rate) {
    h->p.ensure_state(channels, sample_rate);
    h->p.force_active_for_test();
}
float omxo_kat_true_peak_coefficient(uint64_t j, uint64_t factor) { return true_peak_coefficient(j, factor); }
uint64_t omxo_kat_true_peak_delay_len(double sample_rate) { return TruePeakMeter(sample_rate).delay_len; }
double omxo_kat_channel_weight(uint8_t position) { return channel_weight(position); }
uint64_t omxo_kat_window_length(float sample_rate, float secs) { return window_length(sample_rate, secs); }
float omxo_kat_mean_square_to_lufs(double ms, float floor) { return mean_square_to_lufs(ms, floor); }

// ------------------------------------------------------------------ stereometer
void omxo_stereometer_config_default(omx_stereometer_config* out) { to_c(StereometerConfig(), out); }
int omxo_stereometer_create(const omx_stereometer_config* cfg, omxo_stereometer** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    *out = new omxo_stereometer(from_c(*cfg));
    return OMX_NONE;
}
void omxo_stereometer_destroy(omxo_stereometer* h) { delete h; }
int omxo_stereometer_get_config(const omxo_stereometer* h, omx_stereometer_config* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    to_c(h->p.config(), out);
    return OMX_NONE;
}
int omxo_stereometer_update_config(omxo_stereometer* h, const omx_stereometer_config* cfg) {
    if (!h || !cfg) return OMX_ERR_INVALID;
    h->p.update_config(from_c(*cfg));
    return OMX_NONE;
}
int omxo_stereometer_reset_audio(omxo_stereometer* h) {
    if (!h) return OMX_ERR_INVALID;
    h->p.reset_audio();
    return OMX_NONE;
}
int omxo_stereometer_process_block(omxo_stereometer* h, const omx_block* block, omx_stereometer_snapshot* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    if (!h->p.process_block(to_block(block), h->last)) return OMX_NONE;
    for (int b = 0; b < 4; ++b) {
        h->flat[b].clear();
        for (const auto& pt : h->last.points[b]) {
            h->flat[b].push_back(pt.first);
            h->flat[b].push_back(pt.second);
        }
        out->points[b] = h->flat[b].data();
        out->n_points[b] = h->last.points[b].size();
        out->correlations[b] = h->last.correlations[b];
    }
    return OMX_PRODUCED;
}
// Correlator fed with pairs at a fixed alpha (reference test :218-256).
float omxo_kat_correlation(const float* pairs, uint64_t n_pairs, double alpha) {
    Correlator c;
    for (uint64_t i = 0; i < n_pairs; ++i) c.update(pairs[2 * i], pairs[2 * i + 1], alpha);
    return c.value();
}
double omxo_kat_ema_alpha(float sample_rate, float window) { return ema_alpha(sample_rate, window); }

// ------------------------------------------------------------------ oscilloscope
void omxo_oscilloscope_config_default(omx_oscilloscope_config* out) { to_c(OscilloscopeConfig(), out); }
int omxo_oscilloscope_create(const omx_oscilloscope_config* cfg, omxo_oscilloscope** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    *out = new omxo_oscilloscope(from_c(*cfg));
    return OMX_NONE;
}
void omxo_oscilloscope_destroy(omxo_oscilloscope* h) { delete h; }
int omxo_oscilloscope_get_config(const omxo_oscilloscope* h, omx_oscilloscope_config* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    to_c(h->p.config(), out);
    return OMX_NONE;
}
int omxo_oscilloscope_update_config(omxo_oscilloscope* h, const omx_oscilloscope_config* cfg) {
    if (!h || !cfg) return OMX_ERR_INVALID;
    h->p.update_config(from_c(*cfg));
    return OMX_NONE;
}
int omxo_oscilloscope_reset_audio(omxo_oscilloscope* h) {
    if (!h) return OMX_ERR_INVALID;
    h->p.reset_audio();
    return OMX_NONE;
}
int omxo_oscilloscope_process_block(omxo_oscilloscope* h, const omx_block* block, omx_oscilloscope_snapshot* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    if (!h->p.process_block(to_block(block), h->last)) return OMX_NONE;
    out->epoch = h->last.epoch;
    out->channels = h->last.channels;
    out->slots[0] = h->last.slots[0];
    out->slots[1] = h->last.slots[1];
    out->samples_per_channel = h->last.samples_per_channel;
    out->n_samples = h->last.samples.size();
    out->samples = h->last.samples.data();
    return OMX_PRODUCED;
}
int omxo_oscilloscope_last_cycle_rate(const omxo_oscilloscope* h, float* hz) {
    const auto r = h->p.last_cycle_rate();
    if (!r) return 0;
    *hz = *r;
    return 1;
}
int omxo_oscilloscope_last_capture(const omxo_oscilloscope* h, uint32_t* start, float* frac_offset) {
    const auto& c = h->p.last_capture();
    if (!c) return 0;
    if (start) *start = (uint32_t)c->start;
    if (frac_offset) *frac_offset = c->frac_offset;
    return 1;
}
uint64_t omxo_oscilloscope_trace_len(const omxo_oscilloscope* h, int slot) { return h->p.trace_buffer(slot).size(); }
// StableTrigger::find_best (:441-484) on caller-supplied arrays: work[len + search], template[len].  scores[search + 1] receives
// normalized_correlation (:210-236) at EVERY offset (the search itself evaluates only the ones its walk visits).
int omxo_debug_scope_find_best(const float* work, const float* tmpl, uint32_t len, uint32_t search, float period, uint32_t* best_off,
                               float* frac_offset, float* best_score, float* scores) {
    StableTrigger t;
    t.work.assign(work, work + (size_t)len + search);
    t.candidate.assign(tmpl, tmpl + len);
    const auto best = t.find_best(search, period);
    float stats[2];
    correlation_stats(t.candidate, stats);
    if (best_off) *best_off = (uint32_t)best.first;
    if (frac_offset) *frac_offset = best.second;
    if (best_score) *best_score = normalized_correlation(t.work.data() + best.first, t.candidate.data(), len, stats);
    if (scores)
        for (uint32_t o = 0; o <= search; ++o) scores[o] = normalized_correlation(t.work.data() + o, t.candidate.data(), len, stats);
    return 1;
}

// PeriodEstimator::estimate_period on a bare slice (reference test :957-995)
int omxo_kat_estimate_period(const float* samples, uint64_t n, float rate, float* period, float* confidence) {
    PeriodEstimator e;
    const auto r = e.estimate_period(samples, (size_t)n, rate);
    if (!r) return 0;
    *period = r->period;
    *confidence = r->confidence;
    return 1;
}
// StableTrigger driven block by block over a long signal (reference `stable_phase_jitter`, :933-955):
// for block in 1..n_blocks: capture(signal[start..end]); writes start+capture.start+frac per block and
// whether a period is locked.
void omxo_kat_stable_trigger_positions(const float* signal, uint64_t n, uint64_t block, uint64_t n_blocks, float rate,
                                       float segment_duration, uint64_t cycles, float* positions, uint8_t* locked) {
    StableTrigger trigger;
    const size_t base_frames = f2usize((double)std::round(rate * segment_duration));
    const size_t max_period = f2usize((double)std::ceil(rate / PeriodEstimator::MIN_HZ));
    const size_t probe_frames =
        std::max(f2usize((double)std::round(rate * PeriodEstimator::PROBE_SECONDS)), max_period * 2);
    const size_t history_frames = stable_history_frames(max_period, cycles, rate);
    for (uint64_t b = 1; b < n_blocks; ++b) {
        const size_t end = (size_t)(b * block);
        if (end > n) break;
        const size_t start = end > history_frames ? end - history_frames : 0;
        const Capture c = trigger.capture(signal + start, end - start, rate, probe_frames, base_frames, cycles);
        positions[b] = (float)start + (float)c.start + c.frac_offset;
        locked[b] = trigger.period.has_value() ? 1 : 0;
    }
}
// retune_reference on explicit state (reference test :1021-1042)
void omxo_kat_retune_reference(float* reference, uint64_t len_in, float old_period, float new_period, uint64_t len_out,
                               float* out) {
    StableTrigger t;
    t.reference.assign(reference, reference + len_in);
    t.reference_period = old_period;
    t.retune_reference((size_t)len_out, new_period);
    for (size_t i = 0; i < t.reference.size() && i < len_out; ++i) out[i] = t.reference[i];
}
// prepare_template / write_candidate / find_best on explicit state (reference tests :1044-1081)
void omxo_kat_prepare_template(uint64_t len, float period, float* out) {
    StableTrigger t;
    t.reference.assign((size_t)len, 0.0f);
    t.reference_period = period;
    t.prepare_template(period, false);
    for (size_t i = 0; i < len; ++i) out[i] = t.candidate[i];
}
float omxo_kat_write_candidate(const float* reference, uint64_t ref_len, const float* segment, uint64_t seg_len,
                               float period, float* candidate_out) {
    StableTrigger t;
    t.reference.assign(reference, reference + ref_len);
    const float r = t.write_candidate(segment, (size_t)seg_len, period);
    if (candidate_out)
        for (size_t i = 0; i < t.candidate.size(); ++i) candidate_out[i] = t.candidate[i];
    return r;
}
uint64_t omxo_kat_find_best(const float* candidate, uint64_t cand_len, const float* work, uint64_t work_len, uint64_t search,
                            float period, float* frac) {
    StableTrigger t;
    t.candidate.assign(candidate, candidate + cand_len);
    t.work.assign(work, work + work_len);
    const auto r = t.find_best((size_t)search, period);
    if (frac) *frac = r.second;
    return r.first;
}
int omxo_kat_find_rising_zero_crossing(const float* samples, uint64_t n, uint64_t lo, uint64_t hi, int reversed,
                                       uint64_t* index) {
    const auto r = find_rising_zero_crossing(samples, (size_t)n, (size_t)lo, (size_t)hi, reversed != 0);
    if (!r) return 0;
    *index = *r;
    return 1;
}

// ------------------------------------------------------------------ waveform
struct omxo_waveform {
    WaveformProcessor p;
    WaveformProcessor::Update last;
    explicit omxo_waveform(WaveformConfig c) : p(c) {}
};
static WaveformConfig wf_from_c(const omx_waveform_config& c) {
    WaveformConfig o;
    o.sample_rate = c.sample_rate;
    o.scroll_speed = c.scroll_speed;
    o.max_columns = (size_t)c.max_columns;
    o.analyze_bands = c.analyze_bands != 0;
    o.track_history = c.track_history != 0;
    return o;
}
static void wf_to_c(const WaveformConfig& c, omx_waveform_config* o) {
    o->sample_rate = c.sample_rate;
    o->scroll_speed = c.scroll_speed;
    o->max_columns = c.max_columns;
    o->analyze_bands = c.analyze_bands;
    o->track_history = c.track_history;
}
void omxo_waveform_config_default(omx_waveform_config* out) { wf_to_c(WaveformConfig(), out); }
int omxo_waveform_create(const omx_waveform_config* cfg, omxo_waveform** out) {
    if (!cfg || !out) return OMX_ERR_INVALID;
    *out = new omxo_waveform(wf_from_c(*cfg));
    return OMX_NONE;
}
void omxo_waveform_destroy(omxo_waveform* h) { delete h; }
int omxo_waveform_get_config(const omxo_waveform* h, omx_waveform_config* out) {
    wf_to_c(h->p.config(), out);
    return OMX_NONE;
}
int omxo_waveform_update_config(omxo_waveform* h, const omx_waveform_config* cfg) {
    h->p.update_config(wf_from_c(*cfg));
    return OMX_NONE;
}
int omxo_waveform_reset_audio(omxo_waveform* h) {
    h->p.reset_audio();
    return OMX_NONE;
}
int omxo_waveform_prepare(omxo_waveform* h) {
    h->p.prepare();
    return OMX_NONE;
}
int omxo_waveform_process_block(omxo_waveform* h, const omx_block* block, omx_waveform_update* out) {
    if (!h || !block || !out) return OMX_ERR_INVALID;
    if (!h->p.process_block(to_block(block), h->last)) return OMX_NONE;
    std::memset(out, 0, sizeof(*out));
    out->n_columns = h->last.columns.size() / WF_CHANNELS;
    out->columns = h->last.columns.data();
    out->reset = h->last.reset ? 1 : 0;
    out->preview_some = h->last.preview_some ? 1 : 0;
    out->preview_progress = h->last.preview_progress;
    for (int c = 0; c < WF_CHANNELS; ++c) out->preview[c] = h->last.preview[c];
    return OMX_PRODUCED;
}
int omxo_waveform_has_band_analysis(const omxo_waveform* h) { return h->p.has_band_analysis() ? 1 : 0; }
double omxo_waveform_column_phase(const omxo_waveform* h) { return h->p.column_phase(); }
// ThreeBand<Biquad,false> over mono samples -> out[n][3] (reference test :410-436)
void omxo_kat_threeband_12db(float sample_rate, const float* x, uint64_t n, float* out) {
    BandFilter f(sample_rate, BAND_SPLITS_HZ[0], BAND_SPLITS_HZ[1]);
    for (uint64_t i = 0; i < n; ++i) {
        float in[1] = {x[i]}, bands[3][1];
        f.process(in, bands);
        for (int b = 0; b < 3; ++b) out[3 * i + b] = bands[b][0];
    }
}

// ------------------------------------------------------------------ batcher (meter.rs)
struct omxo_batcher {
    DspBatcher b;
};
static AudioFormat fmt_from_c(const omx_audio_format* f) {
    AudioFormat o;
    o.channels = f->channels;
    o.sample_rate = f->sample_rate;
    o.generation = f->generation;
    for (int i = 0; i < MAX_CH; ++i) o.positions[i] = f->positions[i];
    return o;
}
static void fmt_to_c(const AudioFormat& f, omx_audio_format* o) {
    o->generation = f.generation;
    o->sample_rate = f.sample_rate;
    o->channels = (uint32_t)f.channels;
    for (int i = 0; i < MAX_CH; ++i) o->positions[i] = f.positions[i];
}
int omxo_batcher_create(omxo_batcher** out) {
    *out = new omxo_batcher();
    return OMX_NONE;
}
void omxo_batcher_destroy(omxo_batcher* b) { delete b; }
uint64_t omxo_batcher_push(omxo_batcher* b, const float* samples, uint64_t n, const omx_audio_format* format, omx_ingest_fn ingest,
                           void* user) {
    return b->b.push(samples, (size_t)n, fmt_from_c(format), [&](const float* p, size_t len, const AudioFormat& f) {
        omx_audio_format cf;
        fmt_to_c(f, &cf);
        if (ingest) ingest(user, p, len, &cf);
    });
}
uint64_t omxo_batcher_push_silence(omxo_batcher* b, uint64_t frames, const omx_audio_format* format, omx_ingest_fn ingest,
                                   omx_reset_fn reset, void* user) {
    size_t count = 0;
    const bool ok = b->b.push_silence(frames, fmt_from_c(format), [&](const float* p, size_t len, const AudioFormat& f) {
        omx_audio_format cf;
        fmt_to_c(f, &cf);
        if (ingest) ingest(user, p, len, &cf);
    }, &count);
    if (!ok && reset) reset(user);
    return count;
}
void omxo_batcher_reset(omxo_batcher* b, omx_reset_fn reset, void* user) {
    b->b.clear();
    if (reset) reset(user);
}
void omxo_batcher_clear(omxo_batcher* b) { b->b.clear(); }
uint64_t omxo_batcher_pending(const omxo_batcher* b, float* dst, uint64_t cap) {
    const auto& p = b->b.pending();
    if (dst)
        for (size_t i = 0; i < p.size() && i < cap; ++i) dst[i] = p[i];
    return p.size();
}
int omxo_batcher_format(const omxo_batcher* b, omx_audio_format* out) {
    if (!b->b.format()) return 0;
    if (out) fmt_to_c(*b->b.format(), out);
    return 1;
}

// ------------------------------------------------------------------ primitive KATs
float omxo_kat_power_to_db(float p, float floor) { return power_to_db(p, floor); }
float omxo_kat_db_to_power(float db) { return db_to_power(db); }
float omxo_kat_sanitize_sample_rate(float r) { return sanitize_sample_rate(r); }
void omxo_kat_window(uint32_t kind, uint64_t len, float* out) {
    const auto w = window_coefficients(kind, (size_t)len);
    for (size_t i = 0; i < w.size(); ++i) out[i] = w[i];
}
void omxo_kat_bin_normalization(const float* window, uint64_t wlen, uint64_t fft_size, float* out) {
    const auto n = compute_fft_bin_normalization(std::vector<float>(window, window + wlen), (size_t)fft_size);
    for (size_t i = 0; i < n.size(); ++i) out[i] = n[i];
}
// stereo fold of a block: writes frames*2 floats, returns stereo_channels; matrix -> m[8][2]
uint64_t omxo_kat_stereo_frames(const omx_block* block, float* out_lr, float* matrix) {
    const AudioBlock b = to_block(block);
    const size_t frames = b.frame_count();
    for (size_t f = 0; f < frames; ++f) b.stereo_frame(f, out_lr + 2 * f);
    if (matrix)
        for (int i = 0; i < MAX_CH; ++i) {
            matrix[2 * i] = b.stereo[i][0];
            matrix[2 * i + 1] = b.stereo[i][1];
        }
    return b.stereo_channels;
}
// WindowedMeans<1,W,f64> pushed with `values`; writes mean of each window after the last push
void omxo_kat_windowed_means(const uint64_t* capacities, uint32_t windows, const double* values, uint64_t n, double* means) {
    auto run = [&](auto& wm) {
        for (uint64_t i = 0; i < n; ++i) wm.push({values[i]});
        for (uint32_t w = 0; w < windows; ++w) {
            double m[1];
            wm.mean((int)w, m);
            means[w] = m[0];
        }
    };
    if (windows == 1) {
        size_t caps[1] = {(size_t)capacities[0]};
        WindowedMeans<1, 1> wm(caps);
        run(wm);
    } else {
        size_t caps[4] = {(size_t)capacities[0], (size_t)capacities[1], (size_t)capacities[2], (size_t)capacities[3]};
        WindowedMeans<1, 4> wm(caps);
        run(wm);
    }
}
// Biquad: process `n` samples through a fresh filter, optionally clearing state after `clear_after` samples
void omxo_kat_biquad(int highpass, float sample_rate, float frequency, const float* in, uint64_t n, int64_t clear_after,
                     float* out, float coeffs[5]) {
    Biquad f(highpass ? FilterKind::HighPass : FilterKind::LowPass, sample_rate, frequency);
    if (coeffs) {
        coeffs[0] = f.b[0]; coeffs[1] = f.b[1]; coeffs[2] = f.b[2]; coeffs[3] = f.a[0]; coeffs[4] = f.a[1];
    }
    for (uint64_t i = 0; i < n; ++i) {
        if ((int64_t)i == clear_after) f.clear();
        out[i] = f.process(in[i]);
    }
}
// ThreeBand<[Cascade<Biquad,2>;2],true> over stereo pairs -> out[n][3][2]
void omxo_kat_threeband_lr4(float sample_rate, const float* lr, uint64_t n, float* out) {
    BandSplitter s(sample_rate, BAND_SPLITS_HZ[0], BAND_SPLITS_HZ[1]);
    for (uint64_t i = 0; i < n; ++i) {
        float bands[3][2];
        s.process(lr + 2 * i, bands);
        std::memcpy(out + 6 * i, bands, sizeof(bands));
    }
}
// unnormalised complex FFT in f32 and f64 (interleaved re,im) for the FFT self-check
void omxo_kat_fft_f32(float* data, uint64_t n, int inverse) {
    fft_inplace(reinterpret_cast<std::complex<float>*>(data), (size_t)n, inverse != 0);
}
void omxo_kat_fft_f64(double* data, uint64_t n, int inverse) {
    fft_inplace(reinterpret_cast<std::complex<double>*>(data), (size_t)n, inverse != 0);
}

// ------------------------------------------------------------------ cpu_baseline timing legs (bench.py)
// Runs `n_streams` independent SpectrogramProcessors over [stream][frames][channels] PCM split into
// `block_frames`-frame blocks on `threads` host threads; returns seconds and total columns emitted.
double omxo_bench_spectrogram(const omx_spectrogram_config* cfg, const float* pcm, uint64_t n_streams, uint64_t frames,
                              uint32_t channels, uint64_t block_frames, uint32_t threads, uint64_t* columns_out) {
    // The timed region starts when EVERY thread has built its first processor (windows, derivative window by the spectral method,
    // twiddle tables: tens of milliseconds each) and touched its PCM, and ends when the last one is through — until round 5 thread
    // start-up, table construction and first-touch page faults sat inside a region of ~0.25 s of work per thread, and 256 threads
    // measured 7x one.
    const uint32_t T = threads ? threads : 1;
    std::vector<uint64_t> cols(T, 0);
    std::atomic<uint32_t> ready{0};
    std::atomic<bool> go{false};
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < T; ++t) {
        pool.emplace_back([&, t]() {
            std::unique_ptr<SpectrogramProcessor> first;
            if (t < n_streams) {
                first.reset(new SpectrogramProcessor(from_c(*cfg)));
                first->prepare();
                volatile float sink = 0.0f;
                for (uint64_t s = t; s < n_streams; s += T)   // first touch of this thread's PCM rows
                    for (uint64_t i = 0; i < frames * channels; i += 1024) sink = sink + pcm[s * frames * channels + i];
            }
            ready.fetch_add(1);
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            for (uint64_t s = t; s < n_streams; s += T) {
                std::unique_ptr<SpectrogramProcessor> own;
                SpectrogramProcessor* p = first.get();
                if (s != t) {
                    own.reset(new SpectrogramProcessor(from_c(*cfg)));
                    p = own.get();
                }
                const float* base = pcm + s * frames * channels;
                SpectrogramUpdate u;
                for (uint64_t off = 0; off < frames; off += block_frames) {
                    const uint64_t nf = std::min(block_frames, frames - off);
                    const AudioBlock b = AudioBlock::make(base + off * channels, (size_t)(nf * channels), channels, cfg->sample_rate);
                    if (p->process_block(b, u)) cols[t] += u.new_columns.size();
                }
            }
        });
    }
    while (ready.load() < T) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto& th : pool) th.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t total = 0;
    for (auto c : cols) total += c;
    if (columns_out) *columns_out = total;
    return secs;
}

// ---- state-side summary reductions (summary.hpp) ----
int omxo_spectrum_peaks(const float* bins, const float* db, int, uint64_t n_bins, uint64_t n_rows, uint64_t row_stride,
                        float min_f, float max_f, void*, omx_spectrum_peak* out) {
    if (!bins || !db || !out) return OMX_ERR_INVALID;
    for (uint64_t r = 0; r < n_rows; ++r) out[r] = spectrum_peak(bins, db + r * row_stride, (size_t)n_bins, min_f, max_f);
    return OMX_PRODUCED;
}
int omxo_peak_holds_reset(omx_peak_hold* holds, int, uint64_t n, double now, void*) {
    if (!holds) return OMX_ERR_INVALID;
    for (uint64_t i = 0; i < n; ++i) holds[i] = omx_peak_hold{kMeterDbLo, 0, now};
    return OMX_NONE;
}
int omxo_loudness_meters(const omx_loudness_snapshot* snapshots, int, uint64_t n_streams, uint64_t n_blocks, uint32_t left_mode,
                         uint32_t right_mode, double t0, double dt, omx_peak_hold* holds, void*, omx_meter_row* rows) {
    if (!snapshots || !holds || !rows || left_mode > OMX_METER_TRUE_PEAK || right_mode > OMX_METER_TRUE_PEAK) return OMX_ERR_INVALID;
    for (uint64_t s = 0; s < n_streams; ++s)
        for (uint64_t k = 0; k < n_blocks; ++k)
            rows[s * n_blocks + k] = apply_meter_snapshot(snapshots[s * n_blocks + k], left_mode, right_mode, t0 + (double)k * dt, holds + 3 * s);
    return OMX_PRODUCED;
}

// ---- column history ring (splat.hpp: SpectrogramHistoryRing) ----
struct omxo_spectrogram_history {
    SpectrogramHistoryRing ring;
};
int omxo_spectrogram_history_create(uint32_t n_streams, omxo_spectrogram_history** out) {
    if (!out || n_streams != 1) return OMX_ERR_INVALID;  // the checker models one stream
    *out = new omxo_spectrogram_history();
    return OMX_NONE;
}
void omxo_spectrogram_history_destroy(omxo_spectrogram_history* h) { delete h; }
int omxo_spectrogram_history_apply(omxo_spectrogram_history* h, const omx_spectrogram_update* update) {
    if (!h || !update) return OMX_ERR_INVALID;
    h->ring.apply_update(*update);
    return OMX_NONE;
}
int omxo_spectrogram_history_get_info(omxo_spectrogram_history* h, omx_spectrogram_history_info* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    const SpectrogramHistoryRing& r = h->ring;
    *out = omx_spectrogram_history_info{r.col_kind, r.ring_capacity, r.write_slot, r.col_count, r.points_per_column,
                                        r.reassigned_points_per_slot, r.newest_slot(), r.visible_slots()};
    return OMX_NONE;
}
int64_t omxo_spectrogram_history_slot_counts(omxo_spectrogram_history* h, uint64_t stream_index, uint32_t* out, uint64_t capacity) {
    if (!h || stream_index != 0) return OMX_ERR_INVALID;
    for (uint64_t i = 0; i < capacity && i < h->ring.slot_counts.size(); ++i) out[i] = h->ring.slot_counts[i];
    return (int64_t)h->ring.ring_capacity;
}
int omxo_spectrogram_history_fetch_slot(omxo_spectrogram_history* h, uint64_t stream_index, uint32_t slot, void* dst, uint64_t cap,
                                        uint64_t* n_out) {
    if (!h || stream_index != 0 || slot >= h->ring.ring_capacity || !dst) return OMX_ERR_INVALID;
    const SpectrogramHistoryRing& r = h->ring;
    const uint32_t ppc = r.points_per_column;
    if (r.col_kind == OMX_COLUMN_REASSIGNED) {
        const uint64_t n = std::min<uint64_t>(r.slot_counts[slot], ppc);
        std::memcpy(dst, r.points.data() + (size_t)slot * ppc, (size_t)std::min(n, cap) * sizeof(omx_spectrogram_point));
        if (n_out) *n_out = n;
    } else {
        std::memcpy(dst, r.codes.data() + (size_t)slot * ppc, (size_t)std::min<uint64_t>(ppc, cap) * sizeof(uint16_t));
        if (n_out) *n_out = ppc;
    }
    return OMX_NONE;
}
int omxo_spectrogram_history_splat(omxo_spectrogram_history* h, float reassigned_power_scale, const omx_splat_view* view, int, void*,
                                   float* accum, float* db) {
    if (!h || !view || !accum || view->width == 0 || view->height == 0) return OMX_ERR_INVALID;
    h->ring.splat(reassigned_power_scale, *view, accum, db);
    return OMX_PRODUCED;
}

// ---- reassigned-splat accumulation + resolve (splat.hpp) ----
void omxo_splat_view_size(omx_splat_view* view) {
    if (view) splat_view_size(view);
}
int omxo_spectrogram_splat(const omx_spectrogram_point* points, const uint32_t* counts, int, uint64_t n_streams, uint64_t n_columns,
                           uint64_t column_stride, float reassigned_power_scale, const omx_splat_view* view, void*, float* accum,
                           float* db) {
    if (!points || !counts || !view || !accum || view->width == 0 || view->height == 0 || !(view->scale_factor >= 1.0f))
        return OMX_ERR_INVALID;
    spectrogram_splat(points, counts, n_streams, n_columns, column_stride, reassigned_power_scale, *view, accum, db);
    return OMX_PRODUCED;
}

}  // extern "C"
