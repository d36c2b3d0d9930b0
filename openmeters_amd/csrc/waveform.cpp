// Host side of the waveform path: reference src/visuals/waveform/processor.rs:31-52 (config normalisation), :147-211
// (lifecycle), :308-352 (process_block / update_config).  The fractional column phase is advanced on the host with the
// reference's exact f64 add / compare / subtract sequence so the number of emitted columns is known before launch.
#include <algorithm>
#include <map>
#include <tuple>

#include "waveform.hpp"

namespace omx {

constexpr size_t kWfMaxColumns = 8192;        // :11
constexpr float kWfDefaultScroll = 300.0f;    // :13
constexpr float kWfMinRuntimeScroll = 1.0f;   // :15
constexpr float kBandSplits[2] = {200.0f, 2000.0f};

void waveform_config_default(omx_waveform_config* c) {  // :31-40
    c->sample_rate = kDefaultSampleRate;
    c->scroll_speed = kWfDefaultScroll;
    c->max_columns = kWfMaxColumns;
    c->analyze_bands = 1;
    c->track_history = 0;
}
static omx_waveform_config normalized(omx_waveform_config c) {  // :42-52
    c.sample_rate = sanitize_sample_rate(c.sample_rate);
    c.scroll_speed = (std::isfinite(c.scroll_speed) && c.scroll_speed > 0.0f) ? std::fmax(c.scroll_speed, kWfMinRuntimeScroll) : kWfDefaultScroll;
    c.max_columns = std::min<uint64_t>(std::max<uint64_t>(c.max_columns, 1), kWfMaxColumns);
    c.analyze_bands = c.analyze_bands ? 1 : 0;
    c.track_history = (c.track_history && c.analyze_bands) ? 1 : 0;
    return c;
}
static uint32_t window_len(size_t at_reference_rate, float sample_rate) {  // :78-82
    sample_rate = std::fmin(sample_rate, 1000000.0f);
    return (uint32_t)std::max<size_t>(f2usize((double)std::round((float)at_reference_rate * sample_rate / 44100.0f)), 1);
}

WaveformBank::WaveformBank(const omx_waveform_config& cfg, uint32_t n_streams) : n_streams_(n_streams) {
    cfg_ = normalized(cfg);
    state_.reserve((size_t)n_streams_ * 16);
}

void WaveformBank::rebuild() {  // :175-184
    column_phase_ = 0.0;
    ragged_zero_phase_ = true;
    clear_minmax_ = true;
    if (analysis_) reset_trackers();
    reset_pending_ = true;
}
void WaveformBank::reset_trackers() {  // :199-201 (band_analysis is None when analyze_bands is off)
    analysis_ = cfg_.analyze_bands != 0;
    clear_trackers_ = true;
    pushes_ = 0;
    ragged_zero_pushes_ = true;
    kept_invalidate();
}
void WaveformBank::kept_invalidate() {
    kept_.clear();
    kept_free_.clear();
    for (uint32_t k = kept_slots_; k-- > 0;) kept_free_.push_back(k);
    kept_zero_void_ = true;
}
// room for `slots` totals (contents preserved); false when the table would pass 1 GiB — the call then leaves no totals and later
// calls read the rings
bool WaveformBank::kept_reserve(size_t slots, hipStream_t stream) {
    if (slots <= kept_slots_) return true;
    const size_t per_slot = (size_t)n_streams_ * 24 * 2;
    const size_t most = ((size_t)1 << 30) / (per_slot * sizeof(double));
    if (slots > most) return false;
    const size_t grown = std::min(most, std::max<size_t>(slots * 2, 64));
    DeviceBuffer<double> bigger;
    bigger.reserve(grown * per_slot);
    if (kept_slots_) {
        OMX_HIP(hipMemcpyAsync(bigger.ptr, kept_totals_.ptr, (size_t)kept_slots_ * per_slot * sizeof(double), hipMemcpyDeviceToDevice, stream));
        OMX_HIP(hipStreamSynchronize(stream));
    }
    std::swap(bigger.ptr, kept_totals_.ptr);
    std::swap(bigger.count, kept_totals_.count);
    for (uint32_t k = (uint32_t)grown; k-- > kept_slots_;) kept_free_.push_back(k);
    kept_slots_ = (uint32_t)grown;
    return true;
}
void WaveformBank::prepare(hipStream_t) {  // :169-173
    if (cfg_.analyze_bands && !analysis_) reset_trackers();
}
void WaveformBank::update_config(const omx_waveform_config& in) {  // :336-352
    const omx_waveform_config n = normalized(in);
    const bool rebuild_all = cfg_.sample_rate != n.sample_rate;
    const bool reset_analysis = cfg_.analyze_bands != n.analyze_bands || cfg_.track_history != n.track_history;
    cfg_ = n;
    if (rebuild_all) rebuild();
    else if (reset_analysis && analysis_) reset_trackers();
}

int WaveformBank::process(const float* pcm, bool pcm_on_device, uint64_t frames, uint32_t channels_in, float sample_rate_in,
                          const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_waveform_bank_update* out) {  // :308-334
    if (ragged_) {
        set_last_error("waveform bank: per-stream positions are in use (process_ragged); reset_audio() returns the bank to lock-step calls");
        return OMX_ERR_INVALID;
    }
    return process_impl(pcm, pcm_on_device, frames, channels_in, sample_rate_in, positions, stream, out, nullptr);
}

int WaveformBank::process_ragged(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask,
                                 uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                 omx_waveform_ragged_update* out) {
    bool any = false;
    for (uint32_t s = 0; s < n_streams_; ++s) {
        if (frames[s] > frames_capacity) {
            set_last_error("waveform process_ragged: frames[s] > frames_capacity");
            return OMX_ERR_INVALID;
        }
        any = any || frames[s] != 0 || (reset_mask && reset_mask[s]);
    }
    last_stream_ = stream;
    if (!any || frames_capacity == 0) return OMX_NONE;
    const RaggedCall rc{frames, reset_mask, out};
    return process_impl(d_pcm, true, frames_capacity, channels, sample_rate, positions, stream, nullptr, &rc);
}

int WaveformBank::process_impl(const float* pcm, bool pcm_on_device, uint64_t frames, uint32_t channels_in, float sample_rate_in,
                               const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_waveform_bank_update* out,
                               const RaggedCall* ragged) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    last_stream_ = stream;
    if (frames == 0) return OMX_NONE;
    const float sample_rate = sanitize_sample_rate(sample_rate_in);
    if (channels != source_channels_ || cfg_.sample_rate != sample_rate) {
        source_channels_ = channels;
        cfg_.sample_rate = sample_rate;
        rebuild();
    }
    prepare(stream);
    const uint32_t color_len = window_len(2048, cfg_.sample_rate), slow_len = window_len(16384, cfg_.sample_rate);
    if (color_len != color_len_ || slow_len != slow_len_ || !color_ring_.ptr) {
        color_len_ = color_len;
        slow_len_ = slow_len;
        color_ring_.reserve((size_t)color_len * n_streams_ * 16);
        hist_ring_.reserve((size_t)slow_len * n_streams_ * 16);
        clear_trackers_ = true;
        pushes_ = 0;
        ragged_zero_pushes_ = true;
        kept_invalidate();
    }
    if (clear_minmax_ && clear_trackers_) {
        OMX_HIP(hipMemsetAsync(state_.ptr, 0, state_.count * sizeof(WaveLaneState), stream));
    } else if (clear_minmax_ || clear_trackers_) {
        // partial clears are rare (config toggles): do them on the host side of the struct
        std::vector<WaveLaneState> h(state_.count);
        OMX_HIP(hipMemcpyAsync(h.data(), state_.ptr, h.size() * sizeof(WaveLaneState), hipMemcpyDeviceToHost, stream));
        OMX_HIP(hipStreamSynchronize(stream));
        for (auto& st : h) {
            if (clear_trackers_) {
                std::memset(st.za, 0, sizeof(st.za));
                std::memset(st.zb, 0, sizeof(st.zb));
                std::memset(st.color, 0, sizeof(st.color));
                std::memset(st.hist, 0, sizeof(st.hist));
            }
            if (clear_minmax_) {
                st.cur_min = st.cur_max = st.cur_last = st.last_sample = 0.0f;
                st.cur_some = st.cur_has_last = st.last_valid = 0;
            }
        }
        OMX_HIP(hipMemcpyAsync(state_.ptr, h.data(), h.size() * sizeof(WaveLaneState), hipMemcpyHostToDevice, stream));
        OMX_HIP(hipStreamSynchronize(stream));
    }
    if (clear_trackers_) {
        OMX_HIP(hipMemsetAsync(color_ring_.ptr, 0, color_ring_.count * sizeof(float), stream));
        OMX_HIP(hipMemsetAsync(hist_ring_.ptr, 0, hist_ring_.count * sizeof(float), stream));
    }
    clear_minmax_ = clear_trackers_ = false;

    const double step = std::min(std::max((double)cfg_.scroll_speed / (double)cfg_.sample_rate, 0.0), 1.0);
    if (ragged) {
        kept_invalidate();  // (per-stream push counts: the kept totals are a lock-step bank's)
        // every stream advances its own phase on the device; a call emits at most floor(phase + frames x step) <= max_cols columns
        const uint64_t max_cols = (uint64_t)std::floor(1.0 + (double)frames * step) + 1;
        if (max_cols > cfg_.max_columns)  // cap_pending_columns (:293-298) would drop the oldest ones: not modelled per stream
            unsupported("waveform process_ragged: frames_capacity x scroll_speed / sample_rate exceeds max_columns");
        if (!ragged_) {  // every stream starts from the bank's common push count and column phase
            r_pushes_.upload(std::vector<uint64_t>(n_streams_, pushes_), stream);
            r_phase_.upload(std::vector<double>(n_streams_, column_phase_), stream);
            h_pushes_.assign(n_streams_, pushes_);   // the host's mirror of the two (run_chunked_ragged)
            h_phase_.assign(n_streams_, column_phase_);
            mirror_valid_ = true;
            ragged_ = true;
            ragged_zero_phase_ = ragged_zero_pushes_ = false;
        }
        if (ragged_zero_phase_) {
            OMX_HIP(hipMemsetAsync(r_phase_.ptr, 0, n_streams_ * sizeof(double), stream));
            std::fill(h_phase_.begin(), h_phase_.end(), 0.0);
        }
        if (ragged_zero_pushes_) {
            OMX_HIP(hipMemsetAsync(r_pushes_.ptr, 0, n_streams_ * sizeof(uint64_t), stream));
            std::fill(h_pushes_.begin(), h_pushes_.end(), 0ull);
        }
        if (ragged_zero_phase_ && ragged_zero_pushes_) mirror_valid_ = true;  // (a rebuild makes every stream equal again)
        ragged_zero_phase_ = ragged_zero_pushes_ = false;
        r_cols_.reserve(n_streams_);
        r_progress_.reserve(n_streams_);
        r_staging_.upload(ragged->frames, ragged->reset_mask, n_streams_, r_frames_, r_mask_, stream);
        columns_.reserve((size_t)(n_streams_ * max_cols * 4), false);
        preview_.reserve((size_t)n_streams_ * 4, false);
        WaveformArgs wa{};
        wa.pcm = pcm;
        wa.frames = frames;  // the row stride of pcm
        wa.n_streams = n_streams_;
        wa.fmt = make_format(channels, positions);
        wa.analyze = analysis_ ? 1 : 0;
        wa.track_history = (analysis_ && cfg_.track_history) ? 1 : 0;
        wa.lp_lo = make_biquad(false, cfg_.sample_rate, kBandSplits[0]);
        wa.hp_lo = make_biquad(true, cfg_.sample_rate, kBandSplits[0]);
        wa.lp_hi = make_biquad(false, cfg_.sample_rate, kBandSplits[1]);
        wa.hp_hi = make_biquad(true, cfg_.sample_rate, kBandSplits[1]);
        wa.step = step;
        wa.color_len = color_len_;
        wa.slow_len = slow_len_;
        wa.color_ring = color_ring_.ptr;
        wa.hist_ring = hist_ring_.ptr;
        wa.state = state_.ptr;
        wa.columns = columns_.ptr;
        wa.preview = preview_.ptr;
        wa.frames_v = r_frames_.ptr;
        wa.reset_v = r_mask_.ptr;
        wa.pushes_v = r_pushes_.ptr;
        wa.phase_v = r_phase_.ptr;
        wa.cols_v = r_cols_.ptr;
        wa.progress_v = r_progress_.ptr;
        wa.max_cols = max_cols;
        last_form_ = 1;
        if (channels == 2 && run_chunked_ragged(wa, ragged->frames, ragged->reset_mask, max_cols, step, stream)) last_form_ = 2;
        launch_waveform(wa, stream);
        OMX_HIP(hipGetLastError());
        last_cols_ = max_cols;
        if (ragged->out) {
            ragged->out->n_streams = n_streams_;
            ragged->out->max_columns = max_cols;
            ragged->out->d_n_columns = r_cols_.ptr;
            ragged->out->d_columns = columns_.ptr;
            ragged->out->d_preview = preview_.ptr;
            ragged->out->d_preview_progress = r_progress_.ptr;
            ragged->out->d_reset = r_mask_.ptr;
        }
        reset_pending_ = false;
        return OMX_PRODUCED;
    }
    // fractional column phase (:253-254, :287-291), exactly as the reference accumulates it
    double phase = column_phase_;
    uint64_t n_emit = 0;
    std::vector<uint32_t> column_ends;  // the frames columns end at (the chunk-parallel form's plan)
    const bool record_ends = frames <= 0x7FFFFFFFull;
    for (uint64_t f = 0; f < frames; ++f) {
        phase += step;
        if (phase >= 1.0) {
            ++n_emit;
            phase -= 1.0;
            if (record_ends) column_ends.push_back((uint32_t)f);
        }
    }
    const uint64_t kept = std::min<uint64_t>(n_emit, cfg_.max_columns);  // cap_pending_columns (:293-298)
    columns_.reserve((size_t)(n_streams_ * std::max<uint64_t>(kept, 1) * 4), host_outputs_ && n_streams_ * kept <= 16384);
    preview_.reserve((size_t)n_streams_ * 4, host_outputs_ && n_streams_ <= 64);

    const float* d_pcm = pcm;
    if (!pcm_on_device) {
        const size_t n = (size_t)n_streams_ * frames * channels;
        d_pcm = staging_.stage(pcm, n, stream);
    }
    const float progress = (float)std::min(std::max(phase, 0.0), 1.0);  // preview (:300-306)
    WaveformArgs wa{};
    wa.pcm = d_pcm;
    wa.frames = frames;
    wa.n_streams = n_streams_;
    wa.fmt = make_format(channels, positions);
    wa.analyze = analysis_ ? 1 : 0;
    wa.track_history = (analysis_ && cfg_.track_history) ? 1 : 0;
    wa.lp_lo = make_biquad(false, cfg_.sample_rate, kBandSplits[0]);
    wa.hp_lo = make_biquad(true, cfg_.sample_rate, kBandSplits[0]);
    wa.lp_hi = make_biquad(false, cfg_.sample_rate, kBandSplits[1]);
    wa.hp_hi = make_biquad(true, cfg_.sample_rate, kBandSplits[1]);
    wa.step = step;
    wa.column_phase = column_phase_;
    wa.pushes = pushes_;
    wa.color_len = color_len_;
    wa.slow_len = slow_len_;
    wa.color_ring = color_ring_.ptr;
    wa.hist_ring = hist_ring_.ptr;
    wa.state = state_.ptr;
    wa.n_emit = n_emit;
    wa.first_kept = n_emit - kept;
    wa.columns = columns_.ptr;
    wa.preview = preview_.ptr;
    wa.write_preview = progress > 0.0f ? 1 : 0;
    // chunk-parallel evaluation for bank-sized calls (waveform_chunked.hip); the sequential kernels — bit-identical to the reference's
    // order — serve everything else and are the fallback when the chunk form meets non-finite input
    last_form_ = 1;
    if (record_ends && channels == 2 && run_chunked(wa, column_ends, phase, stream)) last_form_ = 2;
    if (last_form_ != 2) kept_invalidate();  // the sequential kernels do not keep totals
    launch_waveform(wa, stream);
    OMX_HIP(hipGetLastError());
    column_phase_ = phase;
    if (analysis_) pushes_ += frames;
    last_cols_ = kept;
    if (out) {
        out->n_streams = n_streams_;
        out->n_columns = kept;
        out->d_columns = columns_.ptr;
        out->d_preview = preview_.ptr;
        out->reset = reset_pending_ ? 1 : 0;
        out->preview_some = wa.write_preview;
        out->preview_progress = progress;
        out->_pad = 0;
    }
    reset_pending_ = false;
    return OMX_PRODUCED;
}

// Zero-input transition of the band filters over `frames` frames and its powers 2, 4 ... 32, in f64 from the f32 coefficients: column m
// of T is the state after `frames` steps of Biquad::process (dsp.rs:422-432) with x = 0 started from the unit state e_m.  State
// order: low [LP z0, z1]; mid [HP_low z0, z1, LP_high z0, z1]; high [HP_high z0, z1].  Layout [band][power][4][4].
static std::vector<double> wave_transitions(const BiquadCoef& lp_lo, const BiquadCoef& hp_lo, const BiquadCoef& lp_hi, const BiquadCoef& hp_hi,
                                            uint32_t frames) {
    std::vector<double> T(3 * 6 * 16, 0.0);
    const BiquadCoef* first[3] = {&lp_lo, &hp_lo, &hp_hi};
    const BiquadCoef* second[3] = {nullptr, &lp_hi, nullptr};
    for (int band = 0; band < 3; ++band) {
        const int n = second[band] ? 4 : 2;
        for (int m = 0; m < n; ++m) {
            double z[4] = {0, 0, 0, 0};
            z[m] = 1.0;
            for (uint32_t f = 0; f < frames; ++f) {
                double x = 0.0;
                for (int e = 0; e < n / 2; ++e) {
                    const BiquadCoef& c = e == 0 ? *first[band] : *second[band];
                    const double out = (double)c.b[0] * x + z[2 * e];
                    z[2 * e] = (double)c.b[1] * x - (double)c.a[0] * out + z[2 * e + 1];
                    z[2 * e + 1] = (double)c.b[2] * x - (double)c.a[1] * out;
                    x = out;
                }
            }
            for (int k = 0; k < n; ++k) T[(size_t)band * 96 + (size_t)k * 4 + m] = z[k];
        }
        for (int p = 1; p < 6; ++p) {  // repeated squaring (the scan's doubling steps)
            const double* prev = T.data() + (size_t)band * 96 + (size_t)(p - 1) * 16;
            double* next = T.data() + (size_t)band * 96 + (size_t)p * 16;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    long double acc = 0.0L;
                    for (int k = 0; k < 4; ++k) acc += (long double)prev[i * 4 + k] * (long double)prev[k * 4 + j];
                    next[i * 4 + j] = (double)acc;
                }
        }
    }
    return T;
}

// ---- the chunk-parallel form (waveform_chunked.hip).  A GROUP is a set of streams that move in lock step through a call: the same frame
// count, tracker push count and column phase, hence the same plan (cuts, segments, column table).  A lock-step call is one group
// of all streams; a ragged call has one group per distinct (frames, pushes, phase, reset) among its streams.
struct WaveformBank::ChunkGroup {
    uint64_t frames = 0, pushes0 = 0;
    std::vector<uint32_t> column_ends;   // every column the call emits
    uint64_t first_kept = 0;             // columns before this one are dropped (cap_pending_columns, lock-step calls)
    std::vector<uint32_t> streams;       // bank indices (empty: all streams, identity)
    uint32_t write_preview = 0;
    // kept totals (lock-step calls): the table to take old segments from, and the cuts of this call to leave totals at
    const std::map<uint64_t, uint32_t>* kept = nullptr;
    std::vector<int32_t> keep_cuts;
    std::vector<uint32_t> old_slot;  // (planner) [n_old + 1]
    // filled by the planner
    std::vector<int32_t> cuts;
    std::vector<uint32_t> chunk_seg;
    std::vector<WaveEval> evals;
    uint32_t n_segs = 0, n_old = 0, C = 0, n_chunks = 0;
};

// cuts, segments and the column table of one group; false when the shape is not served
static bool plan_chunk_group(WaveformBank::ChunkGroup& g, uint32_t C, bool history, uint32_t color_len, uint32_t slow_len) {
    const uint64_t frames = g.frames;
    const uint32_t n_chunks = (uint32_t)((frames + C - 1) / C);
    const int64_t F = (int64_t)frames;
    const uint64_t P0 = g.pushes0, Pend = P0 + frames;
    const uint32_t nwin = history ? 3u : 1u;
    const int64_t caps[3] = {(int64_t)color_len, (int64_t)color_len, (int64_t)slow_len};
    const uint64_t ring_len[3] = {color_len, slow_len, slow_len};  // buffer length of the WindowedMeans the window lives in
    const int64_t maxcap = history ? (int64_t)slow_len : (int64_t)color_len;
    const uint64_t n_emit = g.column_ends.size(), first_kept = g.first_kept, kept = n_emit - first_kept;

    std::vector<int32_t>& cuts = g.cuts;
    cuts.clear();
    cuts.reserve(n_emit + (size_t)(kept + 1) * nwin + n_chunks + (size_t)(maxcap / 256) + 16);
    cuts.push_back(-1);
    if (!g.kept) {
        cuts.push_back((int32_t)-maxcap);
        for (int64_t q = -1 - 256; q > -maxcap; q -= 256) cuts.push_back((int32_t)q);  // a grid over the rings' contents (parallelism of the old sums)
    }
    for (int32_t c : g.keep_cuts) cuts.push_back(c);
    for (uint32_t c = 1; c <= n_chunks; ++c) cuts.push_back((int32_t)(std::min<int64_t>((int64_t)c * C, F) - 1));
    for (uint32_t f : g.column_ends) cuts.push_back((int32_t)f);
    auto add_windows = [&](int64_t f) {
        for (uint32_t w = 0; w < nwin; ++w) cuts.push_back((int32_t)(f - caps[w]));
    };
    for (uint64_t q = first_kept; q < n_emit; ++q) add_windows((int64_t)g.column_ends[q]);
    add_windows(F - 1);
    int64_t refresh_cut[3] = {0, 0, 0};
    for (uint32_t w = 0; w < nwin; ++w) {  // refresh_counts (dsp.rs:346-352): the pair restarts at every multiple of the capacity
        const uint64_t r = Pend / (uint64_t)caps[w] * (uint64_t)caps[w];
        refresh_cut[w] = std::max<int64_t>((int64_t)r - (int64_t)P0 - 1, -maxcap);
        cuts.push_back((int32_t)refresh_cut[w]);
    }
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    g.old_slot.clear();
    if (g.kept) {
        // the total at a cut = the total of `pushes covered` values (a window that starts before the stream's first push starts at zero)
        auto slot_of = [&](int32_t cut) {
            const int64_t covered = std::max<int64_t>((int64_t)P0 + (int64_t)cut + 1, 0);
            const auto it = g.kept->find((uint64_t)covered);
            return it == g.kept->end() ? kWaveNoSlot : it->second;
        };
        const size_t n_old_cuts = (size_t)(std::lower_bound(cuts.begin(), cuts.end(), -1) - cuts.begin()) + 1;
        std::vector<int32_t> grid;  // old stretches without kept totals at both ends are read from the rings: a grid for their parallelism
        for (size_t j = 0; j + 1 < n_old_cuts; ++j)
            if (slot_of(cuts[j]) == kWaveNoSlot || slot_of(cuts[j + 1]) == kWaveNoSlot)
                for (int64_t q = (int64_t)cuts[j] + 256; q < (int64_t)cuts[j + 1]; q += 256) grid.push_back((int32_t)q);
        if (!grid.empty()) {
            cuts.insert(cuts.end(), grid.begin(), grid.end());
            std::sort(cuts.begin(), cuts.end());
        }
        for (size_t j = 0; j < cuts.size() && cuts[j] <= -1; ++j) g.old_slot.push_back(slot_of(cuts[j]));
    }
    g.n_segs = (uint32_t)cuts.size() - 1;
    if (g.n_segs > 4096) return false;  // (thousands of columns per call: the plan's scratch grows with segments x streams)
    auto index_of = [&](int64_t cut) { return (uint32_t)(std::lower_bound(cuts.begin(), cuts.end(), (int32_t)cut) - cuts.begin()); };
    const uint32_t n_old = index_of(-1);
    g.n_old = n_old;
    g.C = C;
    g.n_chunks = n_chunks;
    g.chunk_seg.resize(n_chunks);
    for (uint32_t c = 0; c < n_chunks; ++c) g.chunk_seg[c] = index_of((int64_t)c * C - 1);
    g.evals.assign((size_t)kept + 1, WaveEval{});
    auto fill = [&](WaveEval& ev, int64_t f) {
        ev.idx_end = index_of(f);
        for (uint32_t w = 0; w < 3; ++w) {
            const uint32_t ww = w < nwin ? w : 0u;
            ev.idx_start[w] = index_of(f - caps[ww]);
            ev.idx_refresh[w] = ev.idx_end;
            const uint64_t pushes = P0 + (uint64_t)f + 1;
            ev.count[w] = (uint32_t)std::max<uint64_t>(std::min(std::min(pushes, ring_len[ww]), (uint64_t)caps[ww]), 1);
        }
    };
    for (uint64_t i = 0; i < kept; ++i) {
        const uint64_t q = first_kept + i;
        WaveEval& ev = g.evals[(size_t)i];
        fill(ev, (int64_t)g.column_ends[q]);
        ev.mm_from = (q == 0 ? n_old : index_of((int64_t)g.column_ends[q - 1])) - n_old;
        ev.mm_to = ev.idx_end - n_old;
        ev.out = (uint32_t)i;
        ev.carry = q == 0 ? 1u : 0u;
    }
    {
        WaveEval& ev = g.evals[(size_t)kept];
        fill(ev, F - 1);
        for (uint32_t w = 0; w < nwin; ++w) ev.idx_refresh[w] = index_of(refresh_cut[w]);
        ev.mm_from = (n_emit == 0 ? n_old : index_of((int64_t)g.column_ends[n_emit - 1])) - n_old;
        ev.mm_to = ev.idx_end - n_old;
        ev.out = 0xFFFFFFFFu;
        ev.carry = n_emit == 0 ? 1u : 0u;
    }
    cuts.push_back(INT32_MAX);  // (pass B looks up the cut after the one it has just reached: one entry behind the last, never reached)
    return true;
}

static bool chunk_shape_ok(uint64_t frames) { return frames % 2 == 0 && frames >= 1024 && frames <= 0x3FFFFFFFull; }

// Plans every group, uploads the plans as one blob and enqueues the kernels: phase 1 (scratch only, may raise `bad`) of every
// group, then phase 2 of every group.  False (nothing enqueued) when a group's shape is not served.
bool WaveformBank::launch_chunk_groups(const WaveformArgs& wa, std::vector<ChunkGroup>& groups, uint64_t pcm_stride, uint64_t col_stride,
                                       hipStream_t stream) {
    const bool history = wa.track_history != 0;
    auto pad16 = [](size_t n) { return (n + 15) / 16 * 16; };
    size_t blob_bytes = 0;
    uint64_t state_floats = 0, segs_room = 0, local_max = 0;
    uint32_t C = 0;
    // chunk length (one per call: one transition table): enough (chunk, 64 streams) workgroups of three wavefronts to fill the SIMDs
    // several times over — 131072 (stream, chunk) items = 2048 workgroups
    static const uint64_t want_items = [] {
        const char* e = tuning_env("OMX_WAVE_CHUNK_ITEMS");  // tuning hook
        return e ? (uint64_t)std::atoll(e) : 131072ull;
    }();
    C = 256;
    auto items = [&](uint32_t c) {
        uint64_t n = 0;
        for (const ChunkGroup& g : groups) n += (uint64_t)(g.streams.empty() ? n_streams_ : g.streams.size()) * ((g.frames + c - 1) / c);
        return n;
    };
    while (C > 64 && items(C) < want_items) C /= 2;
    for (ChunkGroup& g : groups)
        if (!plan_chunk_group(g, C, history, color_len_, slow_len_)) return false;
    for (ChunkGroup& g : groups) {
        const uint32_t n_local = g.streams.empty() ? n_streams_ : (uint32_t)g.streams.size();
        blob_bytes += pad16(g.cuts.size() * sizeof(int32_t)) + pad16(g.chunk_seg.size() * sizeof(uint32_t)) + pad16(g.evals.size() * sizeof(WaveEval)) +
                      pad16(g.streams.size() * sizeof(uint32_t)) + pad16(g.old_slot.size() * sizeof(uint32_t)) +
                      pad16(g.keep_cuts.size() * 2 * sizeof(uint32_t));
        state_floats += (uint64_t)g.n_chunks * n_local * 3 * 16;
        segs_room = std::max<uint64_t>(segs_room, ((uint64_t)g.n_segs + 1 + 63) / 64 * 64 + 64);  // (the count moves by a few from call to call)
        local_max = std::max<uint64_t>(local_max, n_local);
    }
    std::vector<uint8_t> blob(blob_bytes);
    plan_.reserve(blob_bytes + 65536);
    if (transition_rate_ != cfg_.sample_rate || transition_frames_ != C) {
        transition_.upload(wave_transitions(wa.lp_lo, wa.hp_lo, wa.lp_hi, wa.hp_hi, C), stream);
        transition_rate_ = cfg_.sample_rate;
        transition_frames_ = C;
    }
    const uint64_t per_cut = local_max * 24;
    chunk_state_.reserve((size_t)state_floats);
    seg_sum_.reserve((size_t)(segs_room * per_cut));
    seg_mm_.reserve((size_t)(segs_room * local_max * 12));
    prefix_.reserve((size_t)(segs_room * per_cut * 2));
    bad_.reserve(1);
    OMX_HIP(hipMemsetAsync(bad_.ptr, 0, sizeof(uint32_t), stream));

    std::vector<WaveChunkArgs> args(groups.size());
    std::vector<std::pair<uint64_t, uint32_t>> kept_new;
    size_t at = 0;
    uint64_t state_at = 0;
    for (size_t k = 0; k < groups.size(); ++k) {
        ChunkGroup& g = groups[k];
        const uint32_t n_local = g.streams.empty() ? n_streams_ : (uint32_t)g.streams.size();
        WaveChunkArgs& ca = args[k];
        ca = WaveChunkArgs{};
        auto put = [&](const void* src, size_t bytes) {
            const size_t here = at;
            if (bytes) std::memcpy(blob.data() + at, src, bytes);
            at += pad16(bytes);
            return plan_.ptr + here;
        };
        ca.cuts = reinterpret_cast<const int32_t*>(put(g.cuts.data(), g.cuts.size() * sizeof(int32_t)));
        ca.chunk_seg = reinterpret_cast<const uint32_t*>(put(g.chunk_seg.data(), g.chunk_seg.size() * sizeof(uint32_t)));
        ca.evals = reinterpret_cast<const WaveEval*>(put(g.evals.data(), g.evals.size() * sizeof(WaveEval)));
        const uint8_t* map = put(g.streams.data(), g.streams.size() * sizeof(uint32_t));
        ca.stream_map = g.streams.empty() ? nullptr : reinterpret_cast<const uint32_t*>(map);
        ca.n_local = n_local;
        ca.pcm = wa.pcm;
        ca.frames = g.frames;
        ca.pcm_stride = pcm_stride;
        ca.n_streams = n_streams_;
        ca.m00 = wa.fmt.m[0][0];
        ca.m10 = wa.fmt.m[1][0];
        ca.m01 = wa.fmt.m[0][1];
        ca.m11 = wa.fmt.m[1][1];
        ca.lp_lo = wa.lp_lo;
        ca.hp_lo = wa.hp_lo;
        ca.lp_hi = wa.lp_hi;
        ca.hp_hi = wa.hp_hi;
        auto widen = [](const BiquadCoef& c) {
            return WaveChunkArgs::Coef64{{(double)c.b[0], (double)c.b[1], (double)c.b[2]}, {(double)c.a[0], (double)c.a[1]}};
        };
        ca.lp_lo64 = widen(wa.lp_lo);
        ca.hp_lo64 = widen(wa.hp_lo);
        ca.lp_hi64 = widen(wa.lp_hi);
        ca.history = history ? 1u : 0u;
        ca.chunk_frames = g.C;
        ca.n_chunks = g.n_chunks;
        ca.pushes0 = g.pushes0;
        ca.color_len = color_len_;
        ca.slow_len = slow_len_;
        ca.color_ring = color_ring_.ptr;
        ca.hist_ring = hist_ring_.ptr;
        ca.state = state_.ptr;
        ca.n_segs = g.n_segs;
        ca.n_old_segs = g.n_old;
        ca.n_evals = (uint32_t)g.evals.size();
        ca.chunk_state = chunk_state_.ptr + state_at;
        state_at += (uint64_t)g.n_chunks * n_local * 3 * 16;
        ca.seg_sum = seg_sum_.ptr;
        ca.seg_mm = seg_mm_.ptr;
        ca.prefix_hi = prefix_.ptr;
        ca.prefix_lo = prefix_.ptr + (uint64_t)(g.n_segs + 1) * n_local * 24;
        ca.bad = bad_.ptr;
        ca.base_slot = kWaveNoSlot;
        if (g.kept) {
            // slots for the totals this call leaves (none when the table is full: later calls then read the rings)
            std::vector<uint32_t> keep;
            if (kept_reserve(kept_.size() + g.keep_cuts.size(), stream)) {
                for (int32_t c : g.keep_cuts) {
                    const uint64_t covered = g.pushes0 + (uint64_t)((int64_t)c + 1);
                    const uint32_t slot = kept_free_.back();
                    kept_free_.pop_back();
                    kept_new.emplace_back(covered, slot);
                    keep.push_back((uint32_t)(std::lower_bound(g.cuts.begin(), g.cuts.end(), c) - g.cuts.begin()));
                    keep.push_back(slot);
                }
            }
            ca.old_slot = reinterpret_cast<const uint32_t*>(put(g.old_slot.data(), g.old_slot.size() * sizeof(uint32_t)));
            ca.keep = reinterpret_cast<const uint32_t*>(put(keep.data(), keep.size() * sizeof(uint32_t)));
            ca.n_keep = (uint32_t)(keep.size() / 2);
            ca.totals = kept_totals_.ptr;
            ca.all_kept = std::find(g.old_slot.begin(), g.old_slot.end(), kWaveNoSlot) == g.old_slot.end() ? 1u : 0u;
            const auto base = kept_.find(g.pushes0);
            if (base != kept_.end()) ca.base_slot = base->second;
            ca.void_end = void_end_.ptr;
            ca.first_count = (uint64_t)std::max<int64_t>((int64_t)g.pushes0 + (int64_t)g.cuts[0] + 1, 0);
            ca.end_count = g.pushes0 + g.frames;
        }
        ca.columns = wa.columns;
        ca.preview = wa.preview;
        ca.col_stride = col_stride;
        ca.write_preview = g.write_preview;
    }
    plan_staging_.upload(blob.data(), blob.size(), plan_.ptr, stream);
    for (const WaveChunkArgs& ca : args) launch_waveform_chunked_phase1(ca, transition_.ptr, stream);
    for (const WaveChunkArgs& ca : args) launch_waveform_chunked_phase2(ca, stream);
    OMX_HIP(hipGetLastError());
    for (const auto& kv : kept_new) kept_[kv.first] = kv.second;
    return true;
}

// The chunk-parallel form of one lock-step call.  Returns false when the call's shape is not served (the caller then runs the
// sequential kernel alone); after a true return the caller still launches the sequential kernel, predicated on the `bad` flag.
bool WaveformBank::run_chunked(WaveformArgs& wa, const std::vector<uint32_t>& column_ends, double end_phase, hipStream_t stream) {
    const uint64_t frames = wa.frames;
    if (form_ == 1 || !analysis_ || !chunk_shape_ok(frames)) return false;
    if (color_len_ < 64 || slow_len_ < 64 || slow_len_ > 0x3FFFFFFFu) return false;
    // by shape: the sequential kernels take ~146 ns per frame whatever the bank size (up to 1024 streams), the chunk form ~0.2 ms of
    // launches plus its work — it wins from ~2000 frames per call on (tools/bench_wave_forms.py: 1 stream x 16 384 frames 2.33 -> 0.32 ms,
    // 64 x 65 536 9.3 -> 0.47 ms, 1024 x 1024 0.17 -> 0.22 ms)
    if (form_ != 2 && frames < 2048) return false;
    std::vector<ChunkGroup> groups(1);
    groups[0].frames = frames;
    groups[0].pushes0 = pushes_;
    groups[0].column_ends = column_ends;
    groups[0].first_kept = wa.first_kept;
    groups[0].write_preview = wa.write_preview;
    // ---- kept totals: what later calls will start a window at, inside this call
    {
        const bool history = wa.track_history != 0;
        const uint32_t nwin = history ? 3u : 1u;
        const int64_t caps[3] = {(int64_t)color_len_, (int64_t)color_len_, (int64_t)slow_len_};
        const int64_t maxcap = history ? (int64_t)slow_len_ : (int64_t)color_len_, F = (int64_t)frames;
        const uint64_t P0 = pushes_, Pend = pushes_ + frames;
        for (auto it = kept_.begin(); it != kept_.end() && it->first + (uint64_t)maxcap < P0;) {  // out of every window's reach
            kept_free_.push_back(it->second);
            it = kept_.erase(it);
        }
        if (!kept_.empty() && kept_.find(P0) == kept_.end()) kept_invalidate();  // (a call that left no totals: nothing to continue from)
        void_end_.reserve(1);
        if (kept_zero_void_) OMX_HIP(hipMemsetAsync(void_end_.ptr, 0, sizeof(uint64_t), stream));
        kept_zero_void_ = false;
        if (P0 == 0 && kept_.empty() && kept_reserve(1, stream)) {  // the total of nothing
            const uint32_t slot = kept_free_.back();
            kept_free_.pop_back();
            OMX_HIP(hipMemsetAsync(kept_totals_.ptr + (size_t)slot * n_streams_ * 48, 0, (size_t)n_streams_ * 48 * sizeof(double), stream));
            kept_[0] = slot;
        }
        std::vector<int32_t>& keep = groups[0].keep_cuts;
        auto want = [&](int64_t cut) {
            if (cut >= 0 && cut <= F - 1) keep.push_back((int32_t)cut);
        };
        double phase = end_phase;  // the columns of the frames to come (processor.rs:287-291), as far as a window reaches
        for (int64_t f = F; f < F + maxcap; ++f) {
            phase += wa.step;
            if (phase >= 1.0) {
                phase -= 1.0;
                for (uint32_t w = 0; w < nwin; ++w) want(f - caps[w]);
            }
        }
        for (int64_t e = 2 * F - 1; e - maxcap <= F - 1; e += F)  // the ends of later calls of this length (their pseudo-columns)
            for (uint32_t w = 0; w < nwin; ++w) want(e - caps[w]);
        for (uint32_t w = 0; w < nwin; ++w)  // refresh points (dsp.rs:346-352): the multiples of the window length
            for (uint64_t m = P0 / (uint64_t)caps[w] + 1; m * (uint64_t)caps[w] <= Pend; ++m) want((int64_t)(m * (uint64_t)caps[w] - P0) - 1);
        want(F - 1);
        std::sort(keep.begin(), keep.end());
        keep.erase(std::unique(keep.begin(), keep.end()), keep.end());
        groups[0].kept = &kept_;
    }
    if (!launch_chunk_groups(wa, groups, frames, wa.n_emit - wa.first_kept, stream)) return false;
    wa.run_if = bad_.ptr;
    return true;
}

// The chunk-parallel form of a ragged call: the host keeps a mirror of every stream's push count and column phase (both are pure
// functions of the frame counts and reset flags it has been handed), sorts the call's streams into groups that move in lock step,
// and runs one plan per group.  More than kMaxGroups distinct (frames, pushes, phase) among the streams, or a group whose shape the
// chunk form does not serve: the sequential kernel does the call (the mirror stays valid as long as its replay stays cheap).
// Streams flagged in reset_mask join the group of (frames, 0, 0) after their state has been cleared.
bool WaveformBank::run_chunked_ragged(WaveformArgs& wa, const uint32_t* frames, const uint8_t* reset_mask, uint64_t max_cols, double step,
                                      hipStream_t stream) {
    constexpr size_t kMaxGroups = 8;
    if (!mirror_valid_) return false;
    struct Key {
        uint64_t frames, pushes, phase_bits;
        bool operator<(const Key& o) const { return std::tie(frames, pushes, phase_bits) < std::tie(o.frames, o.pushes, o.phase_bits); }
    };
    std::map<Key, std::vector<uint32_t>> classes;
    for (uint32_t s = 0; s < n_streams_; ++s) {
        const bool reset = reset_mask && reset_mask[s];
        Key k{frames[s], reset ? 0 : h_pushes_[s], 0};
        const double ph = reset ? 0.0 : h_phase_[s];
        std::memcpy(&k.phase_bits, &ph, sizeof(double));
        classes[k].push_back(s);
        if (classes.size() > 4 * kMaxGroups) {  // replaying that many phases on the host costs more than the call: give the mirror up
            mirror_valid_ = false;
            return false;
        }
    }
    // replay every class's phase (the reference's f64 add / compare / subtract, :287-291) and advance the mirror — whichever kernels run
    std::vector<ChunkGroup> groups;
    bool servable = classes.size() <= kMaxGroups && form_ != 1 && analysis_ && color_len_ >= 64 && slow_len_ >= 64 && slow_len_ <= 0x3FFFFFFFu;
    uint64_t longest = 0;
    std::vector<uint32_t> h_cols(n_streams_, 0), resets;
    std::vector<float> h_progress(n_streams_, 0.0f);
    // the mirror advances in copies: committed once the launches below have gone through (or this function has declined the call with
    // the mirror still describing the device, which the sequential kernel then advances itself — see the commit points)
    std::vector<uint64_t> new_pushes = h_pushes_;
    std::vector<double> new_phase = h_phase_;
    for (auto& kv : classes) {
        const Key& k = kv.first;
        double phase;
        std::memcpy(&phase, &k.phase_bits, sizeof(double));
        ChunkGroup g;
        g.frames = k.frames;
        g.pushes0 = k.pushes;
        for (uint64_t f = 0; f < k.frames; ++f) {
            phase += step;
            if (phase >= 1.0) {
                g.column_ends.push_back((uint32_t)f);
                phase -= 1.0;
            }
        }
        const float progress = (float)std::min(std::max(phase, 0.0), 1.0);  // (a stream without frames reports its standing phase)
        g.write_preview = progress > 0.0f ? 1u : 0u;
        g.streams = kv.second;
        for (uint32_t s : kv.second) {
            if (reset_mask && reset_mask[s]) resets.push_back(s);
            new_pushes[s] = k.pushes + (analysis_ ? k.frames : 0);
            new_phase[s] = phase;
            h_cols[s] = (uint32_t)std::min<uint64_t>(g.column_ends.size(), max_cols);
            h_progress[s] = progress;
        }
        if (k.frames == 0) continue;  // nothing to compute for these streams (a reset among them is applied below)
        if (!chunk_shape_ok(k.frames) || g.column_ends.size() > max_cols) servable = false;
        longest = std::max(longest, k.frames);
        groups.push_back(std::move(g));
    }
    // (declining from here on: the sequential kernel runs the call and advances the device counters exactly as replayed above, so the
    // mirror takes the replayed values; an EXCEPTION below leaves the mirror untouched and invalid — the device did not move)
    auto commit = [&] {
        h_pushes_.swap(new_pushes);
        h_phase_.swap(new_phase);
    };
    if (!servable || groups.empty()) {
        commit();
        return false;
    }
    if (form_ != 2 && longest < 2048) {  // (the lock-step rule: short calls stay on the sequential kernels)
        commit();
        return false;
    }
    struct Invalidate {  // (a throw between here and the commit: the mirror no longer knows what the device holds)
        bool& valid;
        bool armed = true;
        ~Invalidate() {
            if (armed) valid = false;
        }
    } guard{mirror_valid_};
    // reset_audio of single streams (:153-155 -> rebuild): their state is cleared before anything reads it (the rings need no clearing:
    // nothing older than the push count — now 0 — is read).  Harmless should the call fall back: the sequential kernel clears them too.
    if (!resets.empty()) {
        reset_list_.reserve(resets.size());
        reset_staging_.upload(resets.data(), resets.size() * sizeof(uint32_t), reset_list_.ptr, stream);
        launch_waveform_reset_streams(state_.ptr, reset_list_.ptr, (uint32_t)resets.size(), stream);
    }
    if (!launch_chunk_groups(wa, groups, wa.frames, max_cols, stream)) {
        guard.armed = false;
        commit();
        return false;
    }
    commit();
    // the per-stream counters the sequential kernels and the caller read: push counts, phases, column counts, preview progress
    const size_t n = n_streams_;
    std::vector<uint8_t> blob(n * (sizeof(uint64_t) + sizeof(double) + sizeof(uint32_t) + sizeof(float)));
    uint8_t* p = blob.data();
    std::memcpy(p, h_pushes_.data(), n * sizeof(uint64_t));
    std::memcpy(p + n * 8, h_phase_.data(), n * sizeof(double));
    std::memcpy(p + n * 16, h_cols.data(), n * sizeof(uint32_t));
    std::memcpy(p + n * 20, h_progress.data(), n * sizeof(float));
    mirror_dev_.reserve(blob.size());
    mirror_staging_.upload(blob.data(), blob.size(), mirror_dev_.ptr, stream);
    // ... copied into pushes_v / phase_v / cols_v / progress_v behind the chunk kernels, unless `bad` (then the sequential kernel writes them)
    launch_waveform_mirror_copy(mirror_dev_.ptr, n_streams_, wa.pushes_v, wa.phase_v, wa.cols_v, wa.progress_v, bad_.ptr, stream);
    guard.armed = false;
    wa.run_if = bad_.ptr;
    return true;
}

int WaveformBank::fetch(uint64_t stream_index, omx_wave_column* columns, omx_wave_column* preview, hipStream_t stream) {
    if (stream_index >= n_streams_) return OMX_ERR_INVALID;
    if (columns && last_cols_)
        copy_out(columns, columns_.ptr + stream_index * last_cols_ * 4, last_cols_ * 4 * sizeof(omx_wave_column), columns_.pinned, stream);
    if (preview) copy_out(preview, preview_.ptr + stream_index * 4, 4 * sizeof(omx_wave_column), preview_.pinned, stream);
    return OMX_NONE;
}

}  // namespace omx
