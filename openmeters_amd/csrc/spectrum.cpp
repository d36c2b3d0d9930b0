// Host side of the spectrum path: the state machine of reference
// src/visuals/spectrum/processor.rs:72-323 (config normalisation, per-trace pending audio, hop
// draining with pending_skip, level-buffer resets) driving the K3 kernels.
#include "spectrum.hpp"

namespace omx {

constexpr float kDefaultSpectrumFloor = -100.0f;  // :22
constexpr size_t kDefaultHopDivisor = 16;         // :24
constexpr size_t kDefaultSpectrumFft = 16384;     // :25

void spectrum_config_default(omx_spectrum_config* c) {  // :39-51
    std::memset(c, 0, sizeof(*c));
    c->sample_rate = kDefaultSampleRate;
    c->window = OMX_WINDOW_HANN;
    c->fft_size = kDefaultSpectrumFft;
    c->hop_size = kDefaultSpectrumFft / kDefaultHopDivisor;
    c->averaging_mode = OMX_AVERAGING_NONE;
    c->averaging_param = 0.0f;
    c->source = OMX_CHANNEL_MID;
    c->secondary_source = OMX_CHANNEL_NONE;
    c->floor_db = kDefaultSpectrumFloor;
}

static void normalize(omx_spectrum_config& c) {  // :53-62
    c.sample_rate = sanitize_sample_rate(c.sample_rate);
    c.fft_size = std::max<uint64_t>(c.fft_size, 1);
    if (c.hop_size == 0) c.hop_size = std::max<uint64_t>(c.fft_size / kDefaultHopDivisor, 1);
    c.floor_db = (std::isfinite(c.floor_db) && c.floor_db < 0.0f) ? c.floor_db : kDefaultSpectrumFloor;  // level.rs:20-26
    c._pad = 0;
}

float a_weight_host(float freq_hz) {  // :410-425
    const double C1 = 20.598997 * 20.598997, C2 = 107.65265 * 107.65265, C3 = 737.86223 * 737.86223,
                 C4 = 12194.217 * 12194.217;
    if (freq_hz <= 0.0f) return -std::numeric_limits<float>::infinity();
    const double f = (double)freq_hz, f2 = f * f;
    const double numerator = C4 * f2 * f2;
    const double denom = (f2 + C1) * std::sqrt((f2 + C2) * (f2 + C3)) * (f2 + C4);
    return (float)(20.0 * std::log10(numerator / denom) + 2.0);
}

SpectrumBank::SpectrumBank(const omx_spectrum_config& cfg, uint32_t n_streams, bool emit_all_hops)
    : n_streams_(n_streams), emit_all_(emit_all_hops) {
    cfg_ = cfg;
    normalize(cfg_);
}

void SpectrumBank::active_traces(bool out[2]) const {  // :174-177
    out[0] = cfg_.source != OMX_CHANNEL_NONE;
    out[1] = cfg_.secondary_source != OMX_CHANNEL_NONE && cfg_.secondary_source != cfg_.source;
}

void SpectrumBank::reset_audio() {  // :112-118
    if (prepared_) reset_level_buffers(last_stream_);
    tail_ = head_;
    pending_skip_ = 0;
    carry_valid_ = false;
    ragged_ = false;  // every stream drops its pending audio: the common host-side positions describe the bank again
}

void SpectrumBank::prepare(hipStream_t stream) {
    if (!prepared_) rebuild_fft(stream);
}

// The reference plans any length (:71-82 only normalises); a shape the HIP path does not compute is a backend failure raised
// before any state changes.
static void require_supported(const omx_spectrum_config& c) {
    const size_t N = (size_t)c.fft_size;
    if (N > (size_t(1) << 24)) unsupported("spectrum FFT longer than 2^24");
    if (N < 1) unsupported("spectrum fft_size 0");
}

void SpectrumBank::rebuild_fft(hipStream_t stream) {  // :126-136
    require_supported(cfg_);
    prepared_ = false;
    const size_t N = (size_t)cfg_.fft_size;
    const std::vector<float> window = window_coefficients(cfg_.window, N);
    d_window_.upload(window, stream);
    d_bin_norm_.upload(fft_bin_normalization(window, N), stream);
    d_tw_fft_.upload(twiddle_table(N, std::max<size_t>(N / 2, 1)), stream);
    blu_m_ = 0;
    if (!is_pow2(N)) {  // any other length (the reference plans it with rustfft): Bluestein's chirp-z on the generic kernel
        const BluesteinHostTables t = bluestein_tables(N);
        blu_m_ = t.m;
        d_blu_chirp_.upload(t.chirp, stream);
        d_blu_bf_.upload(t.bf, stream);
        d_blu_tw_.upload(t.tw_m, stream);
        OMX_HIP(hipStreamSynchronize(stream));  // host vectors above go out of scope
    }
    fast4096_ = N == 16384 || N == 8192 || N == 4096 || N == 2048 || N == 1024;  // fused kernel sizes (every FFT size the GUI offers)
    if (fast4096_) {
        d_tw256_.upload(twiddle_table(256, 256), stream);
        d_tw4096_.upload(twiddle_table(N, N), stream);  // exp(-2 pi i k / N)
    }
    prepared_ = true;
    reset_buffers(stream);
}

void SpectrumBank::reset_buffers(hipStream_t stream) {  // :138-150
    const size_t bins = (size_t)cfg_.fft_size / 2 + 1;
    const float bin_hz = cfg_.sample_rate / (float)cfg_.fft_size;
    freq_bins_.resize(bins);
    a_weight_.resize(bins);
    for (size_t b = 0; b < bins; ++b) {
        const float f = (float)b * bin_hz;
        freq_bins_[b] = f;
        a_weight_[b] = a_weight_host(f);
    }
    d_freq_bins_.upload(freq_bins_, stream);
    d_a_weight_.upload(a_weight_, stream);
    reset_level_buffers(stream);
    tail_ = head_;
    pending_skip_ = 0;
    carry_valid_ = false;
}

void SpectrumBank::reset_level_buffers(hipStream_t stream) {  // :152-168
    const size_t bins = (size_t)cfg_.fft_size / 2 + 1;
    float headroom = 0.0f;  // :332-336 smoothing_state_floor
    for (float w : a_weight_) headroom = std::fmax(headroom, w);
    state_floor_ = std::fmax(db_to_power_host(cfg_.floor_db - headroom), std::numeric_limits<float>::min());
    if (cfg_.averaging_mode != OMX_AVERAGING_NONE) {
        d_smoothed_.reserve((size_t)n_streams_ * 2 * bins);
        OMX_HIP(hipMemsetAsync(d_smoothed_.ptr, 0, (size_t)n_streams_ * 2 * bins * sizeof(float), stream));
    }
    traces_dirty_ = true;  // snapshot traces go back to the floor
}

void SpectrumBank::update_config(const omx_spectrum_config& in, hipStream_t stream) {  // :300-322
    const omx_spectrum_config old = cfg_;
    omx_spectrum_config cfg = in;
    normalize(cfg);
    if (prepared_) require_supported(cfg);  // rejected: the handle keeps its old configuration and tables
    cfg_ = cfg;
    if (!prepared_) return;
    const bool mode_changed = old.averaging_mode != cfg.averaging_mode;
    if (old.fft_size != cfg.fft_size || old.window != cfg.window) {
        rebuild_fft(stream);
        ragged_ = false;  // every stream's pending audio is gone (:138-150): the common host positions describe the bank again
    } else if (old.sample_rate != cfg.sample_rate || old.hop_size != cfg.hop_size || old.source != cfg.source ||
               old.secondary_source != cfg.secondary_source) {
        reset_buffers(stream);
        ragged_ = false;
    } else if (mode_changed || std::fabs(old.floor_db - cfg.floor_db) > std::numeric_limits<float>::epsilon()) {
        reset_level_buffers(stream);
    }
}

void SpectrumBank::ensure_ring(uint64_t incoming, hipStream_t stream) {
    const uint64_t pending = head_ - tail_;
    const uint64_t need = pending + incoming;
    if (need <= ring_cap_ && ring_[0].ptr && ring_[1].ptr) return;
    const uint64_t cap = std::max<uint64_t>(next_pow2(std::max(need, ring_cap_)), 1024);
    for (int t = 0; t < 2; ++t) {
        DeviceBuffer<float> bigger;
        bigger.reserve((size_t)(cap * n_streams_));
        if (pending > 0 && ring_[t].ptr && ring_cap_) {
            for (uint32_t s = 0; s < n_streams_; ++s) {
                uint64_t pos = tail_;
                while (pos < head_) {
                    const uint64_t src_off = pos & (ring_cap_ - 1), dst_off = pos & (cap - 1);
                    const uint64_t run = std::min({head_ - pos, ring_cap_ - src_off, cap - dst_off});
                    OMX_HIP(hipMemcpyAsync(bigger.ptr + s * cap + dst_off, ring_[t].ptr + s * ring_cap_ + src_off,
                                           run * sizeof(float), hipMemcpyDeviceToDevice, stream));
                    pos += run;
                }
            }
            OMX_HIP(hipStreamSynchronize(stream));
        }
        std::swap(ring_[t].ptr, bigger.ptr);
        std::swap(ring_[t].count, bigger.count);
    }
    ring_cap_ = cap;
}

// process_block (:255-269) in three steps (see SpectrogramBank::push_begin): push_sources' bookkeeping, the ingest launch, the hops
int SpectrumBank::push_begin(uint64_t frames, uint32_t channels, float sample_rate_in, hipStream_t stream, IngestSlots& slots) {
    (void)channels;
    slots = IngestSlots{};
    last_stream_ = stream;
    if (ragged_) {
        set_last_error("spectrum bank is in ragged mode (per-stream positions): use process_ragged, or reset_audio() first");
        return OMX_ERR_INVALID;
    }
    if (frames == 0) return OMX_NONE;
    const float sample_rate = sanitize_sample_rate(sample_rate_in);
    if (sample_rate != cfg_.sample_rate) {  // :258-263
        cfg_.sample_rate = sample_rate;
        if (prepared_) reset_buffers(stream);
    }
    prepare(stream);
    bool active[2];
    active_traces(active);
    const uint32_t n_traces = (active[0] ? 1 : 0) + (active[1] ? 1 : 0);
    // ---- push_sources (:271-298)
    const uint64_t skip = std::min<uint64_t>(pending_skip_, frames);
    pending_skip_ -= skip;
    if (skip != frames && n_traces > 0) {
        const uint64_t count = frames - skip;
        ensure_ring(count, stream);
        if (active[0]) { slots.project[slots.n] = (int)cfg_.source; slots.ring[slots.n] = ring_[0].ptr; ++slots.n; }
        if (active[1]) { slots.project[slots.n] = (int)cfg_.secondary_source; slots.ring[slots.n] = ring_[1].ptr; ++slots.n; }
        for (int o = 0; o < slots.n; ++o) {
            slots.cap[o] = ring_cap_;
            slots.head[o] = head_;
        }
        slots.skip = skip;
        slots.count = count;
    }
    return OMX_PRODUCED;
}
void SpectrumBank::push_end(const IngestSlots& slots) { head_ += slots.count; }

int SpectrumBank::process(const float* pcm, bool pcm_on_device, uint64_t frames, uint32_t channels_in, float sample_rate_in,
                          const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_spectrum_bank_update* out) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    IngestSlots slots;
    const int rc = push_begin(frames, channels, sample_rate_in, stream, slots);
    if (rc != OMX_PRODUCED) return rc;
    if (slots.count) {
        const float* d_pcm = pcm;
        if (!pcm_on_device) {
            const size_t n = (size_t)n_streams_ * frames * channels;
            d_pcm = staging_.stage(pcm, n, stream);
        }
        const IngestSlots* one[1] = {&slots};
        launch_ingest_slots(d_pcm, frames, make_format(channels, positions), one, 1, n_streams_, stream);
        push_end(slots);
    }
    return process_pushed(stream, out);
}

int SpectrumBank::process_pushed(hipStream_t stream, omx_spectrum_bank_update* out) {
    bool active[2];
    active_traces(active);
    const uint32_t n_traces = (active[0] ? 1 : 0) + (active[1] ? 1 : 0);
    // ---- process_ready_windows (:179-213)
    if (n_traces == 0) return OMX_NONE;
    const uint64_t N = cfg_.fft_size, hop = cfg_.hop_size;
    const uint64_t tail0 = tail_;
    uint64_t n_hops = 0;
    while (head_ - tail_ >= N) {
        const uint64_t len = head_ - tail_;
        const uint64_t d = std::min<uint64_t>(hop, len);
        tail_ += d;
        pending_skip_ += hop - d;
        ++n_hops;
    }
    if (n_hops == 0) return OMX_NONE;
    if (hop > 0xFFFFFFFFull || n_hops > 0x7FFFFFFFull) unsupported("hop / hop count beyond 2^31");

    const uint64_t bins = N / 2 + 1;
    const bool averaging = cfg_.averaging_mode != OMX_AVERAGING_NONE;
    // AveragingMode::None with latest-hop output: earlier hops cannot influence the snapshot (:364-365)
    const uint64_t first_hop = (!averaging && !emit_all_) ? n_hops - 1 : 0;
    const int rc = launch_hops(tail0, nullptr, nullptr, n_hops, first_hop, n_traces, active, stream);
    if (out) {
        out->bins = bins;
        out->n_streams = n_streams_;
        out->n_hops = n_hops;
        out->n_hops_out = last_hops_out_;
        out->d_traces = d_traces_.ptr;
        out->d_frequency_bins = d_freq_bins_.ptr;
    }
    return rc;
}

// window.rs:76-79: every hop's mean is the reference's sequential f32 fold over the window / N.  The fold of the hops of this call, ahead
// of the transform kernel (window_sum_kernels.hip), one of two ways:
//   walk   four consecutive hops per lane quad walk the union of their windows: (W + 3 hop) dependent adds of latency whatever the
//          call's size — calls that complete many hops;
//   carry  lock-step calls that bring few samples (the reference's cadence, one batcher block per call): every window that has started
//          keeps its running fold in a slot between calls, and a call adds only the samples that arrived since (head - carry_pos adds).
// The carried folds are valid while the bank's positions move by lock-step pushes alone; reset_audio, any reconfiguration that drops
// the pending audio, a ragged call and a walked call invalidate them, and the next carried call folds its open windows from their
// first sample again (still in the ring: pending audio is exactly the open windows).
// (Tried and dropped, round 6: the walk on a side stream in the shadow of the capture group's spectrogram kernel — step 1.731 ... 1.741 ms
// against 1.712 with the walk in line: beside K2 it runs 120 us instead of 65, K2 stretches, and in line it leaves the ring in L2 for
// the transform kernel that follows, 344 -> 283 us.)
bool SpectrumBank::carry_applies(uint64_t tail0) const {
    const uint64_t N = cfg_.fft_size, hop = cfg_.hop_size;
    if (hop > N || (N + hop - 1) / hop > 64) return false;
    const uint64_t from = carry_valid_ ? carry_pos_ : tail0;
    return head_ - from <= N + 3 * hop;  // (the walk's latency)
}

void SpectrumBank::launch_window_sums_for(uint64_t tail0, const uint64_t* tails, const uint32_t* hops, uint64_t n_hops, uint64_t first_hop,
                                          uint32_t n_traces, const bool active[2], hipStream_t stream) {
    const uint64_t N = cfg_.fft_size, hop = cfg_.hop_size;
    const uint64_t slots = (N + hop - 1) / hop;
    const uint64_t hops_launch = n_hops - first_hop;
    d_hop_sums_.reserve((size_t)(n_streams_ * n_traces * hops_launch));
    const float* rings[2] = {nullptr, nullptr};
    uint32_t slot = 0;
    for (int t = 0; t < 2; ++t)
        if (active[t]) rings[slot++] = ring_[t].ptr;
    if (!tails && carry_applies(tail0)) {
        d_carry_.reserve((size_t)(n_streams_ * n_traces * slots));
        WindowCarryArgs c{};
        for (uint32_t t = 0; t < n_traces; ++t) c.ring[t] = rings[t];
        c.n_rings = n_traces;
        c.cap = ring_cap_;
        c.tail = tail0;
        c.carry_pos = carry_valid_ ? carry_pos_ : tail0;
        c.head = head_;
        c.n_windows = (head_ - tail0 + hop - 1) / hop;
        c.hop = (uint32_t)hop;
        c.window = (uint32_t)N;
        c.slots = (uint32_t)slots;
        c.slot0 = carry_valid_ ? carry_slot0_ : 0u;
        c.first_hop = (uint32_t)first_hop;
        c.n_hops = (uint32_t)hops_launch;
        c.n_streams = n_streams_;
        c.carry = d_carry_.ptr;
        c.sums = d_hop_sums_.ptr;
        launch_window_sums_carry(c, stream);
        carry_slot0_ = (uint32_t)((c.slot0 + n_hops) % slots);
        carry_pos_ = head_;
        carry_valid_ = true;
        return;
    }
    if (tails && hop <= N && slots <= 64) {  // ragged: the plan kernel has decided per stream (fold_mode); the carried streams first
        d_carry_.reserve((size_t)(n_streams_ * n_traces * slots));
        WindowCarryArgs c{};
        for (uint32_t t = 0; t < n_traces; ++t) c.ring[t] = rings[t];
        c.n_rings = n_traces;
        c.cap = ring_cap_;
        c.hop = (uint32_t)hop;
        c.window = (uint32_t)N;
        c.slots = (uint32_t)slots;
        c.first_hop = (uint32_t)first_hop;
        c.n_hops = (uint32_t)hops_launch;
        c.n_streams = n_streams_;
        c.carry = d_carry_.ptr;
        c.sums = d_hop_sums_.ptr;
        c.tails = tails;
        c.froms = r_fold_from_.ptr;
        c.heads = r_head_.ptr;
        c.slot0s = r_fold_slot0_.ptr;
        c.modes = r_fold_mode_.ptr;
        launch_window_sums_carry(c, stream);
    }
    carry_valid_ = false;
    WindowSumArgs w{};
    w.modes = tails ? r_fold_mode_.ptr : nullptr;
    for (uint32_t t = 0; t < n_traces; ++t) w.ring[t] = rings[t];
    w.n_rings = n_traces;
    w.cap = ring_cap_;
    w.tail = tail0;
    w.tails = tails;
    w.hops = hops;
    w.hop = (uint32_t)hop;
    w.window = (uint32_t)N;
    w.first_hop = (uint32_t)first_hop;
    w.n_hops = (uint32_t)hops_launch;
    w.n_streams = n_streams_;
    w.sums = d_hop_sums_.ptr;
    launch_window_sums(w, stream);
}

// The power (+ levels) launches of `n_hops` hops per stream slot, starting at hop `first_hop`.  Lock-step: every stream from
// `tail0`; ragged: stream s from tails[s], hops[s] of them (n_hops = the layout stride).
int SpectrumBank::launch_hops(uint64_t tail0, const uint64_t* tails, const uint32_t* hops, uint64_t n_hops, uint64_t first_hop, uint32_t n_traces,
                              const bool active[2], hipStream_t stream) {
    const uint64_t N = cfg_.fft_size, hop = cfg_.hop_size;
    const uint64_t bins = N / 2 + 1;
    const bool averaging = cfg_.averaging_mode != OMX_AVERAGING_NONE;
    const uint64_t hops_out = emit_all_ ? n_hops : 1;
    const uint64_t hops_launch = n_hops - first_hop;

    if (averaging) d_power_.reserve((size_t)(n_streams_ * n_traces * hops_launch * bins));
    const size_t traces_count = (size_t)(n_streams_ * hops_out * 4 * bins);
    if (d_traces_.count < traces_count) {
        d_traces_.reserve(traces_count, host_outputs_ && traces_count * sizeof(float) <= (size_t(1) << 20));
        traces_dirty_ = true;
    }
    if (traces_dirty_ || hops_out != last_hops_out_) {  // inactive traces stay at the floor (:155-157)
        launch_fill(d_traces_.ptr, d_traces_.count, cfg_.floor_db, stream);
        traces_dirty_ = false;
    }
    last_hops_out_ = hops_out;

    timer_.begin(stream);
    SpectrumPowerArgs pa{};
    uint32_t slot = 0, slots[2] = {0, 0};
    for (int t = 0; t < 2; ++t)
        if (active[t]) {
            pa.ring[slot] = ring_[t].ptr;
            slots[slot] = (uint32_t)t;
            ++slot;
        }
    pa.cap = ring_cap_;
    pa.tail = tail0;
    pa.tails = tails;
    pa.hops = hops;
    pa.hop = (uint32_t)hop;
    pa.first_hop = (uint32_t)first_hop;
    pa.n_hops = (uint32_t)hops_launch;
    pa.n_streams = n_streams_;
    pa.n_traces = n_traces;
    pa.fft_size = (uint32_t)N;
    pa.log_fft = log2_exact(N);
    pa.blu = BluesteinPlan{(uint32_t)blu_m_, blu_m_ ? log2_exact(blu_m_) : 0u, reinterpret_cast<const v2f*>(d_blu_chirp_.ptr),
                           reinterpret_cast<const v2f*>(d_blu_bf_.ptr), reinterpret_cast<const v2f*>(d_blu_tw_.ptr)};
    pa.bins = (uint32_t)bins;
    pa.window = d_window_.ptr;
    pa.bin_norm = d_bin_norm_.ptr;
    pa.tw_fft = reinterpret_cast<const v2f*>(d_tw_fft_.ptr);
    pa.tw256 = reinterpret_cast<const v2f*>(d_tw256_.ptr);
    pa.tw4096 = reinterpret_cast<const v2f*>(d_tw4096_.ptr);
    const bool fast = fast4096_ && !force_generic_ && ring_cap_ <= (uint64_t(1) << 30);  // the fused kernel uses 32-bit ring offsets
    uint64_t wgs = 0;
    if (!fast) {
        wgs = std::min<uint64_t>((uint64_t)n_streams_ * n_traces * hops_launch, 1024);
        while (wgs > 1 && wgs * (N + blu_m_) * sizeof(v2f) > (uint64_t(1) << 30)) wgs /= 2;
        d_workspace_.reserve((size_t)(wgs * (N + blu_m_) * 2));
        pa.workspace = reinterpret_cast<v2f*>(d_workspace_.ptr);
    }
    pa.power = d_power_.ptr;
    if (fast) {
        launch_window_sums_for(tail0, tails, hops, n_hops, first_hop, n_traces, active, stream);
        pa.hop_sums = d_hop_sums_.ptr;
    }
    pa.fused_db = averaging ? 0 : 1;
    pa.emit_all = emit_all_ ? 1 : 0;
    pa.n_hops_out = (uint32_t)hops_out;
    pa.trace_slot[0] = slots[0];
    pa.trace_slot[1] = slots[1];
    pa.state_floor = state_floor_;
    pa.floor_db = cfg_.floor_db;
    pa.a_weighting_db = d_a_weight_.ptr;
    pa.traces = d_traces_.ptr;
    launch_spectrum_power(pa, fast, (uint32_t)wgs, stream);
    if (!averaging) {  // the power kernel wrote the traces itself
        timer_.end(stream);
        OMX_HIP(hipGetLastError());
        return OMX_PRODUCED;
    }

    SpectrumLevelsArgs la{};
    la.power = d_power_.ptr;
    la.smoothed = averaging ? d_smoothed_.ptr : nullptr;
    la.traces = d_traces_.ptr;
    la.a_weighting_db = d_a_weight_.ptr;
    la.trace_slot[0] = slots[0];
    la.trace_slot[1] = slots[1];
    la.n_streams = n_streams_;
    la.n_traces = n_traces;
    la.n_hops = (uint32_t)hops_launch;
    la.n_hops_out = (uint32_t)hops_out;
    la.bins = (uint32_t)bins;
    la.mode = cfg_.averaging_mode;
    la.emit_all = emit_all_ ? 1 : 0;
    la.hops = hops;
    const float dt_seconds = (float)hop / cfg_.sample_rate;  // :184
    const float factor = cfg_.averaging_param;
    la.alpha = factor < 0.0f ? 0.0f : (factor > 0.9999f ? 0.9999f : factor);           // :367
    la.decay = db_to_power_host(-std::fmax(cfg_.averaging_param, 0.0f) * dt_seconds);  // :381
    la.state_floor = state_floor_;
    la.floor_db = cfg_.floor_db;
    launch_spectrum_levels(la, stream);
    timer_.end(stream);
    OMX_HIP(hipGetLastError());
    return OMX_PRODUCED;
}


// ------------------------------------------------------------------ ragged calls (per-stream frame counts and reset)
void SpectrumBank::enter_ragged(hipStream_t stream) {
    const size_t S = n_streams_;
    for (DeviceBuffer<uint64_t>* b : {&r_head_, &r_tail_, &r_skip_, &r_ing_head_, &r_hop_tail_}) b->reserve(S);
    for (DeviceBuffer<uint32_t>* b : {&r_ing_skip_, &r_ing_count_, &r_nhops_}) b->reserve(S);
    std::vector<uint64_t> h(S, head_), t(S, tail_), k(S, pending_skip_);  // the common lock-step state becomes every stream's state
    OMX_HIP(hipMemcpyAsync(r_head_.ptr, h.data(), S * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    OMX_HIP(hipMemcpyAsync(r_tail_.ptr, t.data(), S * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    OMX_HIP(hipMemcpyAsync(r_skip_.ptr, k.data(), S * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    // the carried window folds per stream (spectrum_plan_kernel keeps the books): nothing is carried across the switch
    for (DeviceBuffer<uint32_t>* b : {&r_carry_slot0_, &r_carry_valid_, &r_fold_mode_, &r_fold_slot0_}) b->reserve(S);
    for (DeviceBuffer<uint64_t>* b : {&r_carry_pos_, &r_fold_from_}) b->reserve(S);
    OMX_HIP(hipMemsetAsync(r_carry_valid_.ptr, 0, S * sizeof(uint32_t), stream));
    OMX_HIP(hipStreamSynchronize(stream));
    ragged_ = true;
    carry_valid_ = false;
}

int SpectrumBank::process_ragged(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask,
                                 uint32_t channels_in, float sample_rate_in, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                 omx_spectrum_ragged_update* out) {
    IngestArgs ia{};
    const int rc = ragged_plan(d_pcm, frames_capacity, frames, reset_mask, channels_in, sample_rate_in, positions, stream, ia);
    if (rc != OMX_PRODUCED) return rc;  // an error, or no active trace
    launch_ingest(ia, n_streams_, stream);
    OMX_HIP(hipGetLastError());
    return ragged_finish(stream, out);
}

// process_ragged in two halves (see SpectrogramBank::ragged_plan): up to the plan kernel, `ia_out` = what the projection launch needs ...
int SpectrumBank::ragged_plan(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask,
                              uint32_t channels_in, float sample_rate_in, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                              IngestArgs& ia_out) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    last_stream_ = stream;
    if (frames_capacity == 0 || frames_capacity > 0xFFFFFFFFull) {
        set_last_error("spectrum process_ragged: frames_capacity must be 1 .. 2^32 - 1");
        return OMX_ERR_INVALID;
    }
    for (uint32_t s = 0; s < n_streams_; ++s)
        if (frames[s] > frames_capacity) {
            set_last_error("spectrum process_ragged: frames[s] > frames_capacity");
            return OMX_ERR_INVALID;
        }
    const float sample_rate = sanitize_sample_rate(sample_rate_in);
    if (sample_rate != cfg_.sample_rate) {  // a format change concerns every stream of the bank (:258-263)
        cfg_.sample_rate = sample_rate;
        if (prepared_) reset_buffers(stream);
        ragged_ = false;  // (reset_buffers dropped every stream's pending audio: the common positions describe the bank again)
    }
    prepare(stream);
    bool active[2];
    active_traces(active);
    const uint32_t n_traces = (active[0] ? 1 : 0) + (active[1] ? 1 : 0);
    if (n_traces == 0) return OMX_NONE;
    if (!ragged_) enter_ragged(stream);
    const uint64_t N = cfg_.fft_size, hop = cfg_.hop_size, bins = N / 2 + 1;
    if (hop > 0xFFFFFFFFull) unsupported("hop beyond 2^32");
    // every stream enters a call with fewer than N pending samples (its ready windows were consumed), except right after the switch
    // from lock-step mode, where the common pending count is known
    const uint64_t pending_bound = std::max<uint64_t>(head_ - tail_, N ? N - 1 : 0);
    const uint64_t most = pending_bound + frames_capacity;
    const uint64_t max_hops = most >= N ? (most - N) / std::max<uint64_t>(std::min(hop, N), 1) + 1 : 0;
    if (max_hops > 0x7FFFFFFFull / std::max<uint64_t>(n_streams_, 1)) unsupported("too many hops in one call");

    // rings: room for the pending samples + this call's; growth re-homes every stream's pending samples on the device
    if (most > ring_cap_ || !ring_[0].ptr || !ring_[1].ptr) {
        const uint64_t cap = std::max<uint64_t>(next_pow2(std::max(most, ring_cap_)), 1024);
        for (int t = 0; t < 2; ++t) {
            DeviceBuffer<float> bigger;
            bigger.reserve((size_t)(cap * n_streams_));
            if (ring_[t].ptr && ring_cap_) launch_ring_rehome(ring_[t].ptr, ring_cap_, bigger.ptr, cap, r_head_.ptr, r_tail_.ptr, n_streams_, stream);
            OMX_HIP(hipStreamSynchronize(stream));  // the old buffer is freed below
            std::swap(ring_[t].ptr, bigger.ptr);
            std::swap(ring_[t].count, bigger.count);
        }
        ring_cap_ = cap;
    }
    // level state and trace rows exist before a reset can touch them
    const bool averaging = cfg_.averaging_mode != OMX_AVERAGING_NONE;
    const uint64_t hops_out = emit_all_ ? std::max<uint64_t>(max_hops, 1) : 1;
    {
        const size_t traces_count = (size_t)(n_streams_ * hops_out * 4 * bins);
        if (d_traces_.count < traces_count) {
            d_traces_.reserve(traces_count, false);
            traces_dirty_ = true;
        }
        if (traces_dirty_ || hops_out != last_hops_out_) {
            launch_fill(d_traces_.ptr, d_traces_.count, cfg_.floor_db, stream);
            traces_dirty_ = false;
        }
        last_hops_out_ = hops_out;
    }
    // the call's per-stream inputs (small: through double-buffered pinned memory, no stream synchronisation)
    r_staging_.upload(frames, reset_mask, n_streams_, r_frames_, r_mask_, stream);
    if (reset_mask) {
        launch_spectrum_reset_streams(r_mask_.ptr, n_streams_, averaging ? d_smoothed_.ptr : nullptr, 2 * bins, d_traces_.ptr, hops_out * 4 * bins,
                                      cfg_.floor_db, stream);
    }
    SpectrumPlanArgs pl{};
    pl.n_streams = n_streams_;
    pl.fft_size = N;
    pl.hop = hop;
    pl.max_hops = (uint32_t)max_hops;
    pl.frames = r_frames_.ptr;
    pl.reset_mask = reset_mask ? r_mask_.ptr : nullptr;
    pl.head = r_head_.ptr;
    pl.tail = r_tail_.ptr;
    pl.pending_skip = r_skip_.ptr;
    pl.ing_skip = r_ing_skip_.ptr;
    pl.ing_count = r_ing_count_.ptr;
    pl.ing_head = r_ing_head_.ptr;
    pl.hop_tail = r_hop_tail_.ptr;
    pl.n_hops = r_nhops_.ptr;
    {
        const uint64_t slots = (N + hop - 1) / hop;
        const bool fast = fast4096_ && !force_generic_ && ring_cap_ <= (uint64_t(1) << 30);
        pl.fold_slots = (fast && hop <= N && slots <= 64) ? (uint32_t)slots : 0u;
        pl.carry_pos = r_carry_pos_.ptr;
        pl.carry_slot0 = r_carry_slot0_.ptr;
        pl.carry_valid = r_carry_valid_.ptr;
        pl.fold_mode = r_fold_mode_.ptr;
        pl.fold_from = r_fold_from_.ptr;
        pl.fold_slot0 = r_fold_slot0_.ptr;
    }
    launch_spectrum_plan(pl, stream);

    IngestArgs ia{};
    ia.pcm = d_pcm;
    ia.frames_total = frames_capacity;
    ia.count = frames_capacity;  // grid bound; the per-stream values follow
    ia.skips = r_ing_skip_.ptr;
    ia.counts = r_ing_count_.ptr;
    ia.heads = r_ing_head_.ptr;
    ia.fmt = make_format(channels, positions);
    ia.n_out = 0;
    if (active[0]) { ia.project[ia.n_out] = (int)cfg_.source; ia.ring[ia.n_out] = ring_[0].ptr; ++ia.n_out; }
    if (active[1]) { ia.project[ia.n_out] = (int)cfg_.secondary_source; ia.ring[ia.n_out] = ring_[1].ptr; ++ia.n_out; }
    ia.cap = ring_cap_;
    ia.last_nonzero = nullptr;
    ia.partial_nonzero = nullptr;
    ia_out = ia;
    head_ = tail_ = 0;  // from here on only the bound above uses them (pending_bound = N - 1)
    pending_skip_ = 0;
    pend_max_hops_ = max_hops;
    pend_hops_out_ = hops_out;
    return OMX_PRODUCED;
}

// ... and the hop kernels + the update once the samples are in the rings
int SpectrumBank::ragged_finish(hipStream_t stream, omx_spectrum_ragged_update* out) {
    bool active[2];
    active_traces(active);
    const uint32_t n_traces = (active[0] ? 1 : 0) + (active[1] ? 1 : 0);
    const uint64_t max_hops = pend_max_hops_, hops_out = pend_hops_out_, bins = cfg_.fft_size / 2 + 1;
    int rc = OMX_NONE;
    if (max_hops > 0) rc = launch_hops(0, r_hop_tail_.ptr, r_nhops_.ptr, max_hops, 0, n_traces, active, stream);
    if (out) {
        std::memset(out, 0, sizeof(*out));
        out->bins = bins;
        out->n_streams = n_streams_;
        out->max_hops = max_hops;
        out->n_hops_out = hops_out;
        out->d_n_hops = r_nhops_.ptr;
        out->d_traces = d_traces_.ptr;
        out->d_frequency_bins = d_freq_bins_.ptr;
    }
    return rc;
}

int SpectrumBank::fetch(uint64_t stream_index, uint64_t hop, float* dst, hipStream_t stream) {
    if (stream_index >= n_streams_ || hop >= last_hops_out_ || !d_traces_.ptr) {
        set_last_error("spectrum fetch: index out of range");
        return OMX_ERR_INVALID;
    }
    const uint64_t bins = cfg_.fft_size / 2 + 1;
    copy_out(dst, d_traces_.ptr + (stream_index * last_hops_out_ + hop) * 4 * bins, 4 * bins * sizeof(float), d_traces_.pinned, stream);
    return OMX_NONE;
}

int SpectrumSingle::process_block(const omx_block* block, omx_spectrum_snapshot* out) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(block->channels, 1), OMX_MAX_CHANNELS);
    if (block->n_samples < channels) return OMX_NONE;
    omx_spectrum_bank_update bu;
    const int rc = bank.process(block->samples, false, block->n_samples / channels, channels, block->sample_rate,
                                block->positions, nullptr, &bu);
    if (rc != OMX_PRODUCED) return rc;
    traces.resize(4 * bu.bins);
    const int frc = bank.fetch(0, 0, traces.data(), nullptr);
    if (frc < 0) return frc;
    out->bins = bu.bins;
    out->frequency_bins = bank.frequency_bins().data();
    for (int t = 0; t < 2; ++t)
        for (int w = 0; w < 2; ++w) out->traces[t][w] = traces.data() + (size_t)(t * 2 + w) * bu.bins;
    return OMX_PRODUCED;
}

}  // namespace omx
