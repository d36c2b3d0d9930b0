// Host side of the loudness path: reference src/visuals/loudness/processor.rs:22-55 (K-weighting
// design), :73-97 (true-peak FIR taps), :164-251 (channel state lifecycle) driving kernel K4.
#include "loudness.hpp"

namespace omx {

// By shape, a bank call of this many blocks takes the chunk-parallel form WHATEVER the bank size: the sequential kernels cost ~37 us per
// block however few streams there are (one wavefront per 64 slots walks the frames), the chunk form ~0.12 ms of launches plus its
// work (tools/bench_meter_forms.py: 1 stream x 64 blocks 2.37 -> 0.12 ms, 16 streams x 8 blocks 0.30 -> 0.12 ms; until round 4 the rule
// also asked for >= 4096 (slot, block) items, i.e. it left small banks on the slow side by 20x).
constexpr uint64_t kChunkedFromBlocks = 4;

constexpr float kLoudnessDefaultFloor = -99.9f;                           // :11
constexpr float kLoudnessWindowsSecs[4] = {3.0f, 0.4f, 0.3f, 1.0f};       // :13
constexpr size_t kTruePeakTaps = 48;                                      // :75

void loudness_config_default(omx_loudness_config* c) {
    c->sample_rate = kDefaultSampleRate;
    c->floor_db = kLoudnessDefaultFloor;
}

void k_weighting_coefficients(double fs, double b[5], double a[5]) {  // :22-55
    double f0 = 1681.974450955533, g = 3.999843853973347, q = 0.7071752369554196;
    double k = std::tan(M_PI * f0 / fs);
    const double vh = std::pow(10.0, g / 20.0);
    const double vb = std::pow(vh, 0.4996667741545416);
    double a0 = 1.0 + k / q + k * k;
    const double pb[3] = {(vh + vb * k / q + k * k) / a0, 2.0 * (k * k - vh) / a0, (vh - vb * k / q + k * k) / a0};
    const double pa[3] = {1.0, 2.0 * (k * k - 1.0) / a0, (1.0 - k / q + k * k) / a0};
    f0 = 38.13547087602444;
    q = 0.5003270373238773;
    k = std::tan(M_PI * f0 / fs);
    a0 = 1.0 + k / q + k * k;
    const double rb[3] = {1.0, -2.0, 1.0};
    const double ra[3] = {1.0, 2.0 * (k * k - 1.0) / a0, (1.0 - k / q + k * k) / a0};
    auto conv = [](const double p[3], const double r[3], double out[5]) {
        out[0] = p[0] * r[0];
        out[1] = p[0] * r[1] + p[1] * r[0];
        out[2] = p[0] * r[2] + p[1] * r[1] + p[2] * r[0];
        out[3] = p[1] * r[2] + p[2] * r[1];
        out[4] = p[2] * r[2];
    };
    conv(pb, rb, b);
    conv(pa, ra, a);
}

static float true_peak_coefficient(size_t j, size_t factor) {  // :79-84
    const double offset = (double)j - (double)kTruePeakTaps * 0.5;
    const double window = 0.5 * (1.0 - std::cos(2.0 * M_PI * (double)j / (double)kTruePeakTaps));
    const double x = offset * M_PI / (double)factor;
    return (float)(window * std::sin(x) / x);
}
static size_t window_length(float sample_rate, float secs) {  // :68-71
    const float len = sample_rate * secs;
    return len < 1.0f ? 1 : f2usize((double)len);
}
static double channel_weight(uint8_t position) {  // :174-183
    switch (position) {
        case OMX_POS_LOW_FREQUENCY: return 0.0;
        case OMX_POS_REAR_LEFT:
        case OMX_POS_REAR_RIGHT:
        case OMX_POS_SIDE_LEFT:
        case OMX_POS_SIDE_RIGHT: return 1.41;
        default: return 1.0;
    }
}

// Zero-input transition of the K-weighting TDF-II over `frames` samples and its powers 2, 4 ... 32 as double-double pairs:
// [6][4][4] high parts, then [6][4][4] low parts.  One step with x = 0 (loudness/processor.rs:153-162):
//   y = f0; f0' = f1 - a1 y; f1' = f2 - a2 y; f2' = f3 - a3 y; f3' = -a4 y.
// The filter's poles sit at 1 - O(f / fs) (38 Hz high-pass: 0.9988 double pole at 192 kHz), the companion form is far from normal,
// and the block transition inherits that: entries of 3e5 whose products with the state cancel to the state's own size.  Entries
// rounded to f64 put 4e-4 of the state into a 192 kHz scan (1e-3 dB on a 15 Hz channel), an 80-bit recurrence still 2e-13 of an
// entry; so every power comes from its own recurrence in double-double arithmetic (~1e-32) and the device scan keeps the pairs.
namespace {
struct DD {
    double h, l;
};
inline DD dd_two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
inline DD dd_add(DD a, DD b) {
    DD s = dd_two_sum(a.h, b.h);
    const DD t = dd_two_sum(a.l, b.l);
    s.l += t.h;
    s = {s.h + s.l, s.l - ((s.h + s.l) - s.h)};
    s.l += t.l;
    return {s.h + s.l, s.l - ((s.h + s.l) - s.h)};
}
inline DD dd_mul_d(DD a, double b) {  // a * b, b an f64
    const double p = a.h * b, e = std::fma(a.h, b, -p) + a.l * b;
    return {p + e, e - ((p + e) - p)};
}
}  // namespace
static std::vector<double> k_weighting_transitions(const double b[5], const double a[5], uint64_t frames) {
    (void)b;
    std::vector<double> T(2 * 6 * 16, 0.0);
    auto store = [&](int p, const DD (&P)[4][4]) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                T[(size_t)p * 16 + (size_t)i * 4 + j] = P[i][j].h;
                T[(size_t)(6 + p) * 16 + (size_t)i * 4 + j] = P[i][j].l;
            }
    };
    // Every power from its own recurrence (frames << p steps): squaring the pairs loses what the cancellation inside the product takes
    // (2^-54 of the largest entry at 192 kHz by the fourth power, tests/test_cpu_boundary.py) — the recurrence does not cancel.
    // Blocks beyond 128 k frames (4 M steps for the six powers) square the highest power they can afford instead.
    DD P[4][4];
    DD f[4][4];  // column m: the state that started as unit vector m
    for (int m = 0; m < 4; ++m)
        for (int k = 0; k < 4; ++k) f[m][k] = {k == m ? 1.0 : 0.0, 0.0};
    uint64_t n = 0;
    int p = 0;
    for (; p < 6 && ((frames << p) <= (1ull << 22) || p == 0); ++p) {
        for (; n < (frames << p); ++n)
            for (int m = 0; m < 4; ++m) {
                const DD y = f[m][0];
                f[m][0] = dd_add(f[m][1], dd_mul_d(y, -a[1]));
                f[m][1] = dd_add(f[m][2], dd_mul_d(y, -a[2]));
                f[m][2] = dd_add(f[m][3], dd_mul_d(y, -a[3]));
                f[m][3] = dd_mul_d(y, -a[4]);
            }
        for (int m = 0; m < 4; ++m)
            for (int k = 0; k < 4; ++k) P[k][m] = f[m][k];
        store(p, P);
    }
    for (; p < 6; ++p) {
        DD S[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                DD acc = {0, 0};
                for (int k = 0; k < 4; ++k) acc = dd_add(acc, dd_add(dd_mul_d(P[i][k], P[k][j].h), dd_mul_d(P[i][k], P[k][j].l)));
                S[i][j] = acc;
            }
        std::memcpy(P, S, sizeof(P));
        store(p, P);
    }
    return T;
}

// The zero-state end state of a block is linear in its samples: f_L = sum_k A^(L-1-k) B x_k, with one step of the TDF-II written as
// f' = A f + B x (B_i = b_i - a_i b_0: y = b_0 x + f_0 substituted).  W[k][i] = (A^(L-1-k) B)_i, from the zero-input recurrence in
// double-double arithmetic, rounded to f64.  The device evaluates the four dot products in the pass that computes the true peak
// (no recurrence, no second K-weighting pass over the PCM): loudness_chunked.hip.
static std::vector<double> k_weighting_zero_state_weights(const double b[5], const double a[5], uint64_t frames) {
    std::vector<double> W((size_t)frames * 4);
    DD g[4];
    for (int i = 0; i < 4; ++i) g[i] = dd_add(DD{b[i + 1], 0.0}, dd_mul_d(DD{b[0], 0.0}, -a[i + 1]));
    for (uint64_t n = 0; n < frames; ++n) {
        const uint64_t k = frames - 1 - n;
        for (int i = 0; i < 4; ++i) W[(size_t)k * 4 + i] = g[i].h;
        const DD y = g[0];
        g[0] = dd_add(g[1], dd_mul_d(y, -a[1]));
        g[1] = dd_add(g[2], dd_mul_d(y, -a[2]));
        g[2] = dd_add(g[3], dd_mul_d(y, -a[3]));
        g[3] = dd_mul_d(y, -a[4]);
    }
    return W;
}

// test hook (omx_debug_k_weighting_transition, no device needed): the double-double block transition and its powers for a rate and block length
void k_weighting_transition_debug(double sample_rate, uint64_t frames, double out[192]) {
    double b[5], a[5];
    k_weighting_coefficients(sample_rate, b, a);
    const std::vector<double> t = k_weighting_transitions(b, a, frames);
    std::memcpy(out, t.data(), 192 * sizeof(double));
}

LoudnessBank::LoudnessBank(const omx_loudness_config& cfg, uint32_t n_streams) : n_streams_(n_streams) {
    cfg_ = cfg;  // :225-232: the config is stored as given; the weighting uses the sanitised rate
    k_weighting_coefficients((double)sanitize_sample_rate(cfg.sample_rate), b_, a_);
}

void LoudnessBank::clear_state(hipStream_t stream) {
    frames_seen_ = 0;
    if (state_.ptr) OMX_HIP(hipMemsetAsync(state_.ptr, 0, state_.count * sizeof(LoudnessChannelState), stream));
    if (ring_.ptr) OMX_HIP(hipMemsetAsync(ring_.ptr, 0, ring_.count * sizeof(RingT), stream));
    state_clean_ = true;
    q_valid_ = true;  // no samples yet: the running totals of the chunk-parallel path start from nothing
}

void LoudnessBank::reset_audio() {  // :234-236 every ChannelState back to default
    clear_state(last_stream_);
    ragged_ = false;
}

int LoudnessBank::process_ragged(const float* d_pcm, uint64_t block_frames, uint64_t max_blocks, const uint32_t* n_blocks,
                                 const uint8_t* reset_mask, uint32_t channels_in, float sample_rate,
                                 const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_loudness_ragged_update* out) {
    if (block_frames == 0 || block_frames > 0xFFFFFFFFull || max_blocks > 0xFFFFFFFFull) {
        set_last_error("loudness process_ragged: block_frames must be in 1 ... 2^32 - 1");
        return OMX_ERR_INVALID;
    }
    for (uint32_t s = 0; s < n_streams_; ++s)
        if (n_blocks[s] > max_blocks) {
            set_last_error("loudness process_ragged: n_blocks[s] > max_blocks");
            return OMX_ERR_INVALID;
        }
    return ragged_impl(d_pcm, block_frames * std::max<uint64_t>(max_blocks, 1), block_frames, nullptr, max_blocks, n_blocks, reset_mask, channels_in,
                       sample_rate, positions, stream, out);
}

// One block per capture, each of its own length: what VisualManager::ingest_samples hands LoudnessProcessor::process_block
// (registry.rs:396-418) when every capture has its own batcher (meter.rs:40-69: 1 ... 4 quanta per chunk, ONE call per chunk).
int LoudnessBank::process_chunks(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask,
                                 uint32_t channels_in, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                 omx_loudness_ragged_update* out) {
    if (frames_capacity == 0 || frames_capacity > 0xFFFFFFFFull) {
        set_last_error("loudness process_chunks: frames_capacity must be in 1 ... 2^32 - 1");
        return OMX_ERR_INVALID;
    }
    h_blocks_.resize(n_streams_);
    uint64_t longest = 1;
    for (uint32_t s = 0; s < n_streams_; ++s) {
        if (frames[s] > frames_capacity) {
            set_last_error("loudness process_chunks: frames[s] > frames_capacity");
            return OMX_ERR_INVALID;
        }
        h_blocks_[s] = frames[s] != 0 ? 1u : 0u;  // block.is_empty(): nothing happens
        longest = std::max<uint64_t>(longest, frames[s]);
    }
    return ragged_impl(d_pcm, frames_capacity, longest, frames, 1, h_blocks_.data(), reset_mask, channels_in, sample_rate, positions, stream, out);
}

int LoudnessBank::ragged_impl(const float* d_pcm, uint64_t row_frames, uint64_t block_frames, const uint32_t* frames_v, uint64_t max_blocks,
                              const uint32_t* n_blocks, const uint8_t* reset_mask, uint32_t channels_in, float sample_rate,
                              const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_loudness_ragged_update* out) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    last_stream_ = stream;
    bool any = false, any_reset = false;
    for (uint32_t s = 0; s < n_streams_; ++s) {
        any = any || n_blocks[s] != 0;
        any_reset = any_reset || (reset_mask && reset_mask[s]);
    }
    const bool clean_before = state_clean_;
    ensure_state(channels, sample_rate, stream);  // (a rate / channel-count change clears every stream, as in the lock-step call)
    const bool cleared_now = state_clean_ && !clean_before;
    if (!ragged_) {  // every stream starts from the bank's common counter
        r_seen_.upload(std::vector<uint64_t>(n_streams_, frames_seen_), stream);
        h_seen_.assign(n_streams_, frames_seen_);
        ragged_ = true;
    } else if (cleared_now) {
        OMX_HIP(hipMemsetAsync(r_seen_.ptr, 0, n_streams_ * sizeof(uint64_t), stream));
        h_seen_.assign(n_streams_, 0);
    }
    if (!any && !any_reset) return OMX_NONE;
    // per-stream counts / flags: pinned staging -> device (the caller's arrays are borrowed for the call only)
    r_staging_.upload(n_blocks, reset_mask, n_streams_, r_blocks_, r_mask_, stream, frames_v, &r_frames_);
    const uint64_t slots = std::max<uint64_t>(max_blocks, 1);
    snapshots_.reserve((size_t)(n_streams_ * slots), false);
    LoudnessArgs la{};
    fill_args(la, d_pcm, block_frames, slots, channels, positions);
    la.frames_total = row_frames;
    la.seen_v = r_seen_.ptr;
    la.blocks_v = r_blocks_.ptr;
    la.reset_v = r_mask_.ptr;
    la.frames_v = frames_v ? r_frames_.ptr : nullptr;
    // the host's mirror of the per-stream counters decides the form: every stream on the 64-sample sub-block grid
    bool grid_ok = !frames_v && block_frames % 64 == 0 && max_blocks >= 2;
    for (uint32_t s = 0; s < n_streams_; ++s) {
        if (reset_mask && reset_mask[s]) h_seen_[s] = 0;
        grid_ok = grid_ok && h_seen_[s] % 64 == 0;
        h_seen_[s] += (uint64_t)n_blocks[s] * (frames_v ? frames_v[s] : block_frames);
    }
    const bool chunked = grid_ok && chunked_mode_ != 0 && (chunked_mode_ == 1 || max_blocks >= kChunkedFromBlocks);
    last_form_ = chunked ? 2 : 1;
    timer_.begin(stream);
    if (chunked) {
        run_chunked(la, stream);
    } else {
        launch_loudness(la, stream);
        q_valid_ = false;
    }
    timer_.end(stream);
    OMX_HIP(hipGetLastError());
    state_clean_ = false;  // (frames_seen_ is meaningless from here on: the counters are per stream)
    last_blocks_ = slots;
    if (out) {
        out->n_streams = n_streams_;
        out->max_blocks = slots;
        out->d_n_blocks = r_blocks_.ptr;
        out->d_snapshots = snapshots_.ptr;
        out->d_reset = r_mask_.ptr;
        out->d_block_frames = frames_v ? r_frames_.ptr : nullptr;
    }
    return any ? OMX_PRODUCED : OMX_NONE;
}

void LoudnessBank::ensure_state(uint32_t requested, float sample_rate_in, hipStream_t stream) {  // :238-251
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(requested, 1), OMX_MAX_CHANNELS);
    const float sample_rate = sanitize_sample_rate(sample_rate_in);
    const bool rate_changed = cfg_.sample_rate != sample_rate;
    if (rate_changed) {
        cfg_.sample_rate = sample_rate;
        k_weighting_coefficients((double)sample_rate, b_, a_);
    }
    uint64_t len = 1;
    for (int w = 0; w < 4; ++w) len = std::max<uint64_t>(len, window_length(cfg_.sample_rate, kLoudnessWindowsSecs[w]));
    const bool realloc = ring_len_ != len || !ring_.ptr;
    if (realloc) {
        ring_len_ = len;
        ring_.reserve((size_t)(len * (((uint64_t)n_streams_ * 8 + 63) / 64) * 64));  // [group of 64 slots][ring slot][64]
        state_.reserve((size_t)n_streams_ * 8);
    }
    if (rate_changed || channels_ != channels || realloc) {
        channels_ = channels;
        clear_state(stream);
    }
}

void LoudnessBank::fill_args(LoudnessArgs& la, const float* d_pcm, uint64_t block_frames, uint64_t n_blocks, uint32_t channels,
                             const uint8_t positions[OMX_MAX_CHANNELS]) {
    la.pcm = d_pcm;
    la.frames_total = block_frames * n_blocks;
    la.block_frames = (uint32_t)block_frames;
    la.n_blocks = (uint32_t)n_blocks;
    la.n_streams = n_streams_;
    la.channels = channels;
    la.slot_shift = channels == 1 ? 0u : (channels == 2 ? 1u : (channels == 4 ? 2u : 3u));  // state / ring are cleared on a channel change
    for (int i = 0; i < 5; ++i) {
        la.b[i] = b_[i];
        la.a[i] = a_[i];
    }
    for (int i = 0; i < OMX_MAX_CHANNELS; ++i) {
        la.weights[i] = channel_weight(positions[i]);
        la.positions[i] = positions[i];
    }
    for (size_t tap = 0; tap < 12; ++tap)
        for (size_t phase = 0; phase < 3; ++phase) la.fir4[tap][phase] = true_peak_coefficient(tap * 4 + phase + 1, 4);
    for (size_t tap = 0; tap < 24; ++tap) la.fir2[tap] = true_peak_coefficient(tap * 2 + 1, 2);
    const double sr = (double)cfg_.sample_rate;
    la.delay_len = sr < 96000.0 ? 12 : (sr < 192000.0 ? 24 : 0);  // :107-114
    for (int w = 0; w < 4; ++w) la.capacities[w] = std::max<uint64_t>(window_length(cfg_.sample_rate, kLoudnessWindowsSecs[w]), 1);
    la.ring_len = ring_len_;
    la.frames_seen = frames_seen_;
    la.ring = ring_.ptr;
    la.state = state_.ptr;
    la.floor_db = cfg_.floor_db;
    la.snapshots = snapshots_.ptr;
}

// Chunk-parallel evaluation of one call (lock-step, or ragged when la.blocks_v is set): loudness_chunked.hip.
void LoudnessBank::run_chunked(LoudnessArgs& la, hipStream_t stream) {
    const uint64_t block_frames = la.block_frames, n_blocks = la.n_blocks, frames = block_frames * n_blocks;
    const uint64_t slots = (uint64_t)n_streams_ << la.slot_shift;
    bool off_grid = false;  // 44.1 / 88.2 kHz: window lengths that are not multiples of 64 samples
    for (int w = 0; w < 4; ++w) off_grid = off_grid || la.capacities[w] % 64 != 0;
    if (transition_rate_ != cfg_.sample_rate || transition_frames_ != block_frames) {
        transition_.upload(k_weighting_transitions(b_, a_, block_frames), stream);
        zs_weights_.upload(k_weighting_zero_state_weights(b_, a_, block_frames), stream);
        transition_rate_ = cfg_.sample_rate;
        transition_frames_ = block_frames;
    }
    LoudChunkArgs ca{};
    ca.pcm = la.pcm;
    ca.frames_total = frames;
    ca.block_frames = (uint32_t)block_frames;
    ca.n_blocks = (uint32_t)n_blocks;
    ca.n_streams = n_streams_;
    ca.channels = la.channels;
    ca.slot_shift = la.slot_shift;
    for (int i = 0; i < 5; ++i) {
        ca.b[i] = la.b[i];
        ca.a[i] = la.a[i];
    }
    for (int i = 0; i < OMX_MAX_CHANNELS; ++i) {
        ca.weights[i] = la.weights[i];
        ca.positions[i] = la.positions[i];
    }
    std::memcpy(ca.fir4, la.fir4, sizeof(ca.fir4));
    std::memcpy(ca.fir2, la.fir2, sizeof(ca.fir2));
    ca.delay_len = la.delay_len;
    for (int w = 0; w < 4; ++w) ca.capacities[w] = la.capacities[w];
    ca.ring_len = ring_len_;
    ca.frames_seen = frames_seen_;
    ca.seen_v = la.seen_v;
    ca.blocks_v = la.blocks_v;
    ca.reset_v = la.reset_v;
    ca.ring = ring_.ptr;
    ca.state = state_.ptr;
    ca.floor_db = cfg_.floor_db;
    ca.snapshots = snapshots_.ptr;
    chunk_filter_.reserve((size_t)(slots * n_blocks * 4));
    sub_sums_.reserve((size_t)(slots * (frames / 64)));
    // the running totals of every sub-block a window of this call can start in: the ring's and the call's, as a power of two
    uint64_t q_need = 4096;
    while (q_need < ring_len_ / 64 + frames / 64 + 2) q_need *= 2;
    if (q_need > q_len_) {
        q_len_ = q_need;
        q_ring_.reserve((size_t)((uint64_t)n_streams_ * 8 * q_len_ * 2));  // high words, then low words
        OMX_HIP(hipMemsetAsync(q_ring_.ptr, 0, q_ring_.count * sizeof(double), stream));
        tails_.release();
        if (!state_clean_) q_valid_ = false;  // re-indexed: the totals come back from the sample ring
    }
    ca.q_ring = q_ring_.ptr;
    ca.q_lo = q_ring_.ptr + (uint64_t)n_streams_ * 8 * q_len_;
    ca.q_len = q_len_;
    if (off_grid && !tails_.ptr) {
        tails_.reserve((size_t)((uint64_t)n_streams_ * 8 * kLoudnessWindows * q_len_));
        OMX_HIP(hipMemsetAsync(tails_.ptr, 0, tails_.count * sizeof(double), stream));
        if (!state_clean_) q_valid_ = false;  // the tails of what is already in the ring
    }
    ca.tails = off_grid ? tails_.ptr : nullptr;
    for (int w = 0; w < 4; ++w) ca.tail_len[w] = (uint32_t)(la.capacities[w] % 64);
    bad_.reserve(1);
    OMX_HIP(hipMemsetAsync(bad_.ptr, 0, sizeof(uint32_t), stream));
    ca.chunk_filter = chunk_filter_.ptr;
    ca.zs_weights = zs_weights_.ptr;
    ca.sub_sums = sub_sums_.ptr;
    ca.bad = bad_.ptr;
    ca.scan_dd = cfg_.sample_rate > 100000.0f ? 1u : 0u;
    rebuild_scratch_.reserve((size_t)(slots * (ring_len_ / 64 + 1)));
    // The running totals grow with everything a stream has played since its last reset, and a window sum is the difference of two
    // of them.  They are kept as double-double pairs (q_ring / q_lo), so the difference is exact to ~1e-16 of the WINDOW sum, not of
    // the total (as plain f64 totals, 20 s at full scale put 5e-4 dB into a -100 dBFS passage that followed).  The periodic rebuild
    // from the sample ring (every `rebase_frames_` frames; it restarts the totals at the oldest sample the ring holds) is kept as a
    // second line of defence and as the test hook it always was.
    if (q_valid_ && q_age_ > rebase_frames_) q_valid_ = false;
    if (!q_valid_) {  // (also: earlier calls went through the sequential kernels)
        launch_loudness_rebuild_q(ca, rebuild_scratch_.ptr, nullptr, stream);
        q_valid_ = true;
        q_age_ = 0;
    }
    q_age_ += frames;
    launch_loudness_chunked(ca, transition_.ptr, stream);
    OMX_HIP(hipGetLastError());
    // non-finite PCM (pass A's flag): nothing above touched the state; the sequential kernel does the call instead, and
    // the running totals are rebuilt from the ring it leaves
    la.run_if = bad_.ptr;
    launch_loudness(la, stream);
    LoudChunkArgs after = ca;
    after.frames_seen = frames_seen_ + frames;  // (ragged: the counters the sequential kernel advanced)
    after.reset_v = nullptr;
    launch_loudness_rebuild_q(after, rebuild_scratch_.ptr, bad_.ptr, stream);
}

int LoudnessBank::process(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels_in,
                          float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                          const omx_loudness_snapshot** d_snapshots) {  // :253-311
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    last_stream_ = stream;
    if (block_frames == 0 || n_blocks == 0) return OMX_NONE;  // block.is_empty()
    if (block_frames > 0xFFFFFFFFull || n_blocks > 0xFFFFFFFFull) unsupported("loudness block shape beyond 2^32");
    if (ragged_) {
        set_last_error("loudness bank: per-stream positions are in use (process_ragged); reset_audio() returns the bank to lock-step calls");
        return OMX_ERR_INVALID;
    }
    ensure_state(channels, sample_rate, stream);
    const uint64_t frames = block_frames * n_blocks;
    const float* d_pcm = pcm;
    if (!pcm_on_device) {
        const size_t n = (size_t)n_streams_ * frames * channels;
        d_pcm = staging_.stage(pcm, n, stream);
    }
    snapshots_.reserve((size_t)(n_streams_ * n_blocks), host_outputs_ && n_streams_ * n_blocks <= 4096);
    LoudnessArgs la{};
    fill_args(la, d_pcm, block_frames, n_blocks, channels, positions);
    // chunk-parallel evaluation for bank-sized calls (loudness_chunked.hip): every block of the call in parallel
    const bool shape_ok = block_frames % 64 == 0 && n_blocks >= 2 && frames_seen_ % 64 == 0;
    const bool chunked = shape_ok && chunked_mode_ != 0 && (chunked_mode_ == 1 || n_blocks >= kChunkedFromBlocks);
    timer_.begin(stream);
    last_form_ = chunked ? 2 : 1;
    if (chunked) {
        run_chunked(la, stream);
    } else {
        launch_loudness(la, stream);
        q_valid_ = false;
    }
    timer_.end(stream);
    OMX_HIP(hipGetLastError());
    frames_seen_ += frames;
    state_clean_ = false;
    last_blocks_ = n_blocks;
    if (d_snapshots) *d_snapshots = snapshots_.ptr;
    return OMX_PRODUCED;
}

int LoudnessBank::fetch(uint64_t stream_index, uint64_t block, omx_loudness_snapshot* dst, hipStream_t stream) {
    if (stream_index >= n_streams_ || block >= last_blocks_) {
        set_last_error("loudness fetch: index out of range");
        return OMX_ERR_INVALID;
    }
    copy_out(dst, snapshots_.ptr + stream_index * last_blocks_ + block, sizeof(*dst), snapshots_.pinned, stream);
    return OMX_NONE;
}

}  // namespace omx
