// CaptureGroup — see capture_group.hpp / include/omx.h.  Reference: VisualManager::{ingest_samples, reset_audio}
// (src/visuals/registry.rs:360-365, :396-418); the block partition of the block-based visuals is the batcher's (src/meter.rs:16-25).
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "capture_group.hpp"

namespace omx {

#ifdef OMX_TUNING
// host time per section of ingest_ragged (tuning build, OMX_GROUP_HOST_TIMES=1): printed when the process ends
struct HostSections {
    static constexpr int kN = 12;
    double us[kN] = {};
    uint64_t calls = 0;
    bool on = std::getenv("OMX_GROUP_HOST_TIMES") != nullptr;
    ~HostSections() {
        if (!on || !calls) return;
        const char* names[kN] = {"prologue", "spectrum finish", "loudness", "waveform", "stereometer", "oscilloscope", "join+stats",
                                 "spectrogram plan", "spectrum plan", "shared ingest", "spectrogram finish", ""};
        for (int k = 0; k < 11; ++k) std::fprintf(stderr, "ingest_ragged host: %-22s %7.1f us per call\n", names[k], us[k] / (double)calls);
    }
};
static HostSections g_host_sections;
struct HostLap {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(int k) {
        const auto n = std::chrono::steady_clock::now();
        g_host_sections.us[k] += std::chrono::duration<double, std::micro>(n - t).count();
        t = n;
    }
};
#define OMX_LAP(k) host_lap.lap(k)
#else
struct HostLap {};
#define OMX_LAP(k) (void)0
#endif

void capture_group_config_default(omx_capture_group_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->n_streams = 1;
    spectrogram_config_default(&c->spectrogram);
    spectrum_config_default(&c->spectrum);
    loudness_config_default(&c->loudness);
    stereometer_config_default(&c->stereometer);
    oscilloscope_config_default(&c->oscilloscope);
    waveform_config_default(&c->waveform);
}

constexpr uint32_t kAllVisuals = OMX_VISUAL_SPECTROGRAM | OMX_VISUAL_SPECTRUM | OMX_VISUAL_LOUDNESS | OMX_VISUAL_STEREOMETER |
                                 OMX_VISUAL_OSCILLOSCOPE | OMX_VISUAL_WAVEFORM;

void CaptureGroup::ensure_bank(uint32_t visual) {
    const uint32_t S = cfg_.n_streams;
    if (visual == OMX_VISUAL_SPECTROGRAM && !spectrogram_) spectrogram_.reset(new SpectrogramBank(cfg_.spectrogram, S));
    if (visual == OMX_VISUAL_SPECTRUM && !spectrum_) spectrum_.reset(new SpectrumBank(cfg_.spectrum, S, cfg_.spectrum_emit_all_hops != 0));
    if (visual == OMX_VISUAL_LOUDNESS && !loudness_) loudness_.reset(new LoudnessBank(cfg_.loudness, S));
    if (visual == OMX_VISUAL_STEREOMETER && !stereometer_) stereometer_.reset(new StereometerBank(cfg_.stereometer, S));
    if (visual == OMX_VISUAL_OSCILLOSCOPE && !oscilloscope_) oscilloscope_.reset(new OscilloscopeBank(cfg_.oscilloscope, S));
    if (visual == OMX_VISUAL_WAVEFORM && !waveform_) waveform_.reset(new WaveformBank(cfg_.waveform, S));
}

CaptureGroup::CaptureGroup(const omx_capture_group_config& cfg) : cfg_(cfg) {
    enabled_ = cfg.visuals & kAllVisuals;
    for (uint32_t bit = 1; bit <= OMX_VISUAL_WAVEFORM; bit <<= 1)
        if (enabled_ & bit) ensure_bank(bit);
    int prio_low = 0, prio_high = 0;
    OMX_HIP(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    const char* prio_env = tuning_env("OMX_GROUP_PRIO");  // tuning hook: "p0,p1,p2,p3" with 1 = the highest priority, 0 = default, -1 = lowest
    int want[kSideStreams] = {0, 0, 0, 0};
    if (prio_env) (void)std::sscanf(prio_env, "%d,%d,%d,%d", &want[0], &want[1], &want[2], &want[3]);
    for (int i = 0; i < kSideStreams; ++i) {
        if (want[i] != 0) OMX_HIP(hipStreamCreateWithPriority(&side_[i], hipStreamNonBlocking, want[i] > 0 ? prio_high : prio_low));
        else OMX_HIP(hipStreamCreateWithFlags(&side_[i], hipStreamNonBlocking));
        OMX_HIP(hipEventCreateWithFlags(&join_[i], hipEventDisableTiming));
    }
    OMX_HIP(hipEventCreateWithFlags(&fork_, hipEventDisableTiming));
}

CaptureGroup::~CaptureGroup() {
    for (int i = 0; i < kSideStreams; ++i) {
        if (side_[i]) {
            (void)hipStreamSynchronize(side_[i]);
            (void)hipStreamDestroy(side_[i]);
        }
        if (join_[i]) (void)hipEventDestroy(join_[i]);
    }
    if (fork_) (void)hipEventDestroy(fork_);
}

void CaptureGroup::reset_audio() {  // registry.rs:360-365: every module's reset_audio
    if (spectrogram_) spectrogram_->reset_audio();
    if (spectrum_) spectrum_->reset_audio();
    if (loudness_) loudness_->reset_audio();
    if (stereometer_) stereometer_->reset_audio();
    if (oscilloscope_) oscilloscope_->reset_audio();
    if (waveform_) waveform_->reset_audio();
    holds_valid_ = false;  // LoudnessState::reset_audio: fresh PeakHolds (loudness/state.rs:153-160)
    clock_ = 0.0;          // ... on a fresh sample clock
    ragged_stats_live_ = false;
    ragged_ = false;       // a bank-wide reset returns every bank to lock-step positions
    for (auto& p : pending_reset_) p.clear();  // every bank, enabled or not, has just been reset
    have_generation_ = false;  // registry.rs:361 format_generation = None
}

int CaptureGroup::set_enabled(uint32_t visual, bool on) {
    if (visual == 0 || (visual & (visual - 1)) != 0 || !(visual & kAllVisuals)) {
        set_last_error("capture group: `visual` must be exactly one OMX_VISUAL_* bit");
        return OMX_ERR_INVALID;
    }
    if (on) {
        if (ragged_ && !(enabled_ & visual)) {
            // a bank that has not seen the group's ragged calls holds lock-step positions: its first ragged call switches it over on
            // its own (every bank's process_ragged does), so nothing to do here beyond creating it
        }
        ensure_bank(visual);  // Entry::set_enabled: module.prepare()
        enabled_ |= visual;
    } else {
        enabled_ &= ~visual;
    }
    return OMX_NONE;
}

int CaptureGroup::update_config(uint32_t visual, const void* config, hipStream_t stream) {
    if (!config) return OMX_ERR_INVALID;
    switch (visual) {
        case OMX_VISUAL_SPECTROGRAM:
            cfg_.spectrogram = *static_cast<const omx_spectrogram_config*>(config);
            if (spectrogram_) spectrogram_->update_config(cfg_.spectrogram, stream);
            return OMX_NONE;
        case OMX_VISUAL_SPECTRUM:
            cfg_.spectrum = *static_cast<const omx_spectrum_config*>(config);
            if (spectrum_) spectrum_->update_config(cfg_.spectrum, stream);
            return OMX_NONE;
        case OMX_VISUAL_STEREOMETER:
            cfg_.stereometer = *static_cast<const omx_stereometer_config*>(config);
            if (stereometer_) stereometer_->update_config(cfg_.stereometer);
            return OMX_NONE;
        case OMX_VISUAL_OSCILLOSCOPE:
            cfg_.oscilloscope = *static_cast<const omx_oscilloscope_config*>(config);
            if (oscilloscope_) oscilloscope_->update_config(cfg_.oscilloscope);
            return OMX_NONE;
        case OMX_VISUAL_WAVEFORM:
            cfg_.waveform = *static_cast<const omx_waveform_config*>(config);
            if (waveform_) waveform_->update_config(cfg_.waveform);
            return OMX_NONE;
        case OMX_VISUAL_LOUDNESS:
            set_last_error("capture group: LoudnessProcessor has no update_config (loudness/processor.rs:225-253)");
            return OMX_ERR_INVALID;
        default:
            set_last_error("capture group: `visual` must be exactly one OMX_VISUAL_* bit");
            return OMX_ERR_INVALID;
    }
}

bool CaptureGroup::note_format_generation(uint64_t generation) {
    const bool changed = have_generation_ && generation_ != generation;
    if (changed) reset_audio();
    have_generation_ = true;
    generation_ = generation;
    return changed;
}

// The reset mask a bank's ragged call gets: the caller's mask, OR-ed with the per-capture resets that passed while the visual was
// disabled.  A disabled bank that exists only records the mask (VisualManager::reset_audio reaches every entry, registry.rs:360-365;
// Entry.enabled gates ingest alone, :413-417).  Returns nullptr when there is nothing to reset.
const uint8_t* CaptureGroup::mask_for(int vi, bool bank_enabled, bool bank_exists, const uint8_t* reset_mask) {
    const uint32_t S = cfg_.n_streams;
    std::vector<uint8_t>& pend = pending_reset_[vi];
    if (!bank_exists) return nullptr;  // created on enable, from nothing: no state a reset could clear
    if (!bank_enabled) {
        if (reset_mask) {
            bool any = false;
            for (uint32_t s = 0; s < S; ++s) any = any || reset_mask[s] != 0;
            if (any) {
                if (pend.empty()) pend.assign(S, 0);
                for (uint32_t s = 0; s < S; ++s) pend[s] |= reset_mask[s] ? 1 : 0;
            }
        }
        return nullptr;
    }
    if (pend.empty()) return reset_mask;
    std::vector<uint8_t>& m = mask_scratch_[vi];
    m.assign(S, 0);
    for (uint32_t s = 0; s < S; ++s) m[s] = (pend[s] || (reset_mask && reset_mask[s])) ? 1 : 0;
    // the merged mask stays pending until the bank's call has succeeded (ingest_ragged's `note`): a call that fails before the bank has
    // applied it — a plan error, an exception — must not lose the resets that arrived while the visual was disabled
    pend = m;
    mask_merged_[vi] = true;
    return m.data();
}

// Runs `body` between the fork of the side streams and their join onto the caller's stream; the join is enqueued on every path out
// of `body`, so an error inside it (a failed launch, a refused shape) never leaves the caller's stream unordered against side
// streams that may still be reading the caller's PCM.
template <class Body>
static int forked(hipStream_t stream, hipStream_t (&side)[kSideStreams], hipEvent_t fork, hipEvent_t (&join)[kSideStreams], bool (&used)[kSideStreams],
                  Body&& body) {
    OMX_HIP(hipEventRecord(fork, stream));
    auto rejoin = [&] {
        for (int i = 0; i < kSideStreams; ++i)
            if (used[i]) {
                (void)hipEventRecord(join[i], side[i]);
                (void)hipStreamWaitEvent(stream, join[i], 0);
            }
    };
    int rc;
    try {
        rc = body();
    } catch (...) {
        rejoin();
        throw;
    }
    rejoin();
    return rc;
}

void CaptureGroup::set_timing(bool on) {
    if (spectrogram_) spectrogram_->timer().enabled = on;
}

int CaptureGroup::ingest(const float* d_pcm, uint64_t frames, uint32_t channels_in, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS],
                         hipStream_t stream, omx_capture_group_update* out) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    const uint32_t S = cfg_.n_streams;
    omx_capture_group_update up;
    std::memset(&up, 0, sizeof(up));
    if (frames == 0) {  // block.is_empty(): nothing happens in any visual
        if (out) *out = up;
        return OMX_NONE;
    }
    if (ragged_) {
        set_last_error("capture group: per-capture positions are in use (omx_capture_group_ingest_ragged); lock-step calls resume after reset_audio");
        return OMX_ERR_INVALID;
    }
    // entry.enabled (registry.rs:413-417): a disabled visual's bank is skipped and keeps its state
    SpectrogramBank* spectrogram = (enabled_ & OMX_VISUAL_SPECTROGRAM) ? spectrogram_.get() : nullptr;
    SpectrumBank* spectrum = (enabled_ & OMX_VISUAL_SPECTRUM) ? spectrum_.get() : nullptr;
    LoudnessBank* loudness = (enabled_ & OMX_VISUAL_LOUDNESS) ? loudness_.get() : nullptr;
    StereometerBank* stereometer = (enabled_ & OMX_VISUAL_STEREOMETER) ? stereometer_.get() : nullptr;
    OscilloscopeBank* oscilloscope = (enabled_ & OMX_VISUAL_OSCILLOSCOPE) ? oscilloscope_.get() : nullptr;
    WaveformBank* waveform = (enabled_ & OMX_VISUAL_WAVEFORM) ? waveform_.get() : nullptr;
    const float sr = sanitize_sample_rate(sample_rate);
    // registry.rs:407-417: ONE AudioBlock of `frames` frames to every visual, whatever the chunk the batcher handed over (256 ... 1024
    // frames at 48 kHz, meter.rs:61-64) — one trigger evaluation, one true-peak take, one stereo_channels scan per call.  A host that
    // queues several chunks and replays them in one call says so with cfg.block_frames (then: frames / block_frames blocks).
    uint64_t block = frames, n_blocks = 1;
    if (cfg_.block_frames != 0 && frames % cfg_.block_frames == 0) {
        block = cfg_.block_frames;
        n_blocks = frames / block;
    }
    up.n_blocks = n_blocks;
    up.block_frames = block;
    const bool stats = stats_ && spectrogram && loudness && stereometer;
    // (no clear of the table ahead of the fork: that was a fill kernel and its launch gap in the one stretch of a call nothing overlaps.
    // Every visual writes its own columns on its own stream, or zeroes them when it produced nothing)
    if (stats) rows_.reserve((size_t)S * OMX_STATS_COLUMNS);
    int worst = OMX_NONE;
    auto note = [&](int rc, uint32_t bit) {
        if (rc < 0) worst = worst < 0 ? worst : rc;
        else if (rc == OMX_PRODUCED) up.produced |= bit;
    };
    bool used[kSideStreams] = {false, false, false, false};
    uint32_t stats_written = 0;
    forked(stream, side_, fork_, join_, used, [&] {
        // ---- the caller's stream: the banks that keep pending audio, fed by one projection of the block
        {
            IngestSlots sg, sp;
            int rc_sg = OMX_NONE, rc_sp = OMX_NONE;
            if (spectrogram) rc_sg = spectrogram->push_begin(frames, channels, sample_rate, stream, sg);
            if (spectrum) rc_sp = spectrum->push_begin(frames, channels, sample_rate, stream, sp);
            if (rc_sg < 0) note(rc_sg, 0);
            if (rc_sp < 0) note(rc_sp, 0);
            const AudioFormatArgs fmt = make_format(channels, positions);
            const bool both = rc_sg == OMX_PRODUCED && rc_sp == OMX_PRODUCED && sg.count && sp.count;
            if (both && shared_ingest_ && sg.skip == sp.skip && sg.count == sp.count && sg.n + sp.n <= OMX_INGEST_MAX_OUT) {
                const IngestSlots* two[2] = {&sg, &sp};
                launch_ingest_slots(d_pcm, frames, fmt, two, 2, S, stream);
                up.ingest_launches += 1;
            } else {
                if (rc_sg == OMX_PRODUCED && sg.count) {
                    const IngestSlots* one[1] = {&sg};
                    launch_ingest_slots(d_pcm, frames, fmt, one, 1, S, stream);
                    up.ingest_launches += 1;
                }
                if (rc_sp == OMX_PRODUCED && sp.count) {
                    const IngestSlots* one[1] = {&sp};
                    launch_ingest_slots(d_pcm, frames, fmt, one, 1, S, stream);
                    up.ingest_launches += 1;
                }
            }
            if (rc_sg == OMX_PRODUCED) {
                spectrogram->push_end(sg);
                note(spectrogram->process_pushed(stream, &up.spectrogram), OMX_VISUAL_SPECTROGRAM);
                // its summary columns right behind its kernel, INSIDE the fork: behind the join they were 12 us of kernel and a launch gap
                // that every call waited for before the next one's side streams could start (the rows' columns are disjoint per visual)
                if (stats && (up.produced & OMX_VISUAL_SPECTROGRAM) && up.spectrogram.d_counts && up.spectrogram.n_columns) {
                    launch_stats_spectrogram(up.spectrogram.d_counts, S, up.spectrogram.n_columns, rows_.ptr, stream);
                    stats_written |= kStatsSpectrogramColumns;
                }
            }
            if (stats && !(stats_written & kStatsSpectrogramColumns)) launch_stats_clear_columns(rows_.ptr, S, kStatsSpectrogramColumns, stream);
            if (rc_sp == OMX_PRODUCED) {
                spectrum->push_end(sp);
                note(spectrum->process_pushed(stream, &up.spectrum), OMX_VISUAL_SPECTRUM);
            }
        }
        // Which stream carries which meter bank: -1 = the caller's stream (behind the spectrogram / spectrum kernels), 0 ... 3 = a side stream.
        // HIP maps streams onto four hardware queues, and the chains are latency-bound kernels on a few dozen workgroups each (alone, per
        // 1024-capture 256-frame call: spectrogram 64 us, spectrum 27, loudness 70, stereometer 75, oscilloscope 115, waveform 69).
        //   regular cadence (one batcher block per call, meter.rs:61-64): caller {spectrogram, spectrum}, side 0 {loudness},
        //     side 1 {stereometer, waveform}, side 2 {oscilloscope} — four chains of 91 / 70 / 144 / 115 us: 260 ... 264 us per call against
        //     280 ... 295 with the oscilloscope behind the spectrum kernel on the caller's stream (round 5's layout; same box, three passes);
        //   catch-up chunks and replayed calls (>= 512 frames at 48 kHz): round 5's layout — caller {spectrogram, spectrum, oscilloscope},
        //     one side stream per meter bank — 462 us per 1024-frame chunk against 512 with the layout above (the trigger pass's > 100 KiB of
        //     LDS per workgroup wait for the waveform kernel's 66 KiB on every CU whichever queue they sit on, ledger Q-anyorder).
        const bool regular = (double)frames * 48000.0 < 512.0 * (double)sr;
        const int lay_regular[4] = {0, 1, 1, 2}, lay_long[4] = {0, 3, 1, -1};  // loudness, waveform, stereometer, oscilloscope
        const int* lay = regular ? lay_regular : lay_long;
        int lay_env[4];
        if (const char* e = tuning_env(regular ? "OMX_GROUP_LAYOUT" : "OMX_GROUP_LAYOUT_LONG")) {  // tuning hook: "ld,wf,st,sc" (side stream, -1 = the caller's)
            if (std::sscanf(e, "%d,%d,%d,%d", &lay_env[0], &lay_env[1], &lay_env[2], &lay_env[3]) == 4) lay = lay_env;
        }
        auto on = [&](int k) -> hipStream_t {
            if (k < 0) return stream;
            if (!used[k]) {
                used[k] = true;
                OMX_HIP(hipStreamWaitEvent(side_[k], fork_, 0));
            }
            return side_[k];
        };
        // ---- one side stream per meter bank (round 4: until then loudness + waveform and stereometer + oscilloscope shared one each, and
        //      at the reference's cadence — one 256-frame block per call, kernels of 50 ... 100 us on a few dozen workgroups — the
        //      longer pair was the call's critical path)
        // ---- side stream 0: loudness (+ its summary columns)
        if (loudness) {
            const hipStream_t s_ld = on(lay[0]);
            {
                const int rc = loudness->process(d_pcm, true, block, n_blocks, channels, sample_rate, positions, s_ld, &up.d_loudness);
                note(rc, OMX_VISUAL_LOUDNESS);
                if (stats && rc == OMX_PRODUCED && up.d_loudness) {
                    // K9: true-peak bars + their peak holds on the sample clock (loudness/state.rs:36-60, 178-217)
                    if (!holds_valid_) {
                        holds_.reserve((size_t)S * 3);
                        launch_peak_holds_reset(holds_.ptr, (uint64_t)S * 3, clock_, s_ld);
                        holds_valid_ = true;
                    }
                    meters_.reserve((size_t)S * n_blocks);
                    const double dt = (double)block / (double)sr;
                    launch_loudness_meters(up.d_loudness, S, n_blocks, OMX_METER_TRUE_PEAK, OMX_METER_LUFS_SHORT_TERM, clock_, dt, holds_.ptr,
                                           meters_.ptr, s_ld);
                    clock_ += (double)n_blocks * dt;
                    launch_stats_loudness(up.d_loudness, meters_.ptr, S, n_blocks, channels, rows_.ptr, s_ld);
                    stats_written |= kStatsLoudnessColumns;
                }
            }
            if (stats && !(stats_written & kStatsLoudnessColumns)) launch_stats_clear_columns(rows_.ptr, S, kStatsLoudnessColumns, s_ld);
            OMX_HIP(hipGetLastError());
        }
        // ---- side stream 3: waveform
        if (waveform) {
            const hipStream_t s_wf = on(lay[1]);
            note(waveform->process(d_pcm, true, frames, channels, sample_rate, positions, s_wf, &up.waveform), OMX_VISUAL_WAVEFORM);
            OMX_HIP(hipGetLastError());
        }
        // ---- side stream 1: stereometer (+ its summary columns)
        if (stereometer) {
            const hipStream_t s_st = on(lay[2]);
            {
                const int rc = stereometer->process(d_pcm, true, block, n_blocks, channels, sample_rate, positions, s_st, &up.stereometer);
                note(rc, OMX_VISUAL_STEREOMETER);
                if (stats && rc == OMX_PRODUCED && up.stereometer.d_correlations) {
                    launch_stats_stereometer(up.stereometer.d_correlations, S, n_blocks, rows_.ptr, s_st);
                    stats_written |= kStatsStereometerColumns;
                }
                if (stats && !(stats_written & kStatsStereometerColumns)) launch_stats_clear_columns(rows_.ptr, S, kStatsStereometerColumns, s_st);
            }
            OMX_HIP(hipGetLastError());
        }
        // ---- side stream 2: oscilloscope
        // ---- oscilloscope: on the CALLER's stream, behind the spectrogram / spectrum kernels.  HIP maps streams onto four hardware
        //      queues: with a side stream of its own (round 4) the fifth stream shared a queue with the waveform bank's and the trigger pass
        //      — one 512-thread workgroup per CU holding > 100 KiB of LDS — started only when that was through (kernel trace of the
        //      streaming cadence: scope chain 155 -> 276 us into the call).  More queues (GPU_MAX_HW_QUEUES = 8) measured WORSE, 353 against
        //      307 us per call: every chain contends for the same 256 CUs.  Here: 291 us, and 466 against 530 for a 1024-frame chunk.
        if (oscilloscope) {
            {
                const int rc = oscilloscope->process(d_pcm, true, block, n_blocks, channels, sample_rate, positions, on(lay[3]));
                note(rc, OMX_VISUAL_OSCILLOSCOPE);
                if (rc == OMX_PRODUCED) {
                    up.oscilloscope.n_streams = S;
                    up.oscilloscope.n_blocks = n_blocks;
                    up.oscilloscope.epoch = oscilloscope->epoch();
                    up.oscilloscope.sample_stride = kScopeTarget;
                    up.oscilloscope.d_headers = reinterpret_cast<const omx_oscilloscope_block_header*>(oscilloscope->d_headers());
                    up.oscilloscope.d_samples = oscilloscope->d_samples();
                }
            }
            OMX_HIP(hipGetLastError());
        }
        return (int)OMX_NONE;
    });
    // ---- joined
    if (stats) {
        OMX_HIP(hipGetLastError());
        up.d_stats_rows = rows_.ptr;
    }
    if (out) *out = up;
    if (worst < 0) return worst;
    return up.produced ? OMX_PRODUCED : OMX_NONE;
}

// Per-capture ingest: capture s delivers frames[s] <= frames_capacity frames this call (0 = nothing arrived) and is reset first where
// reset_mask[s] != 0 — one VisualManager per capture in the reference, each fed by its own DspBatcher and reset on its own
// (registry.rs:360-365, :396-418; meter.rs:27-80).  Every enabled bank takes the call through its own ragged entry point (per-stream
// positions on the device).  What capture s delivers is ONE block (registry.rs:407-417: one AudioBlock per ingest_samples call, whatever
// the chunk length — the block-based banks go through process_chunks); with cfg.block_frames != 0 it is frames[s] / block_frames blocks
// of block_frames frames (a host replaying queued quanta), and the counts must then be multiples of it.
int CaptureGroup::ingest_ragged(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels_in,
                                float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                omx_capture_group_ragged_update* out) {
    HostLap host_lap;
    (void)host_lap;
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    const uint32_t S = cfg_.n_streams;
    omx_capture_group_ragged_update up;
    std::memset(&up, 0, sizeof(up));
    const bool chunks = cfg_.block_frames == 0;  // the reference's partition: one block per capture per call
    const uint64_t block = chunks ? frames_capacity : cfg_.block_frames;
    if (frames_capacity == 0 || frames_capacity % block != 0) {
        set_last_error("capture group ingest_ragged: frames_capacity must be a positive multiple of cfg.block_frames");
        return OMX_ERR_INVALID;
    }
    const uint64_t max_blocks = chunks ? 1 : frames_capacity / block;
    blocks_scratch_.resize(S);
    for (uint32_t s = 0; s < S; ++s) {
        if (frames[s] > frames_capacity || (!chunks && frames[s] % block != 0)) {
            set_last_error(chunks ? "capture group ingest_ragged: frames[s] > frames_capacity"
                                  : "capture group ingest_ragged: frames[s] must be a multiple of cfg.block_frames, at most frames_capacity");
            return OMX_ERR_INVALID;
        }
        blocks_scratch_[s] = chunks ? (frames[s] != 0 ? 1u : 0u) : (uint32_t)(frames[s] / block);
    }
    up.block_frames = chunks ? 0 : block;
    up.max_blocks = max_blocks;
    auto bank_on = [&](uint32_t bit) { return (enabled_ & bit) != 0; };
    SpectrogramBank* spectrogram = bank_on(OMX_VISUAL_SPECTROGRAM) ? spectrogram_.get() : nullptr;
    SpectrumBank* spectrum = bank_on(OMX_VISUAL_SPECTRUM) ? spectrum_.get() : nullptr;
    LoudnessBank* loudness = bank_on(OMX_VISUAL_LOUDNESS) ? loudness_.get() : nullptr;
    StereometerBank* stereometer = bank_on(OMX_VISUAL_STEREOMETER) ? stereometer_.get() : nullptr;
    OscilloscopeBank* oscilloscope = bank_on(OMX_VISUAL_OSCILLOSCOPE) ? oscilloscope_.get() : nullptr;
    WaveformBank* waveform = bank_on(OMX_VISUAL_WAVEFORM) ? waveform_.get() : nullptr;
    // per bank: the caller's mask plus the resets it missed while disabled; disabled banks that exist record this call's mask
    const uint8_t* m_sg = mask_for(0, spectrogram != nullptr, spectrogram_ != nullptr, reset_mask);
    const uint8_t* m_sp = mask_for(1, spectrum != nullptr, spectrum_ != nullptr, reset_mask);
    const uint8_t* m_ld = mask_for(2, loudness != nullptr, loudness_ != nullptr, reset_mask);
    const uint8_t* m_st = mask_for(3, stereometer != nullptr, stereometer_ != nullptr, reset_mask);
    const uint8_t* m_os = mask_for(4, oscilloscope != nullptr, oscilloscope_ != nullptr, reset_mask);
    const uint8_t* m_wf = mask_for(5, waveform != nullptr, waveform_ != nullptr, reset_mask);
    ragged_ = true;
    // Summary rows per capture: the rows and the peak holds persist between calls (a capture that delivers nothing keeps its row), the
    // holds run on every capture's own sample clock, and a per-capture reset restarts that capture's holds (LoudnessState::reset_audio).
    // A reset that arrives while the Loudness visual is disabled is applied by the first stats pass after it is enabled again: the mask
    // the bank gets (m_ld) carries it.
    const bool stats = stats_ && spectrogram && loudness && stereometer;
    if (stats && !ragged_stats_live_) {
        rows_.reserve((size_t)S * OMX_STATS_COLUMNS);
        holds_.reserve((size_t)S * 3);
        clocks_.reserve(S);
        if (!holds_valid_) {  // nothing carried over from lock-step calls
            OMX_HIP(hipMemsetAsync(rows_.ptr, 0, (size_t)S * OMX_STATS_COLUMNS * sizeof(float), stream));
            launch_peak_holds_reset(holds_.ptr, (uint64_t)S * 3, 0.0, stream);
            clock_ = 0.0;
        }
        launch_fill_f64(clocks_.ptr, S, clock_, stream);  // every capture continues the common clock of the lock-step calls so far
        OMX_HIP(hipGetLastError());
        holds_valid_ = true;
        ragged_stats_live_ = true;
    }
    const float sr = sanitize_sample_rate(sample_rate);
    int worst = OMX_NONE;
    auto note = [&](int rc, uint32_t bit) {
        if (rc < 0) worst = worst < 0 ? worst : rc;
        else if (rc == OMX_PRODUCED) up.produced |= bit;
        if (rc >= 0 && bit) {  // the bank has taken its mask: the resets it had missed while disabled are delivered
            int vi = 0;
            while ((1u << vi) != bit) ++vi;
            if (mask_merged_[vi]) {
                pending_reset_[vi].clear();
                mask_merged_[vi] = false;
            }
        }
    };
    bool used[kSideStreams] = {false, false, false, false};
    forked(stream, side_, fork_, join_, used, [&] {
        OMX_LAP(0);
        // ---- the caller's stream: the banks that keep pending audio.  Both plan their per-capture pushes on the device (skip / count /
        //      head per stream); ONE projection launch then feeds the rings of both (registry.rs:407-417: one AudioBlock, every visual)
        if (spectrogram && spectrum && shared_ingest_) {
            IngestArgs parts[2];
            const int rc_sg = spectrogram->ragged_plan(d_pcm, frames_capacity, frames, m_sg, channels, sample_rate, positions, stream, parts[0]);
            OMX_LAP(7);
            if (rc_sg < 0) note(rc_sg, 0);
            if (rc_sg >= 0) {
                // The spectrogram's plan has advanced its per-capture positions: from here on its samples MUST be written and its columns
                // computed, whatever happens to the spectrum's plan (an error code, an exception from a reservation) — otherwise its rings
                // would claim samples nobody wrote.
                int rc_sp = OMX_NONE;
                bool sp_threw = false;
                BackendError sp_error{OMX_ERR_BACKEND};
                try {
                    rc_sp = spectrum->ragged_plan(d_pcm, frames_capacity, frames, m_sp, channels, sample_rate, positions, stream, parts[1]);
                } catch (const BackendError& e) {
                    sp_threw = true;
                    sp_error = e;
                }
                OMX_LAP(8);
                if (rc_sp < 0) note(rc_sp, 0);
                const int n_parts = (!sp_threw && rc_sp == OMX_PRODUCED) ? 2 : 1;  // (a Spectrum bank without an active trace takes no samples)
                if (launch_ingest_ragged_parts(parts, n_parts, S, stream)) {
                    up.ingest_launches += 1;
                } else {  // other channel counts: the banks' own launches
                    launch_ingest(parts[0], S, stream);
                    up.ingest_launches += 1;
                    if (n_parts == 2) {
                        launch_ingest(parts[1], S, stream);
                        up.ingest_launches += 1;
                    }
                }
                OMX_HIP(hipGetLastError());
                OMX_LAP(9);
                note(spectrogram->ragged_finish(stream, &up.spectrogram), OMX_VISUAL_SPECTROGRAM);
                if (stats && up.spectrogram.d_n_columns)  // (inside the fork, as in the lock-step call)
                    launch_stats_spectrogram_ragged(up.spectrogram.d_counts, S, up.spectrogram.max_columns, up.spectrogram.d_n_columns, rows_.ptr, stream);
                OMX_LAP(10);
                if (n_parts == 2) note(spectrum->ragged_finish(stream, &up.spectrum), OMX_VISUAL_SPECTRUM);
                if (sp_threw) throw sp_error;  // (the message set_last_error recorded stands)
            }
        } else {
            if (spectrogram) {
                note(spectrogram->process_ragged(d_pcm, frames_capacity, frames, m_sg, channels, sample_rate, positions, stream, &up.spectrogram),
                     OMX_VISUAL_SPECTROGRAM);
                up.ingest_launches += 1;
                if (stats && up.spectrogram.d_n_columns)
                    launch_stats_spectrogram_ragged(up.spectrogram.d_counts, S, up.spectrogram.max_columns, up.spectrogram.d_n_columns, rows_.ptr, stream);
            }
            if (spectrum) {
                note(spectrum->process_ragged(d_pcm, frames_capacity, frames, m_sp, channels, sample_rate, positions, stream, &up.spectrum),
                     OMX_VISUAL_SPECTRUM);
                up.ingest_launches += 1;
            }
        }
        OMX_LAP(1);
        // stream layout by call length, as in the lock-step call (see there)
        const bool regular = (double)frames_capacity * 48000.0 < 512.0 * (double)sr;
        const int lay_regular[4] = {0, 1, 1, 2}, lay_long[4] = {0, 3, 1, -1};  // loudness, waveform, stereometer, oscilloscope
        const int* lay = regular ? lay_regular : lay_long;
        int lay_env[4];
        if (const char* e = tuning_env(regular ? "OMX_GROUP_LAYOUT" : "OMX_GROUP_LAYOUT_LONG")) {  // tuning hook: "ld,wf,st,sc" (side stream, -1 = the caller's)
            if (std::sscanf(e, "%d,%d,%d,%d", &lay_env[0], &lay_env[1], &lay_env[2], &lay_env[3]) == 4) lay = lay_env;
        }
        auto on = [&](int k) -> hipStream_t {
            if (k < 0) return stream;
            if (!used[k]) {
                used[k] = true;
                OMX_HIP(hipStreamWaitEvent(side_[k], fork_, 0));
            }
            return side_[k];
        };
        if (loudness) {
            const hipStream_t s_ld = on(lay[0]);
            note(chunks ? loudness->process_chunks(d_pcm, frames_capacity, frames, m_ld, channels, sample_rate, positions, s_ld, &up.loudness)
                        : loudness->process_ragged(d_pcm, block, max_blocks, blocks_scratch_.data(), m_ld, channels, sample_rate, positions,
                                                   s_ld, &up.loudness),
                 OMX_VISUAL_LOUDNESS);
            if (stats && up.loudness.d_snapshots && up.loudness.d_n_blocks)
                launch_stats_loudness_ragged(up.loudness.d_snapshots, S, up.loudness.max_blocks, up.loudness.d_n_blocks, up.loudness.d_block_frames,
                                             (uint32_t)block, sr, up.loudness.d_reset, OMX_METER_TRUE_PEAK, OMX_METER_LUFS_SHORT_TERM, channels,
                                             holds_.ptr, clocks_.ptr, rows_.ptr, s_ld);
            OMX_HIP(hipGetLastError());
        }
        OMX_LAP(2);
        if (waveform) {
            const hipStream_t s_wf = on(lay[1]);
            note(waveform->process_ragged(d_pcm, frames_capacity, frames, m_wf, channels, sample_rate, positions, s_wf, &up.waveform),
                 OMX_VISUAL_WAVEFORM);
            OMX_HIP(hipGetLastError());
        }
        OMX_LAP(3);
        if (stereometer) {
            const hipStream_t s_st = on(lay[2]);
            note(chunks ? stereometer->process_chunks(d_pcm, frames_capacity, frames, m_st, channels, sample_rate, positions, s_st,
                                                      &up.stereometer)
                        : stereometer->process_ragged(d_pcm, block, max_blocks, blocks_scratch_.data(), m_st, channels, sample_rate, positions,
                                                      s_st, &up.stereometer),
                 OMX_VISUAL_STEREOMETER);
            if (stats && up.stereometer.d_correlations && up.stereometer.d_n_blocks)
                launch_stats_stereometer_ragged(up.stereometer.d_correlations, S, up.stereometer.max_blocks, up.stereometer.d_n_blocks, rows_.ptr,
                                                s_st);
            OMX_HIP(hipGetLastError());
        }
        OMX_LAP(4);
        if (oscilloscope) {
            const hipStream_t s_sc = on(lay[3]);
            note(chunks ? oscilloscope->process_chunks(d_pcm, frames_capacity, frames, m_os, channels, sample_rate, positions, s_sc,
                                                       &up.oscilloscope)
                        : oscilloscope->process_ragged(d_pcm, block, max_blocks, blocks_scratch_.data(), m_os, channels, sample_rate, positions,
                                                       s_sc, &up.oscilloscope),
                 OMX_VISUAL_OSCILLOSCOPE);
            OMX_HIP(hipGetLastError());
        }
        OMX_LAP(5);
        return (int)OMX_NONE;
    });
    // ---- joined
    if (stats) {
        OMX_HIP(hipGetLastError());
        up.d_stats_rows = rows_.ptr;
    }
    if (out) *out = up;
    OMX_LAP(6);
#ifdef OMX_TUNING
    ++g_host_sections.calls;
#endif
    if (worst < 0) return worst;
    return up.produced ? OMX_PRODUCED : OMX_NONE;
}

}  // namespace omx
