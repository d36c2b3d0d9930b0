// CaptureGroup: VisualManager::ingest_samples (reference src/visuals/registry.rs:396-418) for S captures in lock step — one block to
// every enabled visual's bank, one shared projection for the banks that keep pending audio, meter banks on side streams, summary rows.
#pragma once
#include <memory>
#include <vector>

#include "loudness.hpp"
#include "oscilloscope.hpp"
#include "spectrogram.hpp"
#include "spectrum.hpp"
#include "stereometer.hpp"
#include "summary.hpp"
#include "waveform.hpp"

namespace omx {

void capture_group_config_default(omx_capture_group_config* c);

constexpr int kSideStreams = 4;  // one per meter bank

class CaptureGroup {
public:
    explicit CaptureGroup(const omx_capture_group_config& cfg);
    ~CaptureGroup();
    CaptureGroup(const CaptureGroup&) = delete;
    CaptureGroup& operator=(const CaptureGroup&) = delete;
    void reset_audio();
    // VisualManager::set_enabled (registry.rs:272-277, :349-352): a disabled visual is skipped by ingest and keeps its state; enabling
    // one prepares it (its bank is created here on first use)
    int set_enabled(uint32_t visual, bool on);
    uint32_t enabled() const { return enabled_; }
    // Entry::apply_settings -> processor.update_config (registry.rs:54-58, :266-270): `config` is the omx_<visual>_config of that visual
    int update_config(uint32_t visual, const void* config, hipStream_t stream);
    // ingest_samples' format-generation rule (registry.rs:400-406): a generation that differs from the last one resets every visual
    bool note_format_generation(uint64_t generation);
    // per-capture frame counts and reset flags (one VisualManager per capture in the reference, each fed and reset on its own)
    int ingest_ragged(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels,
                      float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_capture_group_ragged_update* out);
    void set_stats(bool on) { stats_ = on; }
    void set_shared_ingest(bool on) { shared_ingest_ = on; }
    void set_timing(bool on);
    int ingest(const float* d_pcm, uint64_t frames, uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS],
               hipStream_t stream, omx_capture_group_update* out);
    SpectrogramBank* spectrogram() { return spectrogram_.get(); }

private:
    void ensure_bank(uint32_t visual);
    omx_capture_group_config cfg_;
    uint32_t enabled_ = 0;       // OMX_VISUAL_* bits ingest feeds
    bool ragged_ = false;        // per-capture positions (after the first ingest_ragged, until reset_audio)
    bool have_generation_ = false;
    uint64_t generation_ = 0;
    std::vector<uint32_t> blocks_scratch_;
    // per-capture resets a disabled visual has not seen yet (VisualManager::reset_audio resets every entry, enabled or not,
    // registry.rs:360-365): OR-ed into the bank's next ragged call; index = log2(OMX_VISUAL_* bit)
    std::vector<uint8_t> pending_reset_[6];
    std::vector<uint8_t> mask_scratch_[6];
    bool mask_merged_[6] = {false, false, false, false, false, false};  // mask_for merged pending resets into this call's mask (cleared on success)
    const uint8_t* mask_for(int visual_index, bool bank_enabled, bool bank_exists, const uint8_t* reset_mask);
    std::unique_ptr<SpectrogramBank> spectrogram_;
    std::unique_ptr<SpectrumBank> spectrum_;
    std::unique_ptr<LoudnessBank> loudness_;
    std::unique_ptr<StereometerBank> stereometer_;
    std::unique_ptr<OscilloscopeBank> oscilloscope_;
    std::unique_ptr<WaveformBank> waveform_;
    hipStream_t side_[kSideStreams] = {nullptr, nullptr, nullptr, nullptr};  // loudness, stereometer, oscilloscope, waveform
    hipEvent_t fork_ = nullptr, join_[kSideStreams] = {nullptr, nullptr, nullptr, nullptr};
    bool stats_ = false, shared_ingest_ = true;
    // K9 state of the summary rows: the three peak holds per stream and the sample clock of the next applied snapshot
    DeviceBuffer<omx_peak_hold> holds_;
    bool holds_valid_ = false;
    double clock_ = 0.0;
    DeviceBuffer<omx_meter_row> meters_;
    DeviceBuffer<float> rows_;
    // per-capture calls: every capture's own sample clock (device), and whether rows_ / clocks_ already describe per-capture state
    DeviceBuffer<double> clocks_;
    bool ragged_stats_live_ = false;
};

}  // namespace omx
