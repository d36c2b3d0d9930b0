// CaptureGroup: VisualManager::ingest_samples (reference src/visuals/registry.rs:396-418) for S captures in lock step — one block to
// every enabled visual's bank, one shared projection for the banks that keep pending audio, meter banks on side streams, summary rows.
#pragma once
#include <memory>

#include "loudness.hpp"
#include "oscilloscope.hpp"
#include "spectrogram.hpp"
#include "spectrum.hpp"
#include "stereometer.hpp"
#include "summary.hpp"
#include "waveform.hpp"

namespace omx {

void capture_group_config_default(omx_capture_group_config* c);

class CaptureGroup {
public:
    explicit CaptureGroup(const omx_capture_group_config& cfg);
    ~CaptureGroup();
    CaptureGroup(const CaptureGroup&) = delete;
    CaptureGroup& operator=(const CaptureGroup&) = delete;
    void reset_audio();
    void set_stats(bool on) { stats_ = on; }
    void set_shared_ingest(bool on) { shared_ingest_ = on; }
    void set_timing(bool on);
    int ingest(const float* d_pcm, uint64_t frames, uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS],
               hipStream_t stream, omx_capture_group_update* out);
    SpectrogramBank* spectrogram() { return spectrogram_.get(); }

private:
    omx_capture_group_config cfg_;
    std::unique_ptr<SpectrogramBank> spectrogram_;
    std::unique_ptr<SpectrumBank> spectrum_;
    std::unique_ptr<LoudnessBank> loudness_;
    std::unique_ptr<StereometerBank> stereometer_;
    std::unique_ptr<OscilloscopeBank> oscilloscope_;
    std::unique_ptr<WaveformBank> waveform_;
    hipStream_t side_[2] = {nullptr, nullptr};
    hipEvent_t fork_ = nullptr, join_[2] = {nullptr, nullptr};
    bool stats_ = false, shared_ingest_ = true;
    // K9 state of the summary rows: the three peak holds per stream and the sample clock of the next applied snapshot
    DeviceBuffer<omx_peak_hold> holds_;
    bool holds_valid_ = false;
    double clock_ = 0.0;
    DeviceBuffer<omx_meter_row> meters_;
    DeviceBuffer<float> rows_;
};

}  // namespace omx
