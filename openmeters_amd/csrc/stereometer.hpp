// StereometerBank: S independent StereometerProcessors (reference
// src/visuals/stereometer/processor.rs:64-208).  Four lanes per stream: lane 0 = full band, lanes 1-3 =
// low / mid / high of the LR4 split (reference src/dsp.rs:473-504, ThreeBand<[Cascade<Biquad,2>;2], true>).
#pragma once
#include "common.hpp"

namespace omx {

struct BiquadCoef {  // reference src/dsp.rs:379-420
    float b[3];
    float a[2];
};
BiquadCoef make_biquad(bool highpass, float sample_rate, float frequency);

struct StereoLaneState {
    float z[2][2][2][2];  // [stage A/B][cascade element][channel L/R][z0,z1]
    double moments[3];    // Correlator: cross, left power, right power (:34-61)
};

struct StereometerArgs {
    const float* pcm;  // [n_streams][frames_total][channels]
    uint64_t frames_total;
    uint32_t block_frames, n_blocks, n_streams;
    AudioFormatArgs fmt;
    BiquadCoef stage_a[4], stage_b[4];  // per lane (lane 0 unused)
    uint32_t use_a[4], use_b[4];
    uint32_t analyze_bands, emit_band_points;
    double alpha;
    StereoLaneState* state;  // [n_streams][4]
    float* history;          // [n_streams][4][hist_frames][2] ring of the newest pairs
    uint32_t hist_frames;
    uint64_t hist_pos[4];    // absolute pair count pushed so far per band
    float* correlations;     // [n_streams][n_blocks][4]
};
void launch_stereometer(const StereometerArgs& a, hipStream_t stream);
// points[s][band][i] = history pair (oldest + i*frames/target), band points scaled by 0.8 (:152-170)
void launch_stereometer_rehome(const float* from, float* to, uint32_t n_streams, uint32_t from_frames, uint32_t to_frames,
                               const uint64_t hist_pos[4], const uint64_t keep[4], hipStream_t stream);
void launch_stereometer_produced(uint32_t* out, uint32_t n_streams, uint32_t n_blocks, uint64_t len_before, uint32_t block_frames,
                                 uint32_t frames, hipStream_t stream);
void launch_stereometer_points(const float* history, uint32_t n_streams, uint32_t hist_frames, const uint64_t hist_pos[4],
                               const uint32_t band_valid[4], uint32_t target, float* points, hipStream_t stream);

void stereometer_config_default(omx_stereometer_config* c);

class StereometerBank {
public:
    StereometerBank(const omx_stereometer_config& cfg, uint32_t n_streams);
    const omx_stereometer_config& config() const { return cfg_; }
    void update_config(const omx_stereometer_config& cfg);
    void reset_audio();
    int process(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels,
                float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                omx_stereometer_bank_update* out);
    int fetch(uint64_t stream_index, uint64_t block, float correlations[4], uint32_t* produced, hipStream_t stream);
    int fetch_points(uint64_t stream_index, uint32_t band, float* dst, uint64_t* n_pairs, hipStream_t stream);
    hipStream_t last_stream() const { return last_stream_; }
    uint32_t target() const { return last_target_; }

private:
    void init(const omx_stereometer_config& cfg);
    void clear_filters(hipStream_t stream, bool bands_only);
    uint32_t segment_frames() const;

    omx_stereometer_config cfg_{};
    uint32_t n_streams_;
    uint32_t history_channels_ = 0;
    double alpha_ = 0.0;
    uint64_t hist_len_[4] = {0, 0, 0, 0};  // VecDeque lengths (uniform over streams)
    uint64_t hist_pos_[4] = {0, 0, 0, 0};
    uint32_t hist_frames_ = 0, last_target_ = 0;
    uint64_t last_blocks_ = 0;
    bool pending_full_reset_ = true, pending_band_reset_ = false;
    uint32_t band_valid_[4] = {0, 0, 0, 0};
    std::vector<uint32_t> produced_host_;
    DeviceBuffer<StereoLaneState> state_;
    DeviceBuffer<float> history_, history_next_, correlations_, points_, staging_;
    DeviceBuffer<uint32_t> produced_;
    hipStream_t last_stream_ = nullptr;
};

}  // namespace omx
