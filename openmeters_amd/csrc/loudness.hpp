// LoudnessBank: S independent LoudnessProcessors (reference src/visuals/loudness/processor.rs:218-312),
// four lanes per (stream, channel): register pipelines with per-lane window / true-peak phase.
#pragma once
#include "common.hpp"

namespace omx {

constexpr int kLoudnessWindows = 4;
constexpr int kTruePeakMaxDelay = 24;

// Per-channel recurrent state (array of structs in HBM; loaded to registers for the whole call).
struct LoudnessChannelState {
    double sums[kLoudnessWindows][2];         // CompensatedPair::sums   [window][0 = live window, 1 = since refresh]
    double corrections[kLoudnessWindows][2];  // CompensatedPair::corrections
    double filter[4];                         // K-weighting TDF-II state
    float delay[kTruePeakMaxDelay];           // true-peak delay line, delay[0] = newest sample
    float peak;
    float _pad;
};

struct LoudnessArgs {
    const float* pcm;       // [n_streams][frames_total][channels]
    uint64_t frames_total;  // block_frames * n_blocks
    uint32_t block_frames, n_blocks;
    uint32_t n_streams, channels;
    double b[5], a[5];      // K-weighting (loudness/processor.rs:22-55)
    double weights[OMX_MAX_CHANNELS];  // channel_weight(position) (:174-183)
    uint8_t positions[OMX_MAX_CHANNELS];
    float fir4[12][3];      // TRUE_PEAK_FIRS.0 (:90-97)
    float fir2[24];         // TRUE_PEAK_FIRS.1
    uint32_t delay_len;     // 12 (4x), 24 (2x) or 0
    uint64_t capacities[kLoudnessWindows];
    uint64_t ring_len;      // longest capacity
    uint64_t frames_seen;   // pushes since the state was created (head = frames_seen % ring_len)
    uint32_t slot_shift;    // log2 of the (stream, channel) slots per stream: 8, or the channel count itself when it is 1 / 2 / 4
                            // (a 2-channel bank would otherwise spend three quarters of its lanes and workgroups on dead slots)
    double* ring;           // [slot group of 64][ring_len][64] squared K-weighted samples
    LoudnessChannelState* state;  // [n_streams << slot_shift] (allocated for 8 slots per stream)
    float floor_db;
    omx_loudness_snapshot* snapshots;  // [n_streams][n_blocks]
    uint32_t n_meter_blocks;
    uint32_t role_perm;  // TEMP           // split launch: workgroups [0, n) = K-weighting + windows, [n, 2n) = true peak
};
void launch_loudness(const LoudnessArgs& a, hipStream_t stream);

void loudness_config_default(omx_loudness_config* c);
void k_weighting_coefficients(double fs, double b[5], double a[5]);

class LoudnessBank {
public:
    LoudnessBank(const omx_loudness_config& cfg, uint32_t n_streams);
    void reset_audio();
    int process(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels,
                float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                const omx_loudness_snapshot** d_snapshots);
    int fetch(uint64_t stream_index, uint64_t block, omx_loudness_snapshot* dst, hipStream_t stream);
    EventTimer& timer() { return timer_; }
    hipStream_t last_stream() const { return last_stream_; }

private:
    void ensure_state(uint32_t channels, float sample_rate, hipStream_t stream);
    void clear_state(hipStream_t stream);

    omx_loudness_config cfg_{};
    uint32_t n_streams_;
    uint32_t channels_ = 0;  // 0 = no channel state yet (reference: channels.len())
    double b_[5], a_[5];
    uint64_t frames_seen_ = 0, ring_len_ = 0, last_blocks_ = 0;
    bool state_clean_ = false;
    DeviceBuffer<double> ring_;
    DeviceBuffer<LoudnessChannelState> state_;
    DeviceBuffer<omx_loudness_snapshot> snapshots_;
    DeviceBuffer<float> staging_;
    EventTimer timer_;
    hipStream_t last_stream_ = nullptr;
};

}  // namespace omx
