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
    // fallback of the chunk-parallel path (stereometer_chunked.hip): run only when *run_if != 0, starting from state_in
    const uint32_t* run_if;
    const StereoLaneState* state_in;
    // ragged banks (per-stream block counts; nullptr = lock-step; the four-wavefront kernel only): stream s runs blocks_v[s] <= n_blocks
    // blocks, its history rings start at start_v[s][band] (hist_pos above is then unused), its state is cleared first when reset_v[s]
    const uint32_t* blocks_v;
    const uint8_t* reset_v;
    const uint64_t* start_v;  // [n_streams][4], written by stereometer_ragged_plan_kernel
    // chunk calls (process_chunks): stream s's blocks are frames_v[s] frames long (nullptr = block_frames for every stream); its row
    // of `pcm` is frames_total frames long whatever it delivers
    const uint32_t* frames_v;
};
void launch_stereometer(const StereometerArgs& a, hipStream_t stream);
// ragged banks: the bookkeeping of one call for every stream (one lane per stream) — the positions its history pushes start from, the
// new positions and VecDeque lengths (:116, :129, :146-150), the per-block `produced` flags and the bands whose points the call's
// last block yields (:152-170)
struct StereoPlanArgs {
    uint32_t n_streams, max_blocks, block_frames, hist_frames, analyze_bands, emit_band_points;
    const uint32_t* blocks;   // [n_streams]
    const uint32_t* frames;   // [n_streams] per-stream block length (chunk calls), or nullptr = block_frames
    const uint8_t* reset;     // [n_streams]
    uint64_t* pos;            // [n_streams][4] in / out
    uint64_t* len;            // [n_streams][4] in / out
    uint64_t* start;          // [n_streams][4] out
    uint32_t* produced;       // [n_streams][max_blocks] out
    uint32_t* band_valid;     // [n_streams][4] out
    uint32_t zero_len_mask;   // bands whose deques a config change emptied since the last call
};
void launch_stereometer_ragged_plan(const StereoPlanArgs& a, hipStream_t stream);
void launch_stereometer_points_ragged(const float* history, uint32_t n_streams, uint32_t hist_frames, const uint64_t* pos_v,
                                      const uint32_t* band_valid_v, uint32_t target, float* points, hipStream_t stream);

// ---- chunk-parallel evaluation (stereometer_chunked.hip): one chunk = one block of the call, 2-channel input
struct StereoChunkArgs {
    const float* pcm;  // [n_streams][frames_total][2]
    uint64_t frames_total;
    uint32_t block_frames, n_blocks, n_streams;  // chunk length, chunks per stream (= blocks x cpb), streams
    uint32_t cpb;                                // chunks per block (1, 2 or 4): block_frames x cpb = the call's block length
    float m00, m10, m01, m11;  // stereo fold weights: left = (0 + s0 m00) + s1 m10, right = (0 + s0 m01) + s1 m11
    BiquadCoef lp_lo, hp_lo, lp_hi, hp_hi;
    uint32_t analyze_bands, emit_band_points;
    double alpha;
    StereoLaneState* state;  // [n_streams][4]
    float* history;
    uint32_t hist_frames;
    uint64_t hist_pos[4];
    float* correlations;     // [n_streams][n_blocks / cpb][4]: one row per BLOCK
    float* chunk_state;      // [n_streams * n_blocks][3 bands][8 states][2 channels]
    double* chunk_moments;   // [n_streams * n_blocks][4 bands][3]
    uint32_t* bad;           // set when a non-finite sample or filter output was seen: the caller re-runs the sequential kernel
    // ragged calls (nullptr = lock-step): stream s runs blocks_v[s] <= n_blocks blocks, after a reset of its state when reset_v[s] != 0;
    // start_v[s * 4 + band] = the history position of its first frame (the plan kernel's output; hist_pos above is then unused)
    const uint32_t* blocks_v;
    const uint8_t* reset_v;
    const uint64_t* start_v;
};
// d_T: [3][6][8][8] f64, the L-frame zero-input transition of the low / mid / high cascades and its powers 2 ... 32; decay = (1 - alpha)^L
void launch_stereometer_chunked(const StereoChunkArgs& a, const double* d_T, double decay, hipStream_t stream);
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
    // Ragged call (include/omx.h: omx_stereometer_bank_process_ragged): stream s runs n_blocks[s] <= max_blocks blocks (its rows of `d_pcm`
    // are block_frames * max_blocks frames apart); streams flagged in reset_mask get reset_audio() first.  The per-stream history
    // positions and lengths then live on the device; the bank stays ragged until reset_audio() of the whole bank.
    int process_ragged(const float* d_pcm, uint64_t block_frames, uint64_t max_blocks, const uint32_t* n_blocks, const uint8_t* reset_mask,
                       uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                       omx_stereometer_ragged_update* out);
    // Chunk call (include/omx.h: omx_stereometer_bank_process_chunks; VisualManager::ingest_samples, registry.rs:396-418): stream s
    // delivers ONE block of frames[s] <= frames_capacity frames (0 = nothing arrived); one output slot per stream.
    int process_chunks(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels,
                       float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_stereometer_ragged_update* out);
    int fetch(uint64_t stream_index, uint64_t block, float correlations[4], uint32_t* produced, hipStream_t stream);
    int fetch_points(uint64_t stream_index, uint32_t band, float* dst, uint64_t* n_pairs, hipStream_t stream);
    hipStream_t last_stream() const { return last_stream_; }
    uint32_t target() const { return last_target_; }

private:
    void init(const omx_stereometer_config& cfg);
    void clear_filters(hipStream_t stream, bool bands_only);
    struct RaggedCall {
        const uint32_t* n_blocks;
        const uint8_t* reset_mask;
        omx_stereometer_ragged_update* out;
        const uint32_t* frames_v = nullptr;  // chunk calls: per-stream block length, rows of `pcm` row_frames apart
        uint64_t row_frames = 0;
    };
    int process_impl(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels, float sample_rate,
                     const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_stereometer_bank_update* out, const RaggedCall* ragged);
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
    DeviceBuffer<float> history_, history_next_;
    OutBuffer<float> correlations_, points_;
    bool host_outputs_ = false;
    HostStage staging_;
    DeviceBuffer<uint32_t> produced_;
    // chunk-parallel path
    DeviceBuffer<float> chunk_state_;
    DeviceBuffer<double> chunk_moments_, transition_;
    DeviceBuffer<uint32_t> bad_;
    DeviceBuffer<StereoLaneState> state_backup_;
    // ragged mode: per-stream history positions / lengths on the device
    bool ragged_ = false;
    uint32_t ragged_zero_mask_ = 0;  // bands whose deques a config change emptied (applied by the next plan kernel)
    DeviceBuffer<uint64_t> r_pos_, r_len_, r_start_;
    DeviceBuffer<uint32_t> r_valid_;
    DeviceView<uint32_t> r_blocks_, r_frames_;  // (views into r_staging_)
    DeviceView<uint8_t> r_mask_;
    std::vector<uint32_t> h_blocks_;
    RaggedStaging r_staging_;
    float transition_rate_ = 0.0f;
    uint64_t transition_frames_ = 0;
    int last_form_ = 0;      // 1 = sequential kernels, 2 = chunk-parallel (omx_debug_stereometer_bank_last_form)
    int chunked_mode_ = -1;  // -1 = choose by shape, 0 = never, 1 = whenever the shape allows (OMX_OPT_KERNEL_FORM)
public:
    void chunked_mode(int mode) { chunked_mode_ = mode; }
    int last_form() const { return last_form_; }
    void host_outputs(bool on) { host_outputs_ = on; }  // single-stream handles: correlations / points in pinned host memory
private:
    hipStream_t last_stream_ = nullptr;
};

}  // namespace omx
