// SpectrumBank: S independent SpectrumProcessors (reference src/visuals/spectrum/processor.rs:72-323)
// advanced in lock-step.
#pragma once
#include "stft_kernels.hpp"

namespace omx {

void spectrum_config_default(omx_spectrum_config* c);
float a_weight_host(float freq_hz);  // spectrum/processor.rs:410-425

class SpectrumBank {
public:
    SpectrumBank(const omx_spectrum_config& cfg, uint32_t n_streams, bool emit_all_hops);
    const omx_spectrum_config& config() const { return cfg_; }
    void update_config(const omx_spectrum_config& cfg, hipStream_t stream);
    void reset_audio();
    void prepare(hipStream_t stream);
    int process(const float* pcm, bool pcm_on_device, uint64_t frames, uint32_t channels, float sample_rate,
                const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_spectrum_bank_update* out);
    // process() in steps, for callers that run the ingest launch themselves (capture group; see SpectrogramBank::push_begin)
    int push_begin(uint64_t frames, uint32_t channels, float sample_rate, hipStream_t stream, IngestSlots& slots);
    void push_end(const IngestSlots& slots);
    int process_pushed(hipStream_t stream, omx_spectrum_bank_update* out);
    // per-stream frame counts / reset (include/omx.h: omx_spectrum_bank_process_ragged); pcm in device memory
    int process_ragged(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels,
                       float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_spectrum_ragged_update* out);
    // the two halves of process_ragged (a capture group shares one projection launch between the Spectrogram bank and this one);
    // ragged_plan returns OMX_NONE when no trace is active (nothing to ingest, nothing to finish)
    int ragged_plan(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels,
                    float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, IngestArgs& ia_out);
    int ragged_finish(hipStream_t stream, omx_spectrum_ragged_update* out);
    int fetch(uint64_t stream_index, uint64_t hop, float* dst, hipStream_t stream);
    const std::vector<float>& frequency_bins() const { return freq_bins_; }
    uint64_t bins() const { return cfg_.fft_size / 2 + 1; }
    hipStream_t last_stream() const { return last_stream_; }
    void force_generic(bool on) { force_generic_ = on; }
    void host_outputs(bool on) { host_outputs_ = on; }  // single-stream handles: traces in pinned host memory
    EventTimer& timer() { return timer_; }

private:
    void rebuild_fft(hipStream_t stream);
    void reset_buffers(hipStream_t stream);
    void reset_level_buffers(hipStream_t stream);
    void active_traces(bool out[2]) const;
    void ensure_ring(uint64_t incoming, hipStream_t stream);
    void enter_ragged(hipStream_t stream);
    bool carry_applies(uint64_t tail0) const;
    void launch_window_sums_for(uint64_t tail0, const uint64_t* tails, const uint32_t* hops, uint64_t n_hops, uint64_t first_hop, uint32_t n_traces,
                                const bool active[2], hipStream_t stream);
    int launch_hops(uint64_t tail0, const uint64_t* tails, const uint32_t* hops, uint64_t n_hops, uint64_t first_hop, uint32_t n_traces,
                    const bool active[2], hipStream_t stream);

    omx_spectrum_config cfg_{};
    uint32_t n_streams_;
    size_t blu_m_ = 0;  // Bluestein convolution length (0: fft_size is a power of two)
    DeviceBuffer<float> d_blu_chirp_, d_blu_bf_, d_blu_tw_;
    bool emit_all_, prepared_ = false, fast4096_ = false, force_generic_ = false, traces_dirty_ = true;
    uint64_t head_ = 0, tail_ = 0, pending_skip_ = 0, ring_cap_ = 0;
    DeviceBuffer<float> ring_[2];
    HostStage staging_;
    DeviceBuffer<float> d_window_, d_bin_norm_, d_a_weight_, d_freq_bins_, d_tw_fft_, d_tw256_, d_tw4096_, d_workspace_;
    DeviceBuffer<float> d_power_, d_smoothed_, d_hop_sums_, d_carry_;
    // running window folds carried between lock-step calls (launch_window_sums_for)
    bool carry_valid_ = false;
    uint64_t carry_pos_ = 0;
    uint32_t carry_slot0_ = 0;
    OutBuffer<float> d_traces_;
    bool host_outputs_ = false;
    std::vector<float> freq_bins_, a_weight_;
    float state_floor_ = 0.0f;
    uint64_t last_hops_out_ = 0;
    EventTimer timer_;
    hipStream_t last_stream_ = nullptr;
    // ragged mode: per-stream positions on the device (the host-side head_ / tail_ then only bound the pending length)
    uint64_t pend_max_hops_ = 0, pend_hops_out_ = 1;  // between ragged_plan and ragged_finish
    bool ragged_ = false;
    DeviceBuffer<uint64_t> r_head_, r_tail_, r_skip_, r_ing_head_, r_hop_tail_;
    DeviceBuffer<uint32_t> r_ing_skip_, r_ing_count_, r_nhops_;
    DeviceView<uint32_t> r_frames_;  // (views into r_staging_)
    // ragged mode: the carried window folds' bookkeeping per stream (spectrum_plan_kernel)
    DeviceBuffer<uint64_t> r_carry_pos_, r_fold_from_;
    DeviceBuffer<uint32_t> r_carry_slot0_, r_carry_valid_, r_fold_mode_, r_fold_slot0_;
    DeviceView<uint8_t> r_mask_;
    RaggedStaging r_staging_;
};

struct SpectrumSingle {
    SpectrumBank bank;
    std::vector<float> traces;  // [2][2][bins]
    explicit SpectrumSingle(const omx_spectrum_config& c) : bank(c, 1, false) { bank.host_outputs(true); }
    int process_block(const omx_block* block, omx_spectrum_snapshot* out);
};

}  // namespace omx
