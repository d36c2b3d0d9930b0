// SpectrogramBank: S independent SpectrogramProcessors (reference
// src/visuals/spectrogram/processor.rs:170-544) advanced in lock-step, one HIP launch per call.
#pragma once
#include "stft_kernels.hpp"

namespace omx {

void spectrogram_config_default(omx_spectrogram_config* c);
uint64_t col_byte_stride(uint32_t kind, uint32_t points);
uint64_t history_columns(uint32_t kind, uint32_t points, uint64_t requested);
uint16_t pack_classic_db_host(float db);

class SpectrogramBank {
public:
    SpectrogramBank(const omx_spectrogram_config& cfg, uint32_t n_streams);
    const omx_spectrogram_config& config() const { return cfg_; }
    void update_config(const omx_spectrogram_config& cfg, hipStream_t stream);
    void reset_audio();
    void prepare(hipStream_t stream);
    int process(const float* pcm, bool pcm_on_device, uint64_t frames, uint32_t channels, float sample_rate,
                const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_spectrogram_bank_update* out);
    // process() in steps (see spectrogram.cpp): for callers that run the ingest launch themselves (capture group)
    int push_begin(uint64_t frames, uint32_t channels, float sample_rate, hipStream_t stream, IngestSlots& slots);
    void push_end(const IngestSlots& slots);
    int process_pushed(hipStream_t stream, omx_spectrogram_bank_update* out);
    // Ragged call: stream s receives frames[s] <= frames_capacity new frames (its rows of `pcm` are frames_capacity frames apart),
    // streams flagged in reset_mask get reset_audio() first.  The per-stream positions then live on the device
    // (spectrogram_plan_kernel); the bank stays in ragged mode until reset_audio() of the whole bank.
    int process_ragged(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels,
                       float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_spectrogram_ragged_update* out);
    // the two halves of process_ragged (a capture group shares one projection launch between this bank and the Spectrum bank)
    int ragged_plan(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels,
                    float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, IngestArgs& ia_out);
    int ragged_finish(hipStream_t stream, omx_spectrogram_ragged_update* out);
    int fetch_column(uint64_t stream_index, uint64_t column, void* dst, uint64_t cap, uint64_t* n_out, hipStream_t stream);
    EventTimer& timer() { return timer_; }
    void force_generic(bool on) { force_generic_ = on; }
    void kernel_form(int form) { kernel_form_ = form; }
    // single-stream handles: outputs of at most `limit_bytes` go to pinned host memory (read by the host after one stream sync)
    void host_outputs(size_t limit_bytes) { host_output_limit_ = limit_bytes; }
    bool outputs_on_host() const { return d_counts_.pinned; }
    hipStream_t last_stream() const { return last_stream_; }

private:
    void rebuild_fft(hipStream_t stream);
    void launch_columns(uint64_t n_cols, uint64_t tail, const uint64_t* tails, const uint32_t* cols, hipStream_t stream);
    void enter_ragged(hipStream_t stream);
    void ensure_ring(uint64_t incoming, hipStream_t stream);
    void drain(uint64_t count);
    void advance(uint64_t count);
    void clear_last_nonzero(hipStream_t stream);

    omx_spectrogram_config cfg_{};
    uint32_t n_streams_;
    int kernel_form_ = 0;  // OMX_OPT_KERNEL_FORM
    bool prepared_ = false, reset_ = true, fast4096_ = false, fast_zp_ = false, fast_zpr_ = false, classic_zpr_ = false, force_generic_ = false;
    size_t fft_size_ = 0, hilbert_len_ = 0;
    float power_scale_ = 1.0f;
    // pending audio: absolute sample counters shared by all streams (lock-step pushes)
    uint64_t head_ = 0, tail_ = 0, pending_skip_ = 0, ring_cap_ = 0;
    DeviceBuffer<float> ring_;
    HostStage staging_;
    DeviceBuffer<long long> last_nonzero_, partial_nonzero_;
    DeviceBuffer<float> d_window_, d_dwindow_, d_twindow_, d_bin_norm_, d_col_sums_;
    DeviceBuffer<float> d_tw_fft_, d_tw_hilbert_, d_tw256_, d_tw4096_, d_tw8192_, d_twF_, d_workspace_;
    DeviceBuffer<float> d_blu_chirp_, d_blu_bf_, d_blu_tw_;  // Bluestein tables when the transform length is not a power of two
    size_t blu_m_ = 0;
    OutBuffer<omx_spectrogram_point> d_points_;
    OutBuffer<uint32_t> d_counts_;
    OutBuffer<uint16_t> d_codes_;
    size_t host_output_limit_ = 0;
    // ragged mode: per-stream positions on the device
    uint64_t pend_max_cols_ = 0;  // between ragged_plan and ragged_finish
    bool ragged_ = false;
    uint64_t ragged_pending_bound_ = 0;  // ragged mode: the most pending samples a stream can hold beyond read_len - 1 (after update_config)
    DeviceBuffer<uint64_t> r_head_, r_tail_, r_skip_, r_ing_head_, r_col_tail_;
    DeviceBuffer<uint32_t> r_reset_flag_, r_ing_skip_, r_ing_count_, r_ncols_, r_reset_out_;
    DeviceView<uint32_t> r_frames_;  // (views into r_staging_)
    DeviceView<uint8_t> r_mask_;
    RaggedStaging r_staging_;
    uint64_t last_cols_ = 0, last_stride_ = 0;
    uint32_t last_kind_ = OMX_COLUMN_REASSIGNED;
    EventTimer timer_;
    hipStream_t last_stream_ = nullptr;
};

struct SpectrogramSingle {
    SpectrogramBank bank;
    std::vector<uint64_t> offsets;
    std::vector<omx_spectrogram_point> points;
    std::vector<uint16_t> codes;
    explicit SpectrogramSingle(const omx_spectrogram_config& c) : bank(c, 1) { bank.host_outputs(size_t(2) << 20); }
    int process_block(const omx_block* block, omx_spectrogram_update* out);
};

}  // namespace omx
