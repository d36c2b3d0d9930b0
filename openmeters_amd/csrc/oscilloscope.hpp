// OscilloscopeBank: S independent OscilloscopeProcessors (reference
// src/visuals/oscilloscope/processor.rs:570-759).  One workgroup per stream; the NSDF period estimate
// (K6) and the template-correlation trigger (K7) run inside the same kernel, block after block.
#pragma once
#include "stft_kernels.hpp"

namespace omx {

constexpr int kScopeTraces = 3;        // traces[0], traces[1], separate trigger source
constexpr int kScopeTarget = 4096;     // write_snapshot TARGET (:726)

struct ScopeTriggerState {  // StableTrigger persistent fields (:272-282)
    int has_period;
    float period;
    uint32_t missed_periods;
    float reference_period;
    float mean;
    uint32_t ref_len;       // reference.len()
    uint32_t _pad[2];
};

struct ScopeBlockHeader {   // per (stream, block) snapshot header
    uint32_t produced;      // process_block returned Some
    uint32_t channels;
    uint32_t slots[2];
    uint32_t samples_per_channel;
    uint32_t locked;        // last_cycle_rate().is_some() after this block (:602-609)
    float period;           // the period behind last_cycle_rate
    uint32_t capture_start; // Capture::start / frac_offset of the first captured trace
    float capture_frac;
    uint32_t _pad;
};

struct ScopeEstimate {      // PeriodEstimator::estimate_period result for one (stream, block, view), computed ahead of the trigger pass
    int some;
    float period, confidence, last_peak;
};

struct ScopeArgs {
    const float* pcm;       // [n_streams][frames_total][channels]
    uint64_t frames_total;
    uint32_t block_frames, n_blocks, n_streams;
    AudioFormatArgs fmt;
    float sample_rate;
    uint32_t trigger_mode;  // OMX_TRIGGER_*
    uint32_t num_cycles;
    uint32_t trace_channel[2];   // OMX_CHANNEL_*
    uint32_t trigger_source;
    int matching_trace;     // slot whose channel == trigger_source (and active), else -1
    uint32_t separate_source;
    uint32_t base_frames, max_period, probe_frames, history_frames;
    // trace rings: [n_streams][kScopeTraces][cap]; lengths are uniform over streams (lock-step pushes)
    float* rings;
    uint64_t cap;           // power of two >= history_frames + block_frames
    uint64_t head[kScopeTraces];  // absolute position of the next pushed sample at the start of the call
    uint64_t len[kScopeTraces];   // deque length at the start of the call
    // per-stream trigger state + scratch
    ScopeTriggerState* trig;     // [n_streams][kScopeTraces]  (index 2 = the `source` trigger)
    float* reference;       // [n_streams][kScopeTraces][max_kernel]
    float* scratch;         // [n_streams][scratch_stride] work / candidate / periodicity / energy_prefix
    uint64_t scratch_stride;
    uint32_t max_kernel;    // capacity of reference / candidate
    uint32_t fft_size, log_fft;  // NSDF FFT (next_pow2(probe + max_lag))
    const v2f* tw_fft;      // exp(-2*pi*i*k/fft_size), k < fft_size/2
    v2f* fft_global;        // [n_streams][fft_size] when the FFT does not fit the LDS budget, else nullptr
    const v2f* tw256;       // exp(-2 pi i k / 256), exp(-2 pi i k / 4096): tables of the register/LDS 4096-point transform that
    const v2f* tw4096;      // carries the 8192-point autocorrelation (nullptr unless fft_size == 8192)
    ScopeBlockHeader* headers;   // [n_streams][n_blocks]
    float* samples;         // [n_streams][2][kScopeTarget] snapshot of the newest block
    uint32_t phase_timing;  // tuning aid: accumulate per-phase cycles (OMX_SCOPE_PHASES=1)
    uint32_t lds_scratch;   // hot scratch arrays in LDS (fast 8192 configuration whose two phase layouts fit 150 KiB)
    // two-pass form (many blocks per call): every block is pushed into the rings first (cap >= history + frames of the call),
    // the period estimates — a pure function of the trace — are computed for all (stream, block, view) in parallel, and the
    // per-stream kernel only runs the stateful part (stabilise / locate / snapshot) block after block
    ScopeEstimate* estimates;    // [n_streams][n_blocks][kScopeTraces] or nullptr (single-pass form)
    uint32_t est_view_count;     // wide form: the views whose estimate a block of this call can ask for (grid z of the estimate kernel)
    uint32_t est_views[kScopeTraces];
    uint32_t pre_pushed;         // one-workgroup-per-stream kernel: the call's frames are in the rings already and a.estimates holds every estimate
    // ragged banks (per-stream block counts; nullptr = lock-step; single-pass form only): stream s runs blocks_v[s] <= n_blocks
    // blocks from its own ring positions pos_v[s][trace] = {head, len} (head / len above are then unused), after a
    // clear_history() of its own when reset_v[s] != 0 (its epoch_v[s] then advances)
    uint64_t* pos_v;             // [n_streams][kScopeTraces][2]
    const uint32_t* blocks_v;    // [n_streams]
    const uint8_t* reset_v;      // [n_streams]
    uint64_t* epoch_v;           // [n_streams]
    // chunk calls (process_chunks): stream s's blocks are frames_v[s] frames long (nullptr = block_frames for every stream); its row
    // of `pcm` is frames_total frames long whatever it delivers
    const uint32_t* frames_v;    // [n_streams]
    // wide trigger pass with LESS LDS than its worst case (round 4: 88.2 ... 192 kHz, whose worst-case arrays exceed a CU's 160 KiB): it runs
    // the blocks of a stream while their arrays fit and hands the rest to the one-workgroup-per-stream kernel
    uint32_t ref_cap;            // floats of the resident-reference region (0: max_kernel, no hand-over)
    uint32_t* resume_blk;        // [n_streams] first block the capped pass did not run (the stream's block count: it ran them all), or nullptr
    uint64_t* resume_pos;        // [n_streams][kScopeTraces][2] ring {head, len} in front of that block
    uint32_t resume_mode;        // one-workgroup-per-stream kernel: continue at resume_blk[s] from resume_pos (state already past a reset)
};
uint64_t scope_lds_scratch_bytes(uint32_t max_kernel, uint32_t max_period, uint32_t probe_frames);
uint64_t scope_locate_lds_bytes(uint32_t max_kernel, uint32_t max_period);
constexpr int SCOPE_PHASES = 10;
void scope_phase_cycles(unsigned long long out[SCOPE_PHASES], bool reset);
void scope_fast_phase_cycles(unsigned long long out[SCOPE_PHASES], bool reset);  // the wide form's trigger kernel
void launch_oscilloscope(const ScopeArgs& a, hipStream_t stream);
// wide form (scope_fast_kernels.hip): push / estimate / trigger kernels; a.estimates != nullptr, rings hold history + the call
void launch_oscilloscope_fast(const ScopeArgs& a, hipStream_t stream);
// fft_size 16384 / 32768: push / big estimate kernels, then the single-pass kernel on precomputed estimates (a.pre_pushed = 1)
void launch_oscilloscope_big(const ScopeArgs& a, hipStream_t stream);
uint64_t scope_trigger_lds_bytes(uint32_t max_kernel, uint32_t max_period);
// tests: the trigger pass' find_best on caller-supplied device arrays (work[len + search], template[len]); scores[search + 1]
void launch_scope_find_best_debug(const float* d_work, const float* d_tmpl, uint32_t len, uint32_t search, float period, uint32_t* d_best_off,
                                  float* d_frac, float* d_best_score, float* d_scores, hipStream_t stream);
// the newest len samples of every trace into a ring of another capacity (pos_v: per-stream {head, len}, else a.head / a.len)
void launch_scope_rehome(const float* from, uint64_t from_cap, float* to, uint64_t to_cap, const uint64_t* pos_v, const ScopeArgs& a,
                         uint64_t max_len, hipStream_t stream);
uint64_t scope_scratch_floats(uint32_t max_kernel, uint32_t max_search, uint32_t probe_frames, uint32_t max_period);

void oscilloscope_config_default(omx_oscilloscope_config* c);

class OscilloscopeBank {
public:
    OscilloscopeBank(const omx_oscilloscope_config& cfg, uint32_t n_streams);
    const omx_oscilloscope_config& config() const { return cfg_; }
    void update_config(const omx_oscilloscope_config& cfg);
    void reset_audio();
    // returns OMX_PRODUCED when the newest block produced a snapshot for stream 0 .. (per stream flags in headers)
    int process(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels,
                float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream);
    // Ragged call (include/omx.h: omx_oscilloscope_bank_process_ragged): stream s runs n_blocks[s] <= max_blocks blocks (its rows of
    // `d_pcm` are block_frames * max_blocks frames apart); streams flagged in reset_mask get reset_audio() first.  The per-stream ring
    // positions then live on the device; the bank stays ragged until reset_audio() of the whole bank.
    int process_ragged(const float* d_pcm, uint64_t block_frames, uint64_t max_blocks, const uint32_t* n_blocks, const uint8_t* reset_mask,
                       uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                       omx_oscilloscope_ragged_update* out);
    // Chunk call (include/omx.h: omx_oscilloscope_bank_process_chunks; VisualManager::ingest_samples, registry.rs:396-418): stream s
    // delivers ONE block of frames[s] <= frames_capacity frames (0 = nothing arrived): one trigger evaluation, one header slot per stream.
    int process_chunks(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels,
                       float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_oscilloscope_ragged_update* out);
    int fetch_header(uint64_t stream_index, uint64_t block, ScopeBlockHeader* dst, hipStream_t stream);
    int fetch_samples(uint64_t stream_index, float* dst, uint64_t count, hipStream_t stream);
    uint64_t epoch() const { return epoch_; }
    uint32_t n_streams() const { return n_streams_; }
    uint64_t last_blocks() const { return last_blocks_; }
    hipStream_t last_stream() const { return last_stream_; }
    const ScopeBlockHeader* d_headers() const { return headers_.ptr; }
    // test hook: the block at which the capped wide trigger pass handed stream s over in the last call (its block count: never), or -1
    // when the call did not run that form
    long long debug_resume_block(uint32_t s) const;
    const float* d_samples() const { return samples_.ptr; }
    void host_outputs(bool on) { host_outputs_ = on; }  // single-stream handles: headers / samples in pinned host memory

private:
    void rebuild(const omx_oscilloscope_config& cfg);
    void clear_history();
    struct RaggedCall {
        const uint32_t* n_blocks;
        const uint8_t* reset_mask;
        omx_oscilloscope_ragged_update* out;
        const uint32_t* frames_v = nullptr;  // chunk calls: per-stream block length, rows of `pcm` row_frames apart
        uint64_t row_frames = 0;
    };
    int process_impl(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels, float sample_rate,
                     const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, const RaggedCall* ragged);

    omx_oscilloscope_config cfg_{};
    uint32_t n_streams_;
    uint64_t epoch_ = 0;
    bool has_history_channels_ = false;
    uint32_t history_channels_ = 0;
    uint64_t head_[kScopeTraces] = {0, 0, 0}, len_[kScopeTraces] = {0, 0, 0};
    uint64_t cap_ = 0, last_blocks_ = 0;
    bool pending_unlock_ = true;
    uint32_t max_kernel_ = 0, fft_size_ = 0;
    HostStage staging_;
    OutBuffer<float> samples_;
    bool host_outputs_ = false;
    DeviceBuffer<float> rings_, reference_, scratch_, tw_fft_, fft_global_, tw256_, tw4096_;
    DeviceBuffer<ScopeTriggerState> trig_;
    OutBuffer<ScopeBlockHeader> headers_;
    DeviceBuffer<ScopeEstimate> estimates_;
    DeviceBuffer<uint32_t> resume_blk_;   // capped wide trigger pass: where the per-stream kernel takes over
    DeviceBuffer<uint64_t> resume_pos_;
    bool last_capped_ = false;
    hipStream_t last_launch_stream_ = nullptr;
    hipStream_t last_stream_ = nullptr;
    // ragged mode: per-stream ring positions and epochs on the device
    bool ragged_ = false;
    DeviceBuffer<uint64_t> r_pos_, r_epoch_;
    DeviceView<uint32_t> r_blocks_, r_frames_;  // (views into r_staging_)
    std::vector<uint32_t> h_blocks_;
    DeviceView<uint8_t> r_mask_;
    RaggedStaging r_staging_;
};

}  // namespace omx
