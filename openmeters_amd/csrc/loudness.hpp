// LoudnessBank: S independent LoudnessProcessors (reference src/visuals/loudness/processor.rs:218-312),
// four lanes per (stream, channel): register pipelines with per-lane window / true-peak phase.
#pragma once
#include "common.hpp"

namespace omx {

constexpr int kLoudnessWindows = 4;
constexpr int kTruePeakMaxDelay = 24;

// Per-channel recurrent state (array of structs in HBM; loaded to registers for the whole call).
// Ring element: the K-weighted sample as the reference rounds it to f32 before squaring (loudness/processor.rs:161, :276-277), 0 for a
// non-finite one (WindowedMeans::push, dsp.rs:325-333).  What the windows sum is its square, exact in f64 — so the ring carries 4
// bytes per sample where the squares took 8, and every reader squares on the way in (bit-identical sums).
using RingT = float;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline double ring_square(RingT f) {
    const double d = (double)f;
    return d * d;
}

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
    RingT* ring;            // [slot group of 64][ring_len][64] K-weighted samples (f32; the windows sum their squares)
    LoudnessChannelState* state;  // [n_streams << slot_shift] (allocated for 8 slots per stream)
    float floor_db;
    omx_loudness_snapshot* snapshots;  // [n_streams][n_blocks]
    uint32_t n_meter_blocks;
    uint32_t role_perm;  // TEMP           // split launch: workgroups [0, n) = K-weighting + windows, [n, 2n) = true peak
    const uint32_t* run_if;  // fallback launch of the chunk-parallel path: run only when *run_if != 0
    // ragged banks (per-stream block counts; nullptr = lock-step): stream s runs blocks_v[s] <= n_blocks blocks from its own sample
    // counter seen_v[s] (frames_seen above is then unused), after a reset of its state when reset_v[s] != 0.  Lane-quad kernel and the
    // chunk-parallel kernels.
    uint64_t* seen_v;
    const uint32_t* blocks_v;
    const uint8_t* reset_v;
    // chunk calls (process_chunks): stream s's blocks are frames_v[s] frames long (nullptr = block_frames for every stream); its row of
    // `pcm` is frames_total frames long whatever it delivers
    const uint32_t* frames_v;
};
void launch_loudness(const LoudnessArgs& a, hipStream_t stream);

// ---- chunk-parallel evaluation (loudness_chunked.hip)
struct LoudChunkArgs {
    const float* pcm;       // [n_streams][frames_total][channels]
    uint64_t frames_total;
    uint32_t block_frames, n_blocks, n_streams, channels, slot_shift;  // slot = (stream << slot_shift) + channel
    double b[5], a[5];
    double weights[OMX_MAX_CHANNELS];
    uint8_t positions[OMX_MAX_CHANNELS];
    float fir4[12][3];
    float fir2[24];
    uint32_t delay_len;
    uint64_t capacities[kLoudnessWindows];
    uint64_t ring_len, frames_seen;
    RingT* ring;                   // the sequential kernels' ring: [group of 64 slots][ring slot][64]
    LoudnessChannelState* state;   // [slots]
    float floor_db;
    omx_loudness_snapshot* snapshots;
    const double* zs_weights;      // [block_frames][4]: W[k] = A^(L-1-k) B, the zero-state end state's weight on sample k of a block
    double* chunk_filter;          // [slots][n_blocks][4]: pass A's zero-state end states, then the true start states
    double* sub_sums;              // [slots][n_blocks * block_frames / 64]
    double* q_ring;                // [slots][q_len]: running total of the squared samples at the end of every 64-sample sub-block (high word)
    double* q_lo;                  // [slots][q_len]: its low word — the totals are double-double, so that a window sum (a difference of two
                                   // of them) is exact to ~1e-16 of ITSELF whatever the stream has played since its last reset
    uint64_t q_len;                // power of two
    double* tails;                 // [slots][windows][q_len]: sum of the last tail_len[w] samples of every sub-block; null when every
    uint32_t tail_len[kLoudnessWindows];  // tail_len[w] = capacities[w] % 64 is 0 (window starts on the sub-block grid)
    uint32_t* bad;
    uint32_t scan_dd;              // block scan of the K-weighting states in double-double arithmetic (rates above 96 kHz)
    // ragged calls (nullptr = lock-step): stream s runs blocks_v[s] <= n_blocks blocks from its own counter seen_v[s] (advanced by the
    // last kernel of the call), from a cleared state when reset_v[s] != 0
    uint64_t* seen_v;
    const uint32_t* blocks_v;
    const uint8_t* reset_v;
};
void launch_loudness_chunked(const LoudChunkArgs& a, const double* d_T /* [6][4][4] */, hipStream_t stream);
void launch_loudness_rebuild_q(const LoudChunkArgs& a, double* scratch, const uint32_t* only_if, hipStream_t stream);

void loudness_config_default(omx_loudness_config* c);
void k_weighting_coefficients(double fs, double b[5], double a[5]);
void k_weighting_transition_debug(double sample_rate, uint64_t frames, double out[192]);

class LoudnessBank {
public:
    LoudnessBank(const omx_loudness_config& cfg, uint32_t n_streams);
    void reset_audio();
    int process(const float* pcm, bool pcm_on_device, uint64_t block_frames, uint64_t n_blocks, uint32_t channels,
                float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                const omx_loudness_snapshot** d_snapshots);
    // Ragged call (include/omx.h: omx_loudness_bank_process_ragged): stream s runs n_blocks[s] <= max_blocks blocks of block_frames
    // frames (its rows of `d_pcm` are block_frames * max_blocks frames apart); streams flagged in reset_mask are reset first.  The
    // per-stream sample counters then live on the device; the bank stays ragged until reset_audio() of the whole bank.
    int process_ragged(const float* d_pcm, uint64_t block_frames, uint64_t max_blocks, const uint32_t* n_blocks, const uint8_t* reset_mask,
                       uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                       omx_loudness_ragged_update* out);
    // Chunk call (include/omx.h: omx_loudness_bank_process_chunks; VisualManager::ingest_samples, registry.rs:396-418): stream s
    // delivers ONE block of frames[s] <= frames_capacity frames (0 = nothing arrived); one snapshot slot per stream.
    int process_chunks(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels,
                       float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_loudness_ragged_update* out);
    int fetch(uint64_t stream_index, uint64_t block, omx_loudness_snapshot* dst, hipStream_t stream);
    EventTimer& timer() { return timer_; }
    hipStream_t last_stream() const { return last_stream_; }

private:
    void ensure_state(uint32_t channels, float sample_rate, hipStream_t stream);
    void clear_state(hipStream_t stream);
    void run_chunked(LoudnessArgs& la, hipStream_t stream);
    void fill_args(LoudnessArgs& la, const float* d_pcm, uint64_t block_frames, uint64_t n_blocks, uint32_t channels,
                   const uint8_t positions[OMX_MAX_CHANNELS]);
    // the two ragged entry points: `frames_v` null = n_blocks[s] blocks of block_frames, else one block of frames_v[s] where n_blocks[s] != 0
    int ragged_impl(const float* d_pcm, uint64_t row_frames, uint64_t block_frames, const uint32_t* frames_v, uint64_t max_blocks,
                    const uint32_t* n_blocks, const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                    const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_loudness_ragged_update* out);

    omx_loudness_config cfg_{};
    uint32_t n_streams_;
    uint32_t channels_ = 0;  // 0 = no channel state yet (reference: channels.len())
    double b_[5], a_[5];
    uint64_t frames_seen_ = 0, ring_len_ = 0, last_blocks_ = 0;
    bool state_clean_ = false;
    DeviceBuffer<RingT> ring_;
    DeviceBuffer<LoudnessChannelState> state_;
    OutBuffer<omx_loudness_snapshot> snapshots_;
    bool host_outputs_ = false;
    HostStage staging_;
    EventTimer timer_;
    hipStream_t last_stream_ = nullptr;
    // chunk-parallel path
    DeviceBuffer<double> chunk_filter_, sub_sums_, q_ring_, tails_, transition_, zs_weights_, rebuild_scratch_;
    DeviceBuffer<uint32_t> bad_;
    bool q_valid_ = false;
    uint64_t q_len_ = 0;     // entries per slot of q_ring_ / tails_ (power of two, grows with the call size)
    uint64_t q_age_ = 0;     // frames the running totals have accumulated since they were last taken from the ring
    uint64_t rebase_frames_ = 1ull << 22;  // OMX_OPT_LOUDNESS_REBASE_FRAMES
    float transition_rate_ = 0.0f;
    uint64_t transition_frames_ = 0;
    int last_form_ = 0;      // 1 = sequential kernels, 2 = chunk-parallel (omx_debug_loudness_bank_last_form)
    int chunked_mode_ = -1;  // -1 = choose by shape, 0 = never, 1 = whenever the shape allows
    // ragged mode: per-stream sample counters on the device
    bool ragged_ = false;
    DeviceBuffer<uint64_t> r_seen_;
    std::vector<uint64_t> h_seen_;  // the host's mirror of r_seen_ (the call arguments determine it)
    DeviceView<uint32_t> r_blocks_, r_frames_;  // (views into r_staging_)
    DeviceView<uint8_t> r_mask_;
    std::vector<uint32_t> h_blocks_;
    RaggedStaging r_staging_;
public:
    void chunked_mode(int mode) { chunked_mode_ = mode; }
    void rebase_frames(uint64_t frames) { rebase_frames_ = frames; }
    int last_form() const { return last_form_; }
    void host_outputs(bool on) { host_outputs_ = on; }  // single-stream handles: snapshots in pinned host memory
};

}  // namespace omx
