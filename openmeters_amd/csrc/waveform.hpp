// WaveformBank: S independent WaveformProcessors (reference src/visuals/waveform/processor.rs:135-353).
// 16 lanes per stream: lane = channel * 3 + band for channel in {Left, Right, Mid, Side}, band in {low, mid, high}
// (12 live lanes); band-0 lanes also carry the channel's min/max column state machine.
#pragma once
#include <map>

#include "stereometer.hpp"  // BiquadCoef, make_biquad

namespace omx {

struct WaveLaneState {
    float za[2][2], zb[2][2];      // biquad states [L/R][z0,z1]: stage A (HP_low, mid band only), stage B
    double color[4];               // CompensatedPair of the colour window: sum0, sum1, cor0, cor1
    double hist[2][4];             // history windows (fast, slow)
    // min/max column state of the channel (band-0 lanes): current = Option<(min, max, Option<last>)>, last_sample
    float cur_min, cur_max, cur_last, last_sample;
    uint32_t cur_some, cur_has_last, last_valid, _pad;
};

struct WaveformArgs {
    const float* pcm;  // [n_streams][frames][channels]
    uint64_t frames;
    uint32_t n_streams;
    AudioFormatArgs fmt;
    uint32_t analyze, track_history;
    BiquadCoef lp_lo, hp_lo, lp_hi, hp_hi;  // ThreeBand<Biquad,false> (dsp.rs:473-504), 200 / 2000 Hz
    double step;          // (scroll_speed / sample_rate).clamp(0, 1) (:253-254)
    double column_phase;  // at the start of the call
    uint64_t pushes;      // tracker pushes since the trackers were created
    uint32_t color_len, slow_len;
    float* color_ring;    // [color_len][n_streams * 16]
    float* hist_ring;     // [slow_len][n_streams * 16]
    WaveLaneState* state; // [n_streams * 16]
    uint64_t n_emit;      // columns emitted by this call (host-simulated phase)
    uint64_t first_kept;  // n_emit - min(n_emit, max_columns)
    omx_wave_column* columns;  // [n_streams][n_emit - first_kept][4]
    omx_wave_column* preview;  // [n_streams][4]
    uint32_t write_preview;
    // ragged banks (per-stream frame counts; nullptr = lock-step; the role-per-wavefront kernel carries them per lane, the one-wavefront
    // kernel runs ONE stream per workgroup so that they stay workgroup-uniform): stream s receives frames_v[s] <= frames frames (`frames` is then the row stride of pcm),
    // continues from its own tracker push count / column phase, after a reset of its own when reset_v[s] != 0; columns go to
    // columns[s][max_cols][4], their count to cols_v[s], the preview progress to progress_v[s]
    const uint32_t* frames_v;
    const uint8_t* reset_v;
    uint64_t* pushes_v;
    double* phase_v;
    uint32_t* cols_v;
    float* progress_v;
    uint64_t max_cols;
    const uint32_t* run_if;  // fallback launch of the chunk-parallel path: run only when *run_if != 0
};
void launch_waveform(const WaveformArgs& a, hipStream_t stream);

// ---- chunk-parallel form (waveform_chunked.hip; lock-step bank calls)
struct WaveEval {  // one kept column, or (out = 0xFFFFFFFF) the pseudo-column at the end of the call
    uint32_t idx_end;         // cut index of the column's last frame
    uint32_t idx_start[3];    // cut index of (last frame - window length): colour, fast history, slow history
    uint32_t idx_refresh[3];  // (pseudo-column) cut index of the windows' last CompensatedPair refresh
    uint32_t count[3];        // WindowedMeans::mean's divisor (dsp.rs:367-370)
    uint32_t mm_from, mm_to;  // the column's segments [from, to), counted from the first segment of the call
    uint32_t out;             // kept-column index
    uint32_t carry;           // the column began before this call: merge the carried column state
};
constexpr uint32_t kWaveNoSlot = 0xFFFFFFFFu;
struct WaveChunkArgs {
    const float* pcm;  // [n_streams][pcm_stride][2]
    uint64_t frames;   // frames of this group's streams in this call
    uint64_t pcm_stride;
    uint32_t n_streams;          // streams of the bank (ring row stride, state index)
    const uint32_t* stream_map;  // [n_local] bank index of the group's streams (nullptr: identity)
    uint32_t n_local;            // streams of the group: every scratch array is indexed by the LOCAL stream
    float m00, m10, m01, m11;  // the two-channel fold
    BiquadCoef lp_lo, hp_lo, lp_hi, hp_hi;
    // the three sections the chunk kernel runs in f64, widened on the host: as kernel arguments they stay in SGPR pairs (converted on
    // the device they occupied 20 VGPRs)
    struct Coef64 { double b[3], a[2]; } lp_lo64, hp_lo64, lp_hi64;
    uint32_t history;
    uint32_t chunk_frames, n_chunks;
    uint64_t pushes0;  // tracker pushes before the call
    uint32_t color_len, slow_len;
    float* color_ring;
    float* hist_ring;
    WaveLaneState* state;
    const int32_t* cuts;        // [n_segs + 1] ascending, call-relative frame indices (negative: before the call)
    uint32_t n_segs, n_old_segs;
    const uint32_t* chunk_seg;  // [n_chunks] first segment of a chunk
    const WaveEval* evals;
    uint32_t n_evals;           // the kernel's share of the list (columns first, the pseudo-column last)
    float* chunk_state;         // [n_chunks][n_streams][3][16 floats]: low / mid band 4 / 8 f64 states, high band 4 f32
    double* seg_sum;            // [n_segs][n_streams][24]: series (|v| gain, v^2) x channel x band
    float* seg_mm;              // [new segments][n_streams][4][min, max, last]
    double* prefix_hi;          // [n_segs + 1][n_streams][24]
    double* prefix_lo;
    uint32_t* bad;
    // running totals kept from call to call (lock-step calls, waveform.cpp WaveformBank::Totals); old_slot == nullptr: not in use
    const uint32_t* old_slot;   // [n_old_segs + 1] slot of the total at each cut up to and including -1, or kWaveNoSlot
    double* totals;             // [slots][2][n_streams x 24]: double-double totals of everything pushed up to a cut, since the trackers' clear
    const uint32_t* keep;       // [n_keep][2]: (cut index, slot) of the totals this call leaves for later calls
    uint32_t n_keep;
    uint32_t base_slot;         // slot of the total at the call's start (cut -1), or kWaveNoSlot: zero
    uint32_t all_kept;          // every entry of old_slot is a slot
    uint64_t* void_end;         // push count below which kept totals are void (a call the sequential kernel had to do)
    uint64_t first_count;       // pushes covered by cut 0
    uint64_t end_count;         // pushes covered by the call's last frame
    omx_wave_column* columns;   // [n_streams][col_stride][4]
    omx_wave_column* preview;
    uint64_t col_stride;
    uint32_t write_preview;
};
void launch_waveform_chunked_phase1(const WaveChunkArgs& a, const double* d_T, hipStream_t stream);
void launch_waveform_chunked_phase2(const WaveChunkArgs& a, hipStream_t stream);
// ragged calls: the host's per-stream counters ([n] u64 pushes, [n] f64 phases, [n] u32 columns, [n] f32 progress, packed) -> the device
// arrays the sequential kernels and the caller read, unless *bad (then the sequential kernel writes them itself)
// reset_audio of the listed streams: their 16 lane states cleared (filters, compensated pairs, the open column)
void launch_waveform_reset_streams(WaveLaneState* state, const uint32_t* streams, uint32_t n, hipStream_t stream);
void launch_waveform_mirror_copy(const uint8_t* src, uint32_t n, uint64_t* pushes_v, double* phase_v, uint32_t* cols_v, float* progress_v,
                                 const uint32_t* bad, hipStream_t stream);
// role-per-wavefront form (waveform_roles_kernels.hip); launch_waveform picks it whenever it applies (OMX_WAVEFORM_SINGLE=1 pins
// the one-wavefront kernel: A/B runs and the bit-identity test)
bool waveform_roles_applicable(const WaveformArgs& a);
void launch_waveform_roles(const WaveformArgs& a, hipStream_t stream);

void waveform_config_default(omx_waveform_config* c);

class WaveformBank {
public:
    void host_outputs(bool on) { host_outputs_ = on; }  // single-stream handles: columns in pinned host memory
    WaveformBank(const omx_waveform_config& cfg, uint32_t n_streams);
    const omx_waveform_config& config() const { return cfg_; }
    void update_config(const omx_waveform_config& cfg);
    void reset_audio() {
        rebuild();
        ragged_ = false;
    }
    void prepare(hipStream_t stream);
    int process(const float* pcm, bool pcm_on_device, uint64_t frames, uint32_t channels, float sample_rate,
                const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_waveform_bank_update* out);
    // Ragged call (include/omx.h: omx_waveform_bank_process_ragged): stream s receives frames[s] <= frames_capacity frames (its rows
    // of `d_pcm` are frames_capacity frames apart); streams flagged in reset_mask get reset_audio() first.  Push counts and column
    // phases then live on the device per stream; the bank stays ragged until reset_audio() of the whole bank.
    int process_ragged(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels,
                       float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_waveform_ragged_update* out);
    int fetch(uint64_t stream_index, omx_wave_column* columns, omx_wave_column* preview, hipStream_t stream);
    hipStream_t last_stream() const { return last_stream_; }
    uint64_t last_columns() const { return last_cols_; }
    struct ChunkGroup;  // a set of streams that move in lock step through a call of the chunk-parallel form (waveform.cpp)
    void set_form(uint32_t form) { form_ = form; }  // OMX_OPT_KERNEL_FORM: 0 by call shape, 1 sequential, 2 chunk-parallel where it applies
    uint32_t last_form() const { return last_form_; }

private:
    void rebuild();
    void reset_trackers();
    struct RaggedCall {
        const uint32_t* frames;
        const uint8_t* reset_mask;
        omx_waveform_ragged_update* out;
    };
    int process_impl(const float* pcm, bool pcm_on_device, uint64_t frames, uint32_t channels, float sample_rate,
                     const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream, omx_waveform_bank_update* out, const RaggedCall* ragged);

    omx_waveform_config cfg_{};
    uint32_t n_streams_;
    uint32_t source_channels_ = 2;
    bool analysis_ = false, reset_pending_ = true, clear_minmax_ = true, clear_trackers_ = true;
    double column_phase_ = 0.0;
    uint64_t pushes_ = 0, last_cols_ = 0;
    uint32_t color_len_ = 0, slow_len_ = 0;
    DeviceBuffer<float> color_ring_, hist_ring_;
    HostStage staging_;
    DeviceBuffer<WaveLaneState> state_;
    OutBuffer<omx_wave_column> columns_, preview_;
    bool host_outputs_ = false;
    hipStream_t last_stream_ = nullptr;
    // ragged mode: per-stream tracker push counts and column phases on the device
    bool ragged_ = false, ragged_zero_phase_ = false, ragged_zero_pushes_ = false;
    DeviceBuffer<uint64_t> r_pushes_;
    DeviceBuffer<double> r_phase_;
    DeviceView<uint32_t> r_frames_;  // (views into r_staging_)
    DeviceBuffer<uint32_t> r_cols_;
    DeviceBuffer<float> r_progress_;
    DeviceView<uint8_t> r_mask_;
    RaggedStaging r_staging_;
    // chunk-parallel form
    bool run_chunked(WaveformArgs& wa, const std::vector<uint32_t>& column_ends, double end_phase, hipStream_t stream);
    bool run_chunked_ragged(WaveformArgs& wa, const uint32_t* frames, const uint8_t* reset_mask, uint64_t max_cols, double step, hipStream_t stream);
    bool launch_chunk_groups(const WaveformArgs& wa, std::vector<ChunkGroup>& groups, uint64_t pcm_stride, uint64_t col_stride, hipStream_t stream);
    // ragged mode: the host's mirror of the per-stream push counts and column phases (valid while few distinct values exist)
    std::vector<uint64_t> h_pushes_;
    std::vector<double> h_phase_;
    bool mirror_valid_ = false;
    BlobStaging mirror_staging_, reset_staging_;
    DeviceBuffer<uint8_t> mirror_dev_;
    DeviceBuffer<uint32_t> reset_list_;
    uint32_t form_ = 0, last_form_ = 0;
    DeviceBuffer<double> transition_;
    float transition_rate_ = 0.0f;
    uint32_t transition_frames_ = 0;
    BlobStaging plan_staging_;
    DeviceBuffer<uint8_t> plan_;
    DeviceBuffer<float> chunk_state_, seg_mm_;
    DeviceBuffer<double> seg_sum_, prefix_;
    DeviceBuffer<uint32_t> bad_;
    // Running totals kept between lock-step calls of the chunk-parallel form (waveform_chunked.hip: wave_keep_totals_kernel writes them, wave_columns_kernel reads them): the
    // double-double total of every tracker input pushed so far, at the push counts later calls will start a window at.  Any call the
    // chunk-parallel form does not do (sequential by shape or by choice, ragged) empties the table; a call the device hands to the
    // sequential kernel behind the host's back (`bad`) voids it on the device (void_end_).
    std::map<uint64_t, uint32_t> kept_;  // pushes covered -> slot
    std::vector<uint32_t> kept_free_;
    uint32_t kept_slots_ = 0;
    bool kept_zero_void_ = true;
    DeviceBuffer<double> kept_totals_;
    DeviceBuffer<uint64_t> void_end_;
    void kept_invalidate();
    bool kept_reserve(size_t slots, hipStream_t stream);
};

}  // namespace omx
