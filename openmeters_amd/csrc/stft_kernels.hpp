// Launch interfaces of the STFT kernels (see stft_kernels.hip / spectrum_kernels.hip).  Plain structs
// of device pointers and scalars, passed by value as kernel arguments.  Every translation unit of
// the library is compiled by hipcc, so the clang vector type below is available on both sides.
#pragma once
#include "bluestein_plan.hpp"
#include "common.hpp"

namespace omx {

typedef float v2f __attribute__((ext_vector_type(2)));  // (re, im)

// ---------------------------------------------------------------- K0 ingest
constexpr int OMX_INGEST_MAX_OUT = 3;
constexpr int OMX_PROJECT_RAW = -1;  // push frame[0] unprojected (spectrogram mono path)

struct IngestArgs {
    const float* pcm;         // [n_streams][frames_total][channels]
    uint64_t frames_total;    // frames per stream in `pcm`
    uint64_t skip;            // leading frames not pushed (pending_skip)
    uint64_t count;           // frames pushed per stream = frames_total - skip
    AudioFormatArgs fmt;
    int n_out;
    int project[OMX_INGEST_MAX_OUT];
    float* ring[OMX_INGEST_MAX_OUT];  // [n_streams][cap] each
    uint64_t cap;             // power of two
    uint64_t head;            // absolute write position of the first pushed sample
    long long* last_nonzero;     // [n_streams] absolute position of the newest non-zero sample of ring 0, or nullptr
    long long* partial_nonzero;  // [n_streams][ingest_partials_per_stream(count)] scratch, or nullptr
    // ragged banks: per-stream {skip, count, head} (spectrogram_plan_kernel); `count` above is then the largest count (grid)
    const uint32_t* skips;
    const uint32_t* counts;
    const uint64_t* heads;
    // one launch feeding the rings of several banks (capture group: Spectrogram + Spectrum read the same block): when `per_out`,
    // output o lives in a ring of cap_o[o] samples per stream whose next write position is head_o[o] (lock-step banks only)
    int per_out;
    uint64_t cap_o[OMX_INGEST_MAX_OUT];
    uint64_t head_o[OMX_INGEST_MAX_OUT];
};
uint32_t ingest_partials_per_stream(uint64_t count);
// One projection launch for the rings of SEVERAL ragged banks reading the same block (capture group: Spectrogram + Spectrum with per-
// capture frame counts, registry.rs:396-418).  Every part is what a bank would have handed launch_ingest (per-stream skips / counts /
// heads from its own plan kernel, its rings and their capacity); part 0 may carry last_nonzero.  Returns false when the parts cannot share
// a launch (different blocks, more outputs than OMX_INGEST_MAX_OUT): the caller then launches them one by one.
bool launch_ingest_ragged_parts(const IngestArgs* parts, int n_parts, uint32_t n_streams, hipStream_t stream);

// What a lock-step bank wants pushed from one block (push_audio / push_sources of the reference): `skip` leading frames dropped,
// `count` frames projected into its ring(s).  A bank fills its slots in push_begin(); whoever launches the ingest kernel — the bank
// itself, or a capture group for several banks at once — reports back through push_end().
struct IngestSlots {
    int n = 0;
    int project[OMX_INGEST_MAX_OUT];
    float* ring[OMX_INGEST_MAX_OUT];
    uint64_t cap[OMX_INGEST_MAX_OUT], head[OMX_INGEST_MAX_OUT];
    uint64_t skip = 0, count = 0;
    long long* last_nonzero = nullptr;     // of slot 0 (spectrogram), or nullptr
    long long* partial_nonzero = nullptr;
};
// one ingest launch for the slots of one or more banks that agree on (skip, count); slot 0 carries last_nonzero
void launch_ingest_slots(const float* d_pcm, uint64_t frames, const AudioFormatArgs& fmt, const IngestSlots* const* banks, int n_banks,
                         uint32_t n_streams, hipStream_t stream);

// ---- ragged banks: the integer state machine of SpectrogramProcessor (push_audio / process_ready_windows / advance_audio,
// reference spectrogram/processor.rs:281-437, :490-516) per stream ON THE DEVICE, one thread per stream
struct SpectrogramPlanArgs {
    uint32_t n_streams;
    uint64_t read_len, hop, retained;  // retained = history_columns(kind, bins, history_length)
    uint32_t max_cols;                 // layout stride of the outputs (upper bound of any stream's columns in this call)
    const uint32_t* frames;            // [n_streams] new frames per stream in this call
    const uint8_t* reset_mask;         // [n_streams] reset_audio() before the push, or nullptr
    uint64_t* head;                    // [n_streams] state: absolute write position
    uint64_t* tail;                    //             absolute position of the oldest pending sample
    uint64_t* pending_skip;            //             samples still to skip (hop > window, :406-418)
    uint32_t* reset_flag;              //             `reset` of the next update (:511 std::mem::take)
    long long* last_nonzero;           //             audio_last_nonzero as an absolute position (-1 = None)
    uint32_t* ing_skip;                // out [n_streams]: leading frames of the block not pushed
    uint32_t* ing_count;               //                  frames pushed
    uint64_t* ing_head;                //                  absolute position of the first pushed sample
    uint64_t* col_tail;                //                  absolute position of column 0's first sample
    uint32_t* n_cols;                  //                  columns produced
    uint32_t* reset_out;               //                  the update's `reset` (meaningful where n_cols > 0)
};
void launch_spectrogram_plan(const SpectrogramPlanArgs& a, hipStream_t stream);
// ring re-homing on growth with per-stream positions: pending samples keep their absolute positions, only the modulus changes
void launch_ring_rehome(const float* from, uint64_t from_cap, float* to, uint64_t to_cap, const uint64_t* head, const uint64_t* tail,
                        uint32_t n_streams, hipStream_t stream);
void launch_spectrogram_ragged_config(uint32_t n_streams, const uint64_t* head, uint64_t* tail, uint64_t* pending_skip, uint32_t* reset_flag,
                                      long long* last_nonzero, bool trim, uint64_t keep, bool zero_skip, bool clear_nonzero, hipStream_t stream);
void launch_ingest(const IngestArgs& a, uint32_t n_streams, hipStream_t stream);

// ---------------------------------------------------------------- window sums in the reference's order (window_sum_kernels.hip)
// sums[s][r][h] = the sequential f32 fold (window.rs:76-79) over the `window` samples of hop first_hop + h of ring r of stream s.
// The fused classic / spectrum kernels divide it by the window length for the DC-removed copy's mean.
struct WindowSumArgs {
    const float* ring[2];   // [n_streams][cap] each
    uint32_t n_rings;       // 1 or 2
    uint64_t cap;           // power of two <= 2^31
    uint64_t tail;          // absolute position of hop 0's first sample (lock-step)
    const uint64_t* tails;  // ragged: per stream, or nullptr
    const uint32_t* hops;   // ragged: hops of every stream in this call (n_hops = layout stride), or nullptr
    uint32_t hop, window;
    uint32_t first_hop;     // first hop computed (hop h of the launch is hop first_hop + h of the call)
    uint32_t n_hops;        // hops per (stream, ring) computed by this launch = layout stride of `sums`
    uint32_t n_streams;
    float* sums;            // [n_streams][n_rings][n_hops]
    const uint32_t* modes;  // ragged banks: per stream kFoldWalk / kFoldCarry / kFoldNone (spectrum_plan_kernel); the walk serves kFoldWalk
};
enum : uint32_t { kFoldNone = 0, kFoldCarry = 1, kFoldWalk = 2 };
void launch_window_sums(const WindowSumArgs& a, hipStream_t stream);
// The same sums carried from call to call (lock-step banks fed a few samples per call): every window that has started keeps its running
// fold in one of `slots` = ceil(window / hop) slots per (stream, ring); see window_sums_carry_kernel.
struct WindowCarryArgs {
    const float* ring[2];
    uint32_t n_rings;
    uint64_t cap;
    uint64_t tail;        // first sample of window 0 of this call (the bank's tail before its hops were drained)
    uint64_t carry_pos;   // samples below this position are already in the carried folds (== tail: nothing is carried)
    uint64_t head;        // one past the newest sample in the ring
    uint64_t n_windows;   // windows that start below `head`: ceil((head - tail) / hop)
    uint32_t hop, window;
    uint32_t slots, slot0;  // slot of window k = (slot0 + k) % slots
    uint32_t first_hop, n_hops;  // sums[s][r][k - first_hop] for completed windows first_hop <= k < first_hop + n_hops
    uint32_t n_streams;
    float* carry;         // [n_streams][n_rings][slots]
    float* sums;          // [n_streams][n_rings][n_hops]
    // ragged banks: every stream's own positions (spectrum_plan_kernel) — tail / carry_pos / head / slot0 above are then unused, and
    // only the streams whose mode is kFoldCarry are served
    const uint64_t* tails;
    const uint64_t* froms;
    const uint64_t* heads;
    const uint32_t* slot0s;
    const uint32_t* modes;
};
void launch_window_sums_carry(const WindowCarryArgs& a, hipStream_t stream);

// ---------------------------------------------------------------- K2 fast reassigned STFT (W = F = 4096)
struct StftFastArgs {
    const float* ring;  // [n_streams][cap]
    uint64_t cap;
    uint64_t tail;      // absolute position of column 0's first sample
    uint32_t hop;
    uint32_t n_streams;
    uint32_t n_cols;
    uint32_t column_stride;  // points per column slot
    uint32_t window_size;    // classic columns: window length W <= transform length (0 = same as the transform)
    const long long* last_nonzero;
    const float* window;     // [4096]
    const float* dwindow;    // [4096] derivative window
    const float* twindow;    // [4096] time-weighted window
    const float* bin_norm;   // [2049]
    const v2f* tw256;        // exp(-2*pi*i*k/256),  k < 256
    const v2f* tw4096;       // exp(-2*pi*i*k/4096), k < 4096
    const v2f* tw8192;       // exp(-2*pi*i*k/8192) / 2, k < 4096 (2W-point twiddles carrying the real-FFT split's 1/2)
    float bin_hz, max_hz, inv_2pi, inv_hop, latency_hops;
    // two-term cosine-sum windows w[n] = c0 + c1 cos(2 pi n / W) (Hann, Hamming): the pair kernel applies the window on the
    // bins; win_terms = 2 for those, otherwise the kernels that window in the time domain run
    float win_c0, win_c1;
    uint32_t win_terms;
    // every window of the reference is a cosine sum w[n] = sum_m c_m cos(2 pi m n / W) of 1 ... 4 terms (window.rs:20-43: rectangular,
    // Hann, Hamming, Blackman, Blackman-Harris): the three-workgroups-per-CU kernel (stft4096_tri_kernels.hip) serves all five
    float cos_c[4];
    uint32_t cos_terms;  // 1 ... 4 (0: not a cosine sum, never the case for WindowKind)
    // ragged banks (per-stream frame counts): tail and column count of every stream, written by spectrogram_plan_kernel; the
    // scalar `tail` / `n_cols` above then hold nothing / the layout stride (columns per stream slot of points / counts / codes)
    const uint64_t* tails;
    const uint32_t* cols;
    omx_spectrogram_point* points;  // [n_streams][n_cols][column_stride]
    uint32_t* counts;               // [n_streams][n_cols]
    // classic columns: [n_streams][n_cols] sequential f32 window sums (window.rs:76-79; window_sum_kernels.hip), written by the classic
    // launchers ahead of the transform kernel
    float* col_sums;
};
__device__ __forceinline__ uint64_t stft_tail(const StftFastArgs& a, uint32_t s) { return a.tails ? a.tails[s] : a.tail; }
__device__ __forceinline__ uint32_t stft_cols(const StftFastArgs& a, uint32_t s) { return a.cols ? a.cols[s] : a.n_cols; }
constexpr int K2_PHASES = 12;
void k2_phase_cycles(unsigned long long out[K2_PHASES], bool reset);  // tuning builds only (OMX_K2_VARIANT=7)
unsigned long long* k2_phase_buffer();  // device address of the phase counters (OMX_K2_VARIANT=52: the pair kernel's marks)
// form: OMX_OPT_KERNEL_FORM (0 = tuned kernel, 1 = the five-transform kernel of round 1)
void launch_stft_reassigned_4096(const StftFastArgs& a, int form, hipStream_t stream);
int stft_reassigned_4096_transforms_per_frame();  // of form 0
void launch_stft_reassigned_4096_pair(const StftFastArgs& a, hipStream_t stream);  // stft4096_pair_kernels.hip
void launch_stft_reassigned_4096_tri(const StftFastArgs& a, hipStream_t stream);   // stft4096_tri_kernels.hip (form 2)
// tuning builds only (stft4096_swz_kernels.hip): the swizzled pair kernel and the one-column-per-workgroup kernel
void launch_stft_reassigned_4096_swz_pair(const StftFastArgs& a, hipStream_t stream);
void launch_stft_reassigned_4096_col(const StftFastArgs& a, hipStream_t stream);
uint32_t stream_column_grid(uint32_t n_streams, uint32_t n_cols);
// size-templated fused kernel (stft_pow2_kernels.hip): fft_size 1024 / 2048 (/ 4096 as a cross-check of the tuned kernel)
void launch_stft_reassigned_pow2(const StftFastArgs& a, uint32_t fft_size, hipStream_t stream);
// W = F = 8192, Hann / Hamming (stft8192_kernels.hip): `a.tw4096` = exp(-2 pi i k / 8192), `a.tw8192` = exp(-2 pi i k / 16384) / 2
void launch_stft_reassigned_8192(const StftFastArgs& a, hipStream_t stream);
// reassigned 16384: three kernels through an HBM scratch, frames [first, first + count) of the call per launch
constexpr int kBigHalo = 16;                       // neighbour bins kept on either side of 0 ... N/2 (zero padding <= 16)
template <int LOGN>
constexpr int kBigRow = (1 << LOGN) / 2 + 1 + 2 * kBigHalo;  // complex values per (frame, spectrum) row in bins mode
struct BigScratch {
    v2f* sv;          // [chunk][W] analytic slices
    v2f* spec;        // [3][chunk][N/2 + 1], or [2][chunk][kBigRow] in bins mode
    uint32_t first;   // first frame (item = stream * n_cols + column) of this chunk
    uint32_t count;   // frames in this chunk
};
// zero padding beyond 16384 points: zp W-point transforms of modulated slices per spectrum (stft_pow2_kernels.hip)
uint64_t stft_residue_scratch_bytes_per_frame(uint32_t window, uint32_t fft_size);
bool launch_stft_classic_residue(const StftFastArgs& a, uint16_t* codes, uint32_t window, uint32_t zp, const v2f* twF, hipStream_t stream);
bool launch_stft_reassigned_residue(const StftFastArgs& a, uint32_t window, uint32_t zp, const v2f* twF, void* scratch, uint32_t first, uint32_t count,
                                    hipStream_t stream);
void launch_hilbert_16k(const StftFastArgs& a, const BigScratch& sc, bool imag_only, hipStream_t stream);   // stft16384_kernels.hip
void launch_windowed_reassign_16k(const StftFastArgs& a, const BigScratch& sc, bool imag_only, hipStream_t stream);
void launch_windowed_16k(const StftFastArgs& a, const BigScratch& sc, hipStream_t stream);
uint64_t stft_big_scratch_bytes_per_frame();
void launch_stft_reassigned_16384(const StftFastArgs& a, void* scratch, uint32_t first, uint32_t count, hipStream_t stream);
bool launch_stft_reassigned_zp_16384(const StftFastArgs& a, uint32_t window, const v2f* twF, void* scratch, uint32_t first, uint32_t count,
                                     hipStream_t stream);
void launch_stft_reassigned_4096_split(const StftFastArgs& a, void* scratch, uint32_t first, uint32_t count, hipStream_t stream);
// zero-padded fused kernel: window 1024 / 2048, transform 2048 / 4096 (false: combination not covered)
bool launch_stft_reassigned_zp(const StftFastArgs& a, uint32_t window, uint32_t fft_size, const v2f* twF, hipStream_t stream);
// fused classic columns (u16 dB codes, two columns per complex FFT) for the same sizes
void launch_stft_classic_pow2(const StftFastArgs& a, uint16_t* codes, uint32_t fft_size, hipStream_t stream);

// ---------------------------------------------------------------- K1/K2 generic spectrogram
struct StftGenericArgs {
    const float* ring;
    uint64_t cap;
    uint64_t tail;
    uint32_t hop;
    uint32_t n_streams;
    uint32_t n_cols;
    uint32_t column_stride;
    const long long* last_nonzero;
    uint32_t reassign;      // 0 classic, 1 reassigned
    uint32_t window_size;   // W
    uint32_t fft_size;      // F = W * zero_padding_factor
    uint32_t hilbert_len;   // H (reassigned only)
    uint32_t log_fft, log_hilbert;
    const float* window;
    const float* dwindow;
    const float* twindow;
    const float* bin_norm;  // [F/2+1]
    const v2f* tw_fft;      // exp(-2*pi*i*k/F), k < F/2
    const v2f* tw_hilbert;  // exp(-2*pi*i*k/H), k < H/2
    float bin_hz, max_hz, inv_2pi, inv_hop, latency_hops;
    v2f* workspace;            // [n_workgroups][workspace_stride]
    uint64_t workspace_stride; // >= H + 3F (reassigned) or F (classic)
    omx_spectrogram_point* points;
    uint32_t* counts;
    uint16_t* codes;           // [n_streams][n_cols][column_stride] (classic)
    const uint64_t* tails;     // ragged banks: per-stream tail / column count (see StftFastArgs)
    const uint32_t* cols;
    BluesteinPlan blu;         // F not a power of two: chirp-z through a transform of blu.m points (scratch = the workspace's tail)
};
void launch_stft_generic(const StftGenericArgs& a, uint32_t n_workgroups, hipStream_t stream);

// ---------------------------------------------------------------- K3 spectrum
struct SpectrumPowerArgs {
    const float* ring[2];  // per active trace: [n_streams][cap]
    uint64_t cap;
    uint64_t tail;         // absolute position of hop 0's first sample
    uint32_t hop;
    uint32_t first_hop;    // first hop index computed by this launch
    uint32_t n_hops;       // hops computed by this launch
    uint32_t n_streams;
    uint32_t n_traces;     // active traces (1 or 2)
    uint32_t fft_size, log_fft, bins;
    const float* window;
    const float* bin_norm;
    const v2f* tw_fft;     // generic: exp(-2*pi*i*k/N), k < N/2
    const v2f* tw256;      // fast 4096
    const v2f* tw4096;
    v2f* workspace;        // generic: [wgs][fft_size]
    float* power;          // [n_streams][n_traces][n_hops][bins] (when !fused_db)
    const float* hop_sums; // fast kernels: [n_streams][n_traces][n_hops] sequential f32 window sums (window_sum_kernels.hip), written by
                           // the bank ahead of this launch (SpectrumBank::launch_window_sums_for)
    // AveragingMode::None: dB conversion fused into the power kernel, traces written directly
    uint32_t fused_db, emit_all, n_hops_out;
    uint32_t trace_slot[2];
    float state_floor, floor_db;
    const float* a_weighting_db;  // [bins]
    float* traces;         // [n_streams][n_hops_out][2 traces][2 weightings][bins]
    BluesteinPlan blu;     // fft_size not a power of two (generic kernel; scratch behind each workgroup's transform buffer)
    // ragged banks (per-stream frame counts): tail and hop count of every stream, written by spectrum_plan_kernel; `tail` above is then
    // unused and `n_hops` the layout stride (hops per stream slot of `power` / `traces`, the largest count of the call).  With
    // emit_all == 0 a stream's LAST hop is the one written to its slot 0.
    const uint64_t* tails;
    const uint32_t* hops;
};
// (s is workgroup-uniform at every call site; the per-stream values are LOADED, which makes them divergent for the compiler — every
// predicate and address derived from them became exec-mask regions and per-lane arithmetic.  Pinned to SGPRs here.)
__device__ __forceinline__ uint64_t spectrum_tail(const SpectrumPowerArgs& a, uint32_t s) {
    const uint64_t t = a.tails ? a.tails[s] : a.tail;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)t), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(t >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t spectrum_hops(const SpectrumPowerArgs& a, uint32_t s) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.hops ? a.hops[s] : a.n_hops));
}
void launch_spectrum_power(const SpectrumPowerArgs& a, bool fast4096, uint32_t generic_wgs, hipStream_t stream);

struct SpectrumLevelsArgs {
    const float* power;    // [n_streams][n_traces][n_hops][bins]
    float* smoothed;       // [n_streams][2][bins] averaging state, or nullptr for AveragingMode::None
    float* traces;         // [n_streams][n_hops_out][2 traces][2 weightings][bins]
    const float* a_weighting_db;  // [bins]
    uint32_t trace_slot[2];
    uint32_t n_streams, n_traces, n_hops, n_hops_out, bins;
    uint32_t mode, emit_all;
    float alpha, decay, state_floor, floor_db;
    const uint32_t* hops;  // ragged banks: hops of every stream in this call (n_hops = layout stride)
};
void launch_spectrum_levels(const SpectrumLevelsArgs& a, hipStream_t stream);
void launch_fill(float* p, uint64_t n, float v, hipStream_t stream);

// ---- ragged spectrum bank: SpectrumProcessor's integer state machine (push_sources / process_ready_windows, reference
// spectrum/processor.rs:179-213, :271-298) and reset_audio (:112-118) per stream on the device, one thread per stream
struct SpectrumPlanArgs {
    uint32_t n_streams;
    uint64_t fft_size, hop;
    uint32_t max_hops;                 // layout stride (upper bound of any stream's hops in this call)
    const uint32_t* frames;            // [n_streams] frames pushed by each stream in this call
    const uint8_t* reset_mask;         // [n_streams] or nullptr
    uint64_t *head, *tail, *pending_skip;  // [n_streams] per-stream positions (in / out)
    uint32_t *ing_skip, *ing_count;    // [n_streams] per-stream ingest arguments
    uint64_t* ing_head;
    uint64_t* hop_tail;                // [n_streams] absolute position of hop 0's first sample
    uint32_t* n_hops;                  // [n_streams]
    // the window folds of the call (window_sum_kernels.hip): which kernel serves stream s, and the carried folds' bookkeeping per stream
    uint32_t fold_slots;               // ceil(fft_size / hop), or 0 when folds are not carried (hop > fft_size, more than 64 slots, generic path)
    uint64_t* carry_pos;               // [n_streams] state: samples below this position are in the carried folds
    uint32_t* carry_slot0;             //             state: slot of the window that starts at the stream's tail
    uint32_t* carry_valid;             //             state
    uint32_t* fold_mode;               // out [n_streams]: kFoldNone / kFoldCarry / kFoldWalk
    uint64_t* fold_from;               //                  carry: first sample to add to the windows that have started
    uint32_t* fold_slot0;              //                  carry: slot of this call's window 0
};
void launch_spectrum_plan(const SpectrumPlanArgs& a, hipStream_t stream);
// reset_audio of the masked streams' level state: averaging state to 0, trace rows back to the floor
void launch_spectrum_reset_streams(const uint8_t* reset_mask, uint32_t n_streams, float* smoothed, uint64_t smoothed_per_stream, float* traces,
                                   uint64_t traces_per_stream, float floor_db, hipStream_t stream);

// compute_derivative_spectral on the device: n = power of two >= 2, tw = exp(-2*pi*i*k/n) (k < n/2),
// scratch = n complex values.
void launch_derivative_window(const float* window, uint32_t n, const void* tw, void* scratch, float* out,
                              hipStream_t stream);

}  // namespace omx
