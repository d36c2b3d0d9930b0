/*
 * omx.h — C-ABI boundary of the MI355X-native OpenMeters DSP hot path.
 *
 * Every entry point below replaces one inherent method of a reference
 * `<Visual>Processor` (the seam `VisualModule::ingest` calls through,
 * reference src/visuals/registry.rs:107-115, :247-256).  The reference is
 * Rust; a maintainer binds these with an `extern "C"` block (see
 * INTEGRATION.md).  Signatures are plain pointers and sizes: no C++ types,
 * no torch types, nothing HIP-specific leaks out.
 *
 * Conventions
 *   - return  1  : a snapshot was produced   (reference: `Some(snapshot)`)
 *   - return  0  : nothing to show           (reference: `None`)
 *   - return <0  : backend failure (omx_status) — a case the reference does not have
 *   - all snapshot pointers are callee-owned and stay valid until the next call
 *     on the same handle (superset of the reference's borrowed `&SpectrumSnapshot`,
 *     reference src/visuals/spectrum/processor.rs:255)
 *   - handles are not thread-safe; calls on one handle are serialised by the
 *     caller (reference: `Rc<RefCell<..>>`, src/visuals/registry.rs:23)
 *   - PCM is interleaved f32, frame-major `[frame][channel]`
 *     (reference src/dsp.rs:219-229)
 *
 * Two families per visual:
 *   omx_<visual>_*        one stream, host pointers in/out: the drop-in for
 *                         `<Visual>Processor::{new,config,update_config,
 *                         reset_audio,prepare,process_block}`
 *   omx_<visual>_bank_*   S independent streams per call, device-resident
 *                         in/out: the batched MI355X path (one HIP launch
 *                         covers every ready (stream, hop)).
 */
#ifndef OMX_H
#define OMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMX_MAX_CHANNELS 8 /* reference src/dsp.rs:6 MAX_AUDIO_CHANNELS */

typedef enum omx_status {
    OMX_PRODUCED = 1,
    OMX_NONE = 0,
    OMX_ERR_BACKEND = -1,     /* HIP runtime error (see omx_last_error) */
    OMX_ERR_UNSUPPORTED = -2, /* config shape outside what the HIP path implements */
    OMX_ERR_INVALID = -3,     /* null handle / null pointer */
    OMX_ERR_NO_DEVICE = -4    /* no gfx950 device visible: there is no CPU fallback */
} omx_status;

/* reference src/dsp.rs:8-22 ChannelPosition; Aux(n) = OMX_POS_AUX0 + n */
enum {
    OMX_POS_FRONT_LEFT = 0,
    OMX_POS_FRONT_RIGHT = 1,
    OMX_POS_FRONT_CENTER = 2,
    OMX_POS_LOW_FREQUENCY = 3,
    OMX_POS_REAR_LEFT = 4,
    OMX_POS_REAR_RIGHT = 5,
    OMX_POS_SIDE_LEFT = 6,
    OMX_POS_SIDE_RIGHT = 7,
    OMX_POS_MONO = 8,
    OMX_POS_UNKNOWN = 9,
    OMX_POS_AUX0 = 16
};

/* reference src/util/audio/window.rs:9-18 WindowKind */
enum {
    OMX_WINDOW_RECTANGULAR = 0,
    OMX_WINDOW_HANN = 1,
    OMX_WINDOW_HAMMING = 2,
    OMX_WINDOW_BLACKMAN = 3,
    OMX_WINDOW_BLACKMAN_HARRIS = 4
};

/* reference src/util/audio/channel.rs:4-10 Channel */
enum {
    OMX_CHANNEL_LEFT = 0,
    OMX_CHANNEL_RIGHT = 1,
    OMX_CHANNEL_MID = 2,
    OMX_CHANNEL_SIDE = 3,
    OMX_CHANNEL_NONE = 4
};

/* reference src/dsp.rs:108-115 AudioBlock (borrowed for the call only).
 * `positions` are used as given (the reference normalises them upstream in
 * AudioFormat::new, src/dsp.rs:88-101; omx_positions_normalize does the same). */
typedef struct omx_block {
    const float* samples; /* interleaved, n_samples = frames * channels */
    uint64_t n_samples;
    uint32_t channels;    /* clamped to 1..=8 like AudioBlock::with_positions */
    float sample_rate;    /* sanitised like src/util/audio/rate.rs:9-13 */
    uint8_t positions[OMX_MAX_CHANNELS];
} omx_block;

/* reference src/dsp.rs:36-47 ChannelPosition::fallback */
void omx_positions_fallback(uint32_t channels, uint8_t out[OMX_MAX_CHANNELS]);
/* reference src/dsp.rs:49-76 ChannelPosition::normalize */
void omx_positions_normalize(uint32_t channels, const uint8_t in[OMX_MAX_CHANNELS],
                             uint8_t out[OMX_MAX_CHANNELS]);

/* Text of the most recent backend error on this thread ("" if none). */
const char* omx_last_error(void);
/* 1 if a gfx950 device is usable, else 0.  Never falls back to the CPU. */
int omx_device_available(void);
/* Number of HIP devices visible to the process (0 when there is none), and the device of the handles created from now on: a host
 * without the HIP headers — the Rust service, one process per GPU — calls omx_set_device(LOCAL_RANK) once before it creates
 * anything.  The choice is PROCESS-wide: HIP's current device is per host thread, so every entry point of this library binds the
 * calling thread to the selected device first (a capture thread and a UI thread of one host land on the same GPU).  Handles and
 * banks live on that device; the stream argument of every call must belong to it.  One device per process: changing the device
 * while handles exist is not supported.  Returns 0, OMX_ERR_INVALID (index out of range) or OMX_ERR_NO_DEVICE (not a gfx950). */
int omx_device_count(void);
int omx_set_device(int index);
/* Library version string. */
const char* omx_version(void);
/* ABI revision of this header: bumped whenever a struct a host passes or receives changes layout, or an argument changes meaning.  A host
 * compares omx_abi_version() of the library it loaded with the OMX_ABI_VERSION it was compiled against and refuses a mismatch
 * ("ABI revisions" lists what changed in each).
 *   5  round 5: cfg.block_frames == 0 means ONE block per ingest call (registry.rs:396-418), not the batcher quantum;
 *      omx_capture_group_ragged_update grew d_reset / d_block_frames per bank and d_stats_rows, its block_frames reads 0 in that mode
 *   6  round 6: omx_debug_window_sums added; no layout change */
#define OMX_ABI_VERSION 6
int omx_abi_version(void);

/* ===================================================================== *
 * Spectrogram — reference src/visuals/spectrogram/processor.rs
 * ===================================================================== */

/* reference :45-56 SpectrogramConfig (field for field) */
typedef struct omx_spectrogram_config {
    float sample_rate;
    uint32_t window;           /* OMX_WINDOW_* */
    uint64_t fft_size;
    uint64_t hop_size;
    uint64_t history_length;
    uint64_t zero_padding_factor;
    uint32_t use_reassignment; /* bool */
    uint32_t _pad;
} omx_spectrogram_config;

/* reference :37-43 SpectrogramPoint, #[repr(C)] 12-byte Pod */
typedef struct omx_spectrogram_point {
    float time_offset;
    float freq_hz;
    float power;
} omx_spectrogram_point;

enum { OMX_COLUMN_REASSIGNED = 0, OMX_COLUMN_CLASSIC = 1 };

/* reference :160-168 SpectrogramUpdate; `new_columns: Vec<SpectrogramColumn>`
 * is flattened CSR-style: column c owns
 *   points[column_offsets[c] .. column_offsets[c+1])   (reassigned) or
 *   codes [column_offsets[c] .. column_offsets[c+1])   (classic, u16 packed dB) */
typedef struct omx_spectrogram_update {
    uint64_t fft_size;
    uint64_t hop_size;
    uint64_t history_length;
    uint64_t n_columns;
    const uint64_t* column_offsets; /* n_columns + 1 entries */
    const omx_spectrogram_point* points;
    const uint16_t* codes;
    float sample_rate;
    float reassigned_power_scale;
    uint32_t reset; /* bool */
    uint32_t kind;  /* OMX_COLUMN_* */
} omx_spectrogram_update;

typedef struct omx_spectrogram omx_spectrogram;

void omx_spectrogram_config_default(omx_spectrogram_config* out);                 /* :45-59 defaults */
int omx_spectrogram_create(const omx_spectrogram_config* cfg, omx_spectrogram** out); /* ::new :188-206 */
void omx_spectrogram_destroy(omx_spectrogram* h);
int omx_spectrogram_get_config(const omx_spectrogram* h, omx_spectrogram_config* out);    /* ::config :208-210 */
int omx_spectrogram_update_config(omx_spectrogram* h, const omx_spectrogram_config* cfg); /* :518-543 */
int omx_spectrogram_reset_audio(omx_spectrogram* h);                              /* :212-217 */
int omx_spectrogram_prepare(omx_spectrogram* h);                                  /* :219-223 */
int omx_spectrogram_process_block(omx_spectrogram* h, const omx_block* block,
                                  omx_spectrogram_update* out);                   /* :490-516 */

/* reference :103-108 pack_classic_db, :144-158 col_byte_stride / history_columns */
uint16_t omx_pack_classic_db(float db);
uint64_t omx_spectrogram_history_columns(uint32_t kind, uint32_t points, uint64_t requested);

/* ---- batched bank: S independent SpectrogramProcessors, one launch ---- */
typedef struct omx_spectrogram_bank omx_spectrogram_bank;

/* Result of one bank call.  All streams receive the same number of frames per
 * call, so every stream emits the same number of columns.  Device buffers:
 *   d_counts : uint32 [n_streams][n_columns]            points kept per column
 *   d_points : omx_spectrogram_point [n_streams][n_columns][column_stride]
 *              (reassigned; first d_counts[s][c] entries valid, ascending bin order)
 *   d_codes  : uint16 [n_streams][n_columns][column_stride] (classic)            */
typedef struct omx_spectrogram_bank_update {
    uint64_t fft_size;
    uint64_t hop_size;
    uint64_t history_length;
    uint64_t n_streams;
    uint64_t n_columns;
    uint64_t column_stride; /* elements between consecutive columns */
    const uint32_t* d_counts;
    const omx_spectrogram_point* d_points;
    const uint16_t* d_codes;
    float sample_rate;
    float reassigned_power_scale;
    uint32_t reset;
    uint32_t kind;
} omx_spectrogram_bank_update;

int omx_spectrogram_bank_create(const omx_spectrogram_config* cfg, uint32_t n_streams,
                                omx_spectrogram_bank** out);
void omx_spectrogram_bank_destroy(omx_spectrogram_bank* b);
int omx_spectrogram_bank_update_config(omx_spectrogram_bank* b, const omx_spectrogram_config* cfg);
int omx_spectrogram_bank_reset_audio(omx_spectrogram_bank* b);
/* pcm: [n_streams][frames][channels] f32.  pcm_on_device != 0 ⇒ `pcm` is a device
 * pointer (the timed path); otherwise it is copied host→device first.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 * The call enqueues work and returns; outputs are ordered on `stream`. */
int omx_spectrogram_bank_process(omx_spectrogram_bank* b, const float* pcm, int pcm_on_device,
                                 uint64_t frames, uint32_t channels, float sample_rate,
                                 const uint8_t positions[OMX_MAX_CHANNELS], void* stream,
                                 omx_spectrogram_bank_update* out);
/* Ragged call — streams advance independently (one capture service per stream: `visuals/registry.rs:396-418` resets and
 * feeds every VisualManager on its own).  Stream s receives frames[s] (<= frames_capacity, 0 allowed) new frames — row s of
 * `pcm` (DEVICE memory, f32 [n_streams][frames_capacity][channels]) holds them at its start — after reset_audio() when
 * reset_mask[s] != 0 (reset_mask may be NULL).  `frames` and `reset_mask` are HOST arrays of n_streams entries.  The frame-index
 * rule, skip / retention and the `reset` flag are evaluated per stream on the device; outputs are laid out with `max_columns`
 * column slots per stream, of which stream s filled the first d_n_columns[s].  The first ragged call moves the bank to
 * per-stream positions; omx_spectrogram_bank_process is refused from then on until omx_spectrogram_bank_reset_audio. */
typedef struct omx_spectrogram_ragged_update {
    uint64_t fft_size;
    uint64_t hop_size;
    uint64_t history_length;
    uint64_t n_streams;
    uint64_t max_columns;            /* column slots per stream in d_counts / d_points / d_codes */
    uint64_t column_stride;
    const uint32_t* d_n_columns;     /* device [n_streams]: columns produced by each stream in this call */
    const uint32_t* d_reset;         /* device [n_streams]: SpectrogramUpdate::reset of each stream's update */
    const uint32_t* d_counts;        /* device [n_streams][max_columns] (0 beyond a stream's own columns) */
    const omx_spectrogram_point* d_points; /* device [n_streams][max_columns][column_stride] (reassigned) */
    const uint16_t* d_codes;         /* device [n_streams][max_columns][column_stride] (classic) */
    float sample_rate;
    float reassigned_power_scale;
    uint32_t kind;
    uint32_t _pad;
} omx_spectrogram_ragged_update;
int omx_spectrogram_bank_process_ragged(omx_spectrogram_bank* b, const float* pcm, uint64_t frames_capacity,
                                        const uint32_t* frames, const uint8_t* reset_mask, uint32_t channels,
                                        float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS], void* stream,
                                        omx_spectrogram_ragged_update* out);
/* Copy one column of one stream to host memory (synchronises `stream`). */
int omx_spectrogram_bank_fetch_column(omx_spectrogram_bank* b, uint64_t stream_index,
                                      uint64_t column, void* dst, uint64_t dst_capacity_elems,
                                      uint64_t* n_out);
/* Average duration (ms) of the dominant STFT kernel over the launches since the
 * last call, measured with HIP events on the launch stream; resets the tally. */
int omx_spectrogram_bank_kernel_time(omx_spectrogram_bank* b, double* avg_ms, uint64_t* launches);
/* Diagnostics knobs shared by the banks. */
enum {
    OMX_OPT_KERNEL_TIMING = 1, /* value != 0: bracket the dominant kernel with HIP events */
    OMX_OPT_FORCE_GENERIC = 2, /* value != 0: route through the generic any-size kernels (A/B checks) */
    OMX_OPT_KERNEL_FORM = 3,   /* spectrogram bank, reassigned 4096 / hop any: which of the equivalent kernel forms runs.
                                * 0 = tuned kernel (default; round 4: three workgroups per CU); 30 = size-templated kernel; 31 = three-kernel
                                * form through an HBM scratch (both are live code for other sizes).  1 = the round-1 kernel (five
                                * transforms per frame), 2 = the round-2 pair kernel: superseded, compiled into the TUNING library only
                                * (`make TUNING=1`) — the product returns OMX_ERR_UNSUPPORTED for them.  All compute the same columns
                                * (tests cross-check them); unknown values are rejected with OMX_ERR_INVALID. */
    OMX_OPT_LOUDNESS_REBASE_FRAMES = 4 /* loudness bank: a call that finds the chunk-parallel form's running totals older than this many frames
                                * takes them afresh from the sample ring first (default 2^22; checked once per call).  The totals are
                                * double-double pairs, so a window sum — the difference of two of them — is good to ~1e-16 of itself whatever
                                * the stream has played (90 s at full scale, then -100 dBFS: inside 1e-4 dB, tests/test_gpu_parity_meters.py);
                                * the rebuild is a second line of defence and a test hook */
};
/* tuning aid: cycles per phase of the fused 4096 kernel, accumulated by the phase-timing builds of the TUNING library
 * (`make TUNING=1`; setup/load, FFT, Hilbert build, inverse FFT, gather+window, dual FFT, third FFT, reassign+store).
 * The product build returns OMX_ERR_UNSUPPORTED. */
int omx_debug_k2_phase_cycles(uint64_t* out, uint32_t n, int reset);
/* Number of 4096-point complex transforms the default reassigned 4096 kernel executes per frame (bench.py prices the
 * executed-flop fraction of the FP32 vector peak with it).  No device needed. */
int omx_debug_transforms_per_frame(void);
/* test hook, no device needed: the zero-input transition of the K-weighting TDF-II over `frames` samples and its powers 2, 4 ... 32 as
 * the chunk-parallel loudness kernels take it — out[192] = [6][4][4] high parts, then [6][4][4] low parts (double-double pairs) */
int omx_debug_k_weighting_transition(double sample_rate, uint64_t frames, double* out);
/* same for the oscilloscope kernel with OMX_SCOPE_PHASES=1 (ring push, pre-FFT, FFTs, NSDF + peak, locate, snapshot) */
int omx_debug_scope_phase_cycles(uint64_t* out, uint32_t n, int reset);
/* test hook: StableTrigger::find_best (oscilloscope/processor.rs:441-484) of the wide-form trigger pass on caller-supplied host arrays
 * work[len + search], template[len] (period only sets the coarse stride); scores[search + 1] receives the correlation score of EVERY
 * offset as the sweeps compute it.  The arrays must fit the LDS of one CU. */
int omx_debug_scope_find_best(const float* work, const float* tmpl, uint32_t len, uint32_t search, float period, uint32_t* best_off,
                              float* frac_offset, float* best_score, float* scores);
/* test hook: the pre-pass that takes the DC-removed window's mean in the reference's order (window.rs:76-79, a sequential f32 fold).
 * `samples[n_streams][cap]` are rings of `cap` samples (a power of two); hop h of stream s covers the `window` samples from absolute
 * position tail + h * hop (mod cap).  sums[n_streams][n_hops] receives every hop's sum, bit for bit the reference's. */
int omx_debug_window_sums(const float* samples, uint32_t n_streams, uint64_t cap, uint64_t tail, uint32_t hop, uint32_t window, uint32_t n_hops,
                          float* sums);
int omx_spectrogram_bank_set_option(omx_spectrogram_bank* b, uint32_t option, uint64_t value);

/* ===================================================================== *
 * Spectrum — reference src/visuals/spectrum/processor.rs
 * ===================================================================== */

enum { OMX_AVERAGING_NONE = 0, OMX_AVERAGING_EXPONENTIAL = 1, OMX_AVERAGING_PEAK_HOLD = 2 };

/* reference :39-51 SpectrumConfig; AveragingMode (:64-70) is (mode, param):
 * Exponential{factor} / PeakHold{decay_per_second} */
typedef struct omx_spectrum_config {
    float sample_rate;
    uint32_t window;
    uint64_t fft_size;
    uint64_t hop_size;
    uint32_t averaging_mode;
    float averaging_param;
    uint32_t source;           /* OMX_CHANNEL_* */
    uint32_t secondary_source; /* OMX_CHANNEL_* */
    float floor_db;
    uint32_t _pad;
} omx_spectrum_config;

/* reference :33-37 SpectrumSnapshot: traces[trace] = [weighted_db, raw_db] */
typedef struct omx_spectrum_snapshot {
    uint64_t bins;
    const float* frequency_bins;
    const float* traces[2][2]; /* [trace][0 = A-weighted dB, 1 = raw dB], `bins` floats each */
} omx_spectrum_snapshot;

typedef struct omx_spectrum omx_spectrum;

void omx_spectrum_config_default(omx_spectrum_config* out);
int omx_spectrum_create(const omx_spectrum_config* cfg, omx_spectrum** out);      /* ::new :89-106 */
void omx_spectrum_destroy(omx_spectrum* h);
int omx_spectrum_get_config(const omx_spectrum* h, omx_spectrum_config* out);
int omx_spectrum_update_config(omx_spectrum* h, const omx_spectrum_config* cfg);  /* :300-322 */
int omx_spectrum_reset_audio(omx_spectrum* h);                                    /* :112-118 */
int omx_spectrum_prepare(omx_spectrum* h);                                        /* :120-124 */
int omx_spectrum_process_block(omx_spectrum* h, const omx_block* block,
                               omx_spectrum_snapshot* out);                       /* :255-269 */
float omx_a_weight(float freq_hz);                                                /* :410-425 */

typedef struct omx_spectrum_bank omx_spectrum_bank;
/* d_traces: f32 [n_streams][n_hops_out][2 traces][2 weightings][bins]; with
 * emit_all_hops == 0 only the newest hop is materialised (reference semantics:
 * the snapshot reflects the last hop, :268) and n_hops_out == 1. */
typedef struct omx_spectrum_bank_update {
    uint64_t bins;
    uint64_t n_streams;
    uint64_t n_hops;     /* hops processed in this call */
    uint64_t n_hops_out; /* hops materialised in d_traces */
    const float* d_traces;
    const float* d_frequency_bins;
} omx_spectrum_bank_update;
int omx_spectrum_bank_create(const omx_spectrum_config* cfg, uint32_t n_streams, int emit_all_hops,
                             omx_spectrum_bank** out);
void omx_spectrum_bank_destroy(omx_spectrum_bank* b);
int omx_spectrum_bank_update_config(omx_spectrum_bank* b, const omx_spectrum_config* cfg); /* update_config of every stream's processor (spectrum/processor.rs:300-320) */
int omx_spectrum_bank_reset_audio(omx_spectrum_bank* b);
int omx_spectrum_bank_process(omx_spectrum_bank* b, const float* pcm, int pcm_on_device,
                              uint64_t frames, uint32_t channels, float sample_rate,
                              const uint8_t positions[OMX_MAX_CHANNELS], void* stream,
                              omx_spectrum_bank_update* out);
/* Per-stream frame counts and per-stream reset_audio, as for the spectrogram bank (the reference runs one SpectrumProcessor per
 * capture, reset and fed on its own, visuals/registry.rs:396-418): `pcm` is device memory [n_streams][frames_capacity][channels],
 * stream s pushes its first frames[s] <= frames_capacity frames (0 = sits the call out, nothing of it changes); reset_mask (or NULL)
 * [n_streams]: non-zero = reset_audio() of that stream before its push (pending audio dropped, averaging state cleared, traces back
 * to the floor).  The hop rule of process_ready_windows (:179-213) is evaluated per stream on the device; `max_hops` hop slots per
 * stream are laid out, of which stream s filled the first d_n_hops[s] (emit_all_hops), or its slot 0 holds its newest hop
 * (emit_all_hops == 0; untouched when the stream produced none).  The first ragged call moves the bank to per-stream positions;
 * omx_spectrum_bank_process is refused from then on until omx_spectrum_bank_reset_audio. */
typedef struct omx_spectrum_ragged_update {
    uint64_t bins;
    uint64_t n_streams;
    uint64_t max_hops;            /* upper bound of any stream's hops in this call */
    uint64_t n_hops_out;          /* hop slots per stream in d_traces (max_hops with emit_all_hops, else 1) */
    const uint32_t* d_n_hops;     /* device [n_streams]: hops processed by each stream in this call */
    const float* d_traces;        /* device [n_streams][n_hops_out][2][2][bins] */
    const float* d_frequency_bins;
} omx_spectrum_ragged_update;
int omx_spectrum_bank_process_ragged(omx_spectrum_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                     const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                     const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_spectrum_ragged_update* out);
int omx_spectrum_bank_fetch(omx_spectrum_bank* b, uint64_t stream_index, uint64_t hop,
                            float* dst /* [2][2][bins] */);
int omx_spectrum_bank_set_option(omx_spectrum_bank* b, uint32_t option, uint64_t value);

/* ===================================================================== *
 * Loudness — reference src/visuals/loudness/processor.rs
 * ===================================================================== */

/* reference :210-216 LoudnessConfig */
typedef struct omx_loudness_config {
    float sample_rate;
    float floor_db;
} omx_loudness_config;

/* reference :185-194 LoudnessSnapshot */
typedef struct omx_loudness_snapshot {
    float short_term_loudness;
    float momentary_loudness;
    float rms_fast_db[OMX_MAX_CHANNELS];
    float rms_slow_db[OMX_MAX_CHANNELS];
    float true_peak_db[OMX_MAX_CHANNELS];
    uint32_t channel_count;
    uint8_t positions[OMX_MAX_CHANNELS];
    uint32_t _pad;
} omx_loudness_snapshot;

typedef struct omx_loudness omx_loudness;

void omx_loudness_config_default(omx_loudness_config* out);
int omx_loudness_create(const omx_loudness_config* cfg, omx_loudness** out);      /* ::new :225-232 */
void omx_loudness_destroy(omx_loudness* h);
int omx_loudness_reset_audio(omx_loudness* h);                                    /* :234-236 */
int omx_loudness_process_block(omx_loudness* h, const omx_block* block,
                               omx_loudness_snapshot* out);                       /* :253-311 */
/* reference :22-55 k_weighting_coefficients → b[5], a[5] */
void omx_k_weighting_coefficients(double fs, double b[5], double a[5]);

typedef struct omx_loudness_bank omx_loudness_bank;
/* One call = `n_blocks` consecutive blocks of `block_frames` frames for every
 * stream; a snapshot is produced per (stream, block) exactly as if the
 * reference had been fed block by block (true peak resets per block, :301).
 * d_snapshots: omx_loudness_snapshot [n_streams][n_blocks]. */
int omx_loudness_bank_create(const omx_loudness_config* cfg, uint32_t n_streams, uint32_t channels,
                             omx_loudness_bank** out);
void omx_loudness_bank_destroy(omx_loudness_bank* b);
int omx_loudness_bank_reset_audio(omx_loudness_bank* b);
/* WHICH EVALUATION ORDER A CALL GETS (default, OMX_OPT_KERNEL_FORM = 0; omx_loudness_bank_set_option pins one):
 *   sequential kernels (sliding Kahan-Babuska-Neumaier sums in the reference's order) for single-stream handles (one block per call),
 *       calls of fewer than 4 blocks, streams whose sample counter is off the 64-sample grid, and non-finite PCM;
 *   chunk-parallel kernels (window sums as differences of a double-double running total; K-weighting by a block scan) for lock-step AND
 *       ragged calls of >= 4 blocks (ragged: max_blocks >= 4) WHATEVER the bank size (the sequential kernels cost ~37 us per block however
 *       few streams there are: one stream x 64 blocks 2.37 -> 0.12 ms), whose block length is a multiple of 64 frames and which start
 *       at a multiple of 64 frames since the last reset: 1-8 channels, every sample rate (44.1 / 88.2 kHz windows are off the 64-sample grid: loudness_chunked.hip).
 *       Same quantities to ~1e-15 of a window sum (the K-weighting recurrence runs on fused multiply-adds there); LUFS / RMS within
 *       1e-4 dB of the sequential order (measured 1.5e-5), true peak bit-identical.  OMX_OPT_KERNEL_FORM = 1 pins the sequential kernels. */
/* test hook: which evaluation order the bank's last process call took — 1 = sequential kernels, 2 = chunk-parallel (0 = no call yet) */
int omx_debug_loudness_bank_last_form(const omx_loudness_bank* b);
int omx_loudness_bank_process(omx_loudness_bank* b, const float* pcm, int pcm_on_device,
                              uint64_t block_frames, uint64_t n_blocks, uint32_t channels,
                              float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS],
                              void* stream, const omx_loudness_snapshot** d_snapshots);
/* Ragged call — the reference's VisualManager feeds every stream's LoudnessProcessor its own blocks (registry.rs:396-418): stream s
 * runs n_blocks[s] <= max_blocks blocks of block_frames frames; its row of `pcm` (device memory) is block_frames * max_blocks frames
 * long.  Streams flagged in reset_mask (may be NULL) get reset_audio() first.  d_snapshots is [n_streams][max_blocks], of which stream
 * s filled the first d_n_blocks[s].  The first ragged call moves the bank to per-stream sample counters; lock-step
 * omx_loudness_bank_process calls are refused until omx_loudness_bank_reset_audio. */
typedef struct omx_loudness_ragged_update {
    uint64_t n_streams;
    uint64_t max_blocks;
    const uint32_t* d_n_blocks;               /* device: [n_streams] */
    const omx_loudness_snapshot* d_snapshots; /* device: [n_streams][max_blocks] */
    const uint8_t* d_reset;                   /* device: [n_streams] the call's reset flags */
    const uint32_t* d_block_frames;           /* device: [n_streams] per-stream block length (process_chunks), or NULL (block_frames for all) */
} omx_loudness_ragged_update;
int omx_loudness_bank_process_ragged(omx_loudness_bank* b, const float* pcm, uint64_t block_frames, uint64_t max_blocks,
                                     const uint32_t* n_blocks, const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                     const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_loudness_ragged_update* out);
/* Chunk call — exactly what VisualManager::ingest_samples does with a batcher chunk (registry.rs:396-418; meter.rs:40-69 hands out
 * chunks of 1 ... 4 quanta, each as ONE call): capture s delivers ONE block of frames[s] <= frames_capacity frames (0 = nothing
 * arrived: block.is_empty()), so a 1024-frame catch-up chunk is ONE process_block — one true-peak take over the whole chunk
 * (loudness/processor.rs:287-311) — next to a capture that delivered 256.  `pcm` is device memory [n_streams][frames_capacity][channels];
 * frames[] / reset_mask[] are host arrays.  d_snapshots is [n_streams][1] (max_blocks = 1), d_n_blocks[s] = frames[s] != 0.  Ragged
 * bookkeeping as omx_loudness_bank_process_ragged (the two may be mixed); sequential kernels (the reference's order). */
int omx_loudness_bank_process_chunks(omx_loudness_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                         const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                         const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_loudness_ragged_update* out);
int omx_loudness_bank_fetch(omx_loudness_bank* b, uint64_t stream_index, uint64_t block,
                            omx_loudness_snapshot* dst);
int omx_loudness_bank_kernel_time(omx_loudness_bank* b, double* avg_ms, uint64_t* launches);
int omx_loudness_bank_set_option(omx_loudness_bank* b, uint32_t option, uint64_t value);

/* ===================================================================== *
 * Stereometer — reference src/visuals/stereometer/processor.rs
 * ===================================================================== */

/* reference :11-21 StereometerConfig */
typedef struct omx_stereometer_config {
    float sample_rate;
    float segment_duration;
    uint64_t target_sample_count;
    float correlation_window;
    uint32_t analyze_bands;
    uint32_t emit_band_points;
    uint32_t _pad;
} omx_stereometer_config;

/* reference :23-26 StereometerSnapshot: points[band] = (left,right) pairs,
 * band 0 = full band, 1..3 = low/mid/high */
typedef struct omx_stereometer_snapshot {
    const float* points[4]; /* interleaved l,r; n_points[band] pairs */
    uint64_t n_points[4];
    float correlations[4];
} omx_stereometer_snapshot;

typedef struct omx_stereometer omx_stereometer;

void omx_stereometer_config_default(omx_stereometer_config* out);
int omx_stereometer_create(const omx_stereometer_config* cfg, omx_stereometer** out); /* :75-86 */
void omx_stereometer_destroy(omx_stereometer* h);
int omx_stereometer_get_config(const omx_stereometer* h, omx_stereometer_config* out);
int omx_stereometer_update_config(omx_stereometer* h, const omx_stereometer_config* cfg); /* :183-207 */
int omx_stereometer_reset_audio(omx_stereometer* h);                              /* :92-97 */
int omx_stereometer_process_block(omx_stereometer* h, const omx_block* block,
                                  omx_stereometer_snapshot* out);                 /* :99-182 */

typedef struct omx_stereometer_bank omx_stereometer_bank;
/* d_correlations: f32 [n_streams][n_blocks][4]; d_points: f32 [n_streams][4][target][2]
 * for the newest block; d_produced: uint32 [n_streams][n_blocks]. */
typedef struct omx_stereometer_bank_update {
    uint64_t n_streams;
    uint64_t n_blocks;
    uint64_t target;
    const float* d_correlations;
    const float* d_points;
    const uint32_t* d_produced;
} omx_stereometer_bank_update;
int omx_stereometer_bank_create(const omx_stereometer_config* cfg, uint32_t n_streams,
                                omx_stereometer_bank** out);
void omx_stereometer_bank_destroy(omx_stereometer_bank* b);
int omx_stereometer_bank_update_config(omx_stereometer_bank* b, const omx_stereometer_config* cfg); /* update_config of every stream's processor (stereometer/processor.rs:183-207) */
int omx_stereometer_bank_reset_audio(omx_stereometer_bank* b);
/* WHICH EVALUATION ORDER A CALL GETS (default, OMX_OPT_KERNEL_FORM = 0; omx_stereometer_bank_set_option pins one):
 *   sequential kernels — the reference's operation order: points bit-exact, rho error 0 against the CPU restatement —
 *       for single-stream handles (one block per call), calls of fewer than 4 blocks, channel counts other than 2,
 *       blocks that are not a multiple of 16 frames (or shorter than 32), and any call whose PCM is not finite;
 *   chunk-parallel kernels (every block of the call in parallel, block-boundary states by a scan) for 2-channel lock-step AND ragged calls of
 *       >= 4 blocks WHATEVER the bank size (one stream x 64 blocks 1.75 -> 0.10 ms).  The same f32 band filters evaluated block-parallel on fused multiply-adds: points
 *       within 1e-4 of full scale (measured 2.5e-5), rho within 1e-6 on bands within 16 dB of the full level (measured 6e-8) and
 *       within the reference's own f32 filter noise elsewhere (tests/parity.py::check_chunked_rho).  NOT bit-identical to the
 *       sequential order: a host that needs the reference's exact bits sets OMX_OPT_KERNEL_FORM = 1. */
int omx_stereometer_bank_process(omx_stereometer_bank* b, const float* pcm, int pcm_on_device,
                                 uint64_t block_frames, uint64_t n_blocks, uint32_t channels,
                                 float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS],
                                 void* stream, omx_stereometer_bank_update* out);
/* Ragged call — one StereometerProcessor per capture, fed and reset on its own (registry.rs:396-418): stream s runs
 * n_blocks[s] <= max_blocks blocks of block_frames frames; its row of `pcm` (device memory) is block_frames * max_blocks frames long.
 * Streams flagged in reset_mask (may be NULL) get reset_audio() first.  d_correlations is [n_streams][max_blocks][4] and d_produced
 * [n_streams][max_blocks], of which stream s filled the first d_n_blocks[s]; d_points [n_streams][4][target][2] holds, for every
 * stream whose last block of the call produced a snapshot, the bands flagged in d_band_valid [n_streams][4].  The first ragged call
 * moves the bank to per-stream history positions; lock-step omx_stereometer_bank_process calls — and a change of the segment length —
 * are refused until omx_stereometer_bank_reset_audio. */
typedef struct omx_stereometer_ragged_update {
    uint64_t n_streams;
    uint64_t max_blocks;
    uint64_t target;
    const uint32_t* d_n_blocks;
    const float* d_correlations;
    const uint32_t* d_produced;
    const float* d_points;
    const uint32_t* d_band_valid;
} omx_stereometer_ragged_update;
int omx_stereometer_bank_process_ragged(omx_stereometer_bank* b, const float* pcm, uint64_t block_frames, uint64_t max_blocks,
                                        const uint32_t* n_blocks, const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                        const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_stereometer_ragged_update* out);
/* Chunk call — exactly what VisualManager::ingest_samples does with a batcher chunk (registry.rs:396-418; meter.rs:40-69): capture s
 * delivers ONE block of frames[s] <= frames_capacity frames (0 = nothing arrived); `pcm` is device memory
 * [n_streams][frames_capacity][channels].  Outputs as omx_stereometer_bank_process_ragged with max_blocks = 1 (one correlation row, one
 * produced flag per stream); sequential kernels (the reference's order); may be mixed with process_ragged calls. */
int omx_stereometer_bank_process_chunks(omx_stereometer_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                            const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                            const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_stereometer_ragged_update* out);
int omx_stereometer_bank_fetch(omx_stereometer_bank* b, uint64_t stream_index, uint64_t block,
                               float correlations[4], uint32_t* produced);
/* decimated points of one band of the newest snapshot (`stereometer/processor.rs:152-170`) -> dst[2 * n_pairs] (l, r);
 * n_pairs = 0 when that band produced nothing */
int omx_stereometer_bank_fetch_points(omx_stereometer_bank* b, uint64_t stream_index, uint32_t band, float* dst,
                                      uint64_t dst_capacity_pairs, uint64_t* n_pairs);
/* OMX_OPT_KERNEL_FORM: 0 = choose by call shape (default), 1 = sequential kernels only (reference operation order, bit-identical
 * filters), 2 = chunk-parallel evaluation whenever the shape allows (2 channels, blocks of a multiple of 16 frames, >= 2 blocks) */
/* test hook: 1 = the bank's last process call ran the sequential kernels, 2 = the chunk-parallel ones (0 = no call yet) */
int omx_debug_stereometer_bank_last_form(const omx_stereometer_bank* b);
int omx_stereometer_bank_set_option(omx_stereometer_bank* b, uint32_t option, uint64_t value);

/* ===================================================================== *
 * Oscilloscope — reference src/visuals/oscilloscope/processor.rs
 * ===================================================================== */

enum { OMX_TRIGGER_ZERO_CROSSING = 0, OMX_TRIGGER_STABLE = 1 };

/* reference :33-43 OscilloscopeConfig; TriggerMode (:21-31) is (mode, num_cycles) */
typedef struct omx_oscilloscope_config {
    float sample_rate;
    float segment_duration;
    uint32_t trigger_mode;
    uint32_t trigger_source; /* OMX_CHANNEL_* */
    uint64_t num_cycles;
    uint32_t channel_1;
    uint32_t channel_2;
} omx_oscilloscope_config;

/* reference :553-560 OscilloscopeSnapshot */
typedef struct omx_oscilloscope_snapshot {
    uint64_t epoch;
    uint64_t channels;
    uint64_t slots[2];
    uint64_t samples_per_channel;
    uint64_t n_samples; /* channels * samples_per_channel */
    const float* samples;
} omx_oscilloscope_snapshot;

typedef struct omx_oscilloscope omx_oscilloscope;

void omx_oscilloscope_config_default(omx_oscilloscope_config* out);
int omx_oscilloscope_create(const omx_oscilloscope_config* cfg, omx_oscilloscope** out); /* :579-587 */
void omx_oscilloscope_destroy(omx_oscilloscope* h);
int omx_oscilloscope_get_config(const omx_oscilloscope* h, omx_oscilloscope_config* out);
int omx_oscilloscope_update_config(omx_oscilloscope* h, const omx_oscilloscope_config* cfg); /* :752-758 */
int omx_oscilloscope_reset_audio(omx_oscilloscope* h);                            /* :593-600 */
int omx_oscilloscope_process_block(omx_oscilloscope* h, const omx_block* block,
                                   omx_oscilloscope_snapshot* out);               /* :611-712 */
/* Test/diagnostic view of the trigger state (reference `last_cycle_rate`, :602-609):
 * returns 1 and writes Hz when a period is locked, else 0. */
int omx_oscilloscope_last_cycle_rate(const omx_oscilloscope* h, float* hz);
/* Test-only view, like last_cycle_rate: Capture{start, frac_offset} (`oscilloscope/processor.rs:265-269`) behind the newest
 * snapshot — the resampled trace begins at history sample start + frac_offset.  1 if a snapshot exists, else 0. */
int omx_oscilloscope_last_capture(const omx_oscilloscope* h, uint32_t* start, float* frac_offset);


/* ===================================================================== *
 * Waveform — reference src/visuals/waveform/processor.rs
 * ===================================================================== */

/* reference :32-40 WaveformConfig */
typedef struct omx_waveform_config {
    float sample_rate;
    float scroll_speed;
    uint64_t max_columns;
    uint32_t analyze_bands;
    uint32_t track_history;
} omx_waveform_config;

/* reference :54-62 WaveColumn (11 floats) */
typedef struct omx_wave_column {
    float min;
    float max;
    float color_bands[3];
    float rms_db[2][3];
} omx_wave_column;

/* reference :65-76 WaveformPreview / WaveformUpdate: `columns` = n_columns frames of 4 WaveColumns (Left, Right, Mid, Side) */
typedef struct omx_waveform_update {
    uint64_t n_columns;
    const omx_wave_column* columns; /* [n_columns][4] */
    uint32_t reset;
    uint32_t preview_some;          /* preview.columns.is_some() */
    float preview_progress;
    uint32_t _pad;
    omx_wave_column preview[4];
} omx_waveform_update;

typedef struct omx_waveform omx_waveform;
void omx_waveform_config_default(omx_waveform_config* out);
int omx_waveform_create(const omx_waveform_config* cfg, omx_waveform** out);            /* ::new :135-147 */
void omx_waveform_destroy(omx_waveform* h);
int omx_waveform_get_config(const omx_waveform* h, omx_waveform_config* out);           /* ::config :149-151 */
int omx_waveform_update_config(omx_waveform* h, const omx_waveform_config* cfg);        /* :336-352 */
int omx_waveform_reset_audio(omx_waveform* h);                                          /* :153-155 */
int omx_waveform_prepare(omx_waveform* h);                                              /* :157-161 */
int omx_waveform_process_block(omx_waveform* h, const omx_block* block, omx_waveform_update* out); /* :306-334 */

typedef struct omx_waveform_bank omx_waveform_bank;
/* One block per call for every stream.  d_columns: omx_wave_column [n_streams][n_columns][4] (already capped to the
 * newest max_columns); d_preview: omx_wave_column [n_streams][4]. */
typedef struct omx_waveform_bank_update {
    uint64_t n_streams;
    uint64_t n_columns;
    const omx_wave_column* d_columns;
    const omx_wave_column* d_preview;
    uint32_t reset;
    uint32_t preview_some;
    float preview_progress;
    uint32_t _pad;
} omx_waveform_bank_update;
int omx_waveform_bank_create(const omx_waveform_config* cfg, uint32_t n_streams, omx_waveform_bank** out);
void omx_waveform_bank_destroy(omx_waveform_bank* b);
int omx_waveform_bank_update_config(omx_waveform_bank* b, const omx_waveform_config* cfg); /* update_config of every stream's processor (waveform/processor.rs:336-352) */
int omx_waveform_bank_reset_audio(omx_waveform_bank* b);
int omx_waveform_bank_process(omx_waveform_bank* b, const float* pcm, int pcm_on_device, uint64_t frames,
                              uint32_t channels, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS],
                              void* stream, omx_waveform_bank_update* out);
/* Ragged call — one WaveformProcessor per capture, fed and reset on its own (registry.rs:396-418): stream s receives
 * frames[s] <= frames_capacity frames; its row of `pcm` (device memory) is frames_capacity frames long.  Streams flagged in reset_mask
 * (may be NULL) get reset_audio() first (d_reset[s] = 1: the consumer clears its history, as for `reset` of the lock-step update).
 * d_columns is [n_streams][max_columns][4], of which stream s filled the first d_n_columns[s]; d_preview [n_streams][4] is valid where
 * d_preview_progress[s] > 0.  frames_capacity must be small enough that a call cannot emit more than the configuration's max_columns
 * columns.  The first ragged call moves the bank to per-stream push counts and column phases; lock-step omx_waveform_bank_process
 * calls are refused until omx_waveform_bank_reset_audio. */
typedef struct omx_waveform_ragged_update {
    uint64_t n_streams;
    uint64_t max_columns;
    const uint32_t* d_n_columns;
    const omx_wave_column* d_columns;
    const omx_wave_column* d_preview;
    const float* d_preview_progress;
    const uint8_t* d_reset;
} omx_waveform_ragged_update;
int omx_waveform_bank_process_ragged(omx_waveform_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                     const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                     const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_waveform_ragged_update* out);
/* WHICH EVALUATION ORDER A LOCK-STEP CALL GETS (OMX_OPT_KERNEL_FORM; default 0 = by call shape):
 *   1 = the sequential kernels only: the reference's operation order on every value, bit-identical to it
 *       (`waveform/processor.rs:252-291`);
 *   2 = the chunk-parallel form (waveform_chunked.hip) whenever the shape allows: 2 channels, device PCM, band analysis on, an even
 *       frame count >= 1024, at most ~4000 columns in the call.  The band filters restart every 64 ... 256 frames from states
 *       obtained by a scan (one f32 rounding each; the low band's in f64), the window means are differences of double-double
 *       running totals: min / max fields bit-identical, colour bands and RMS history within the bars of
 *       tests/test_gpu_parity_meters.py of the sequential order.  Non-finite or absurdly large (> 1e18) samples send the whole call
 *       through the sequential kernels.  By shape (0) it serves calls of >= 2048 frames (the sequential kernels cost ~146 ns per frame
 *       whatever the bank size; the chunk form ~0.2 ms of launches plus its work): the reference's cadence — one 256-frame block per
 *       call — and every call shorter than 2048 frames stay bit-identical.
 * Ragged calls take the chunk-parallel form too while their streams fall into at most 8 lock-step groups (the same frame count, push
 * count and column phase: the host mirrors both counters; a stream the call resets joins the group of fresh streams); otherwise the
 * sequential kernels run. */
int omx_waveform_bank_set_option(omx_waveform_bank* b, uint32_t option, uint64_t value);
/* test hook: 1 = the bank's last lock-step call ran the sequential kernels alone, 2 = the chunk-parallel form (0 = no call yet) */
int omx_debug_waveform_bank_last_form(const omx_waveform_bank* b);
int omx_waveform_bank_fetch(omx_waveform_bank* b, uint64_t stream_index, omx_wave_column* columns /* [n_columns][4] */,
                            omx_wave_column* preview /* [4] or NULL */);

/* ===================================================================== *
 * DspBatcher — reference src/meter.rs:15-80, :145-166
 * The caller of the processors: turns capture packets into the fixed block
 * partition the per-block snapshots depend on (256 frames @ 48 kHz scaled by
 * the sample rate, backlog chunks of up to 1024 frames, <= 2 s of silence
 * replayed, format change never mixes generations).  Host-side integer logic;
 * every emitted block is handed to `ingest` (VisualManager::ingest_samples,
 * src/visuals/registry.rs:396-418), `reset` mirrors VisualManager::reset_audio.
 * ===================================================================== */
typedef struct omx_audio_format { /* reference src/dsp.rs:79-85 AudioFormat */
    uint64_t generation;
    float sample_rate;
    uint32_t channels;
    uint8_t positions[OMX_MAX_CHANNELS];
} omx_audio_format;
typedef void (*omx_ingest_fn)(void* user, const float* samples, uint64_t n_samples, const omx_audio_format* format);
typedef void (*omx_reset_fn)(void* user);
typedef struct omx_batcher omx_batcher;
int omx_batcher_create(omx_batcher** out);                                     /* DspBatcher::new :33-38 */
void omx_batcher_destroy(omx_batcher* b);
/* DspBatcher::push :40-69 — returns the number of ingest calls made */
uint64_t omx_batcher_push(omx_batcher* b, const float* samples, uint64_t n_samples, const omx_audio_format* format,
                          omx_ingest_fn ingest, void* user);
/* ingest_silence :145-166 — returns the number of ingest calls made (0 after a reset) */
uint64_t omx_batcher_push_silence(omx_batcher* b, uint64_t frames, const omx_audio_format* format, omx_ingest_fn ingest,
                                  omx_reset_fn reset, void* user);
void omx_batcher_reset(omx_batcher* b, omx_reset_fn reset, void* user);        /* :71-74 */
void omx_batcher_clear(omx_batcher* b);                                        /* :76-79 */
/* pending samples (batcher.samples.len()); copies up to `cap` of them to dst when dst != NULL */
uint64_t omx_batcher_pending(const omx_batcher* b, float* dst, uint64_t cap);
/* 1 and *out filled when a format is latched (batcher.format), else 0 */
int omx_batcher_format(const omx_batcher* b, omx_audio_format* out);

/* ---- S DspBatchers with their samples resident on the device (meter.rs:27-80, one DspBatcher per capture; registry.rs:396-418) ----
 * The packet LENGTHS are host values (they come from the capture API); the chunk plan — DspBatcher::push's integer arithmetic
 * (meter.rs:40-69) — is a pure function of them and of the pending counts, so it is made on the host, and the SAMPLES never leave the
 * device: one launch assembles the chunks of every capture, a second one keeps the remainders.  A push yields ROUNDS: round r holds
 * the r-th chunk of every capture that has one, laid out as omx_capture_group_ingest_ragged takes it (one block per capture per call).
 * No CPU fallback: create returns OMX_ERR_NO_DEVICE without a gfx950 device. */
typedef struct omx_batcher_bank omx_batcher_bank;
int omx_batcher_bank_create(uint32_t n_captures, uint64_t max_packet_frames, omx_batcher_bank** out);
void omx_batcher_bank_destroy(omx_batcher_bank* b);
/* One packet per capture: capture s delivers packet_frames[s] <= max_packet_frames frames (0: nothing arrived), interleaved, the first
 * at d_packets + s * packet_stride * format->channels (device memory; packet_frames: host).  clear_mask[s] != 0 (NULL: none) =
 * DspBatcher::clear of capture s before its packet (:76-79: that capture's reset).  A `format` that differs from the previous push's
 * drops every capture's pending samples (:46-48).  *n_rounds = the number of rounds this push produced (0: no capture completed a batch). */
int omx_batcher_bank_push(omx_batcher_bank* b, const float* d_packets, uint64_t packet_stride, const uint32_t* packet_frames,
                          const uint8_t* clear_mask, const omx_audio_format* format, void* stream, uint32_t* n_rounds);
/* ingest_silence (meter.rs:145-166) for every capture: capture s is fed silence_frames[s] frames of silence (0: none; host array) the
 * way the reference feeds it — through DspBatcher::push in pieces of the silence scratch — or, beyond MAX_SILENCE_SECONDS = 2 s of
 * it, is reset instead (:154-157): its pending samples are dropped and reset_out[s] (host, may be NULL) is set to 1 — the caller
 * passes that as the reset mask of its next omx_capture_group_ingest_ragged.  Rounds as after a push (only a capture's FIRST chunk can
 * hold samples — its pending partial batch completed with zeros; every later chunk is zeros and shares one buffer). */
int omx_batcher_bank_push_silence(omx_batcher_bank* b, const uint64_t* silence_frames, const omx_audio_format* format, void* stream,
                                  uint32_t* n_rounds, uint8_t* reset_out);
/* Round r of the last push: *d_pcm = device [n_captures][*chunk_capacity][channels] f32, capture s's chunk = its first (*frames)[s]
 * frames; *frames = host [n_captures], valid until the next push.  (d_pcm, *chunk_capacity, *frames) are the (d_pcm, frames_capacity,
 * frames) of omx_capture_group_ingest_ragged; enqueue that call on the stream the push was given. */
int omx_batcher_bank_round(const omx_batcher_bank* b, uint32_t r, const float** d_pcm, uint64_t* chunk_capacity, const uint32_t** frames);
/* pending samples of capture s (batcher.samples.len()); copies up to `cap` of them to dst (host) when dst != NULL (synchronises `stream`) */
uint64_t omx_batcher_bank_pending(omx_batcher_bank* b, uint32_t s, float* dst, uint64_t cap, void* stream);

/* ---- batched bank: S independent OscilloscopeProcessors, one workgroup per stream ---- */
typedef struct omx_oscilloscope_bank omx_oscilloscope_bank;
/* per (stream, block) result header; `produced` mirrors `process_block(..).is_some()`,
 * `locked`/`period` mirror the trigger behind `last_cycle_rate` (:602-609). */
typedef struct omx_oscilloscope_block_header {
    uint32_t produced;
    uint32_t channels;
    uint32_t slots[2];
    uint32_t samples_per_channel;
    uint32_t locked;
    float period;
    uint32_t capture_start;  /* Capture::start / frac_offset of the first captured trace (`:265-269`): where the resampled */
    float capture_frac;      /* trace begins in the trace history, in samples — diagnostics for the parity tests        */
    uint32_t _pad;
} omx_oscilloscope_block_header;
/* d_headers: [n_streams][n_blocks]; d_samples: f32 [n_streams][2][sample_stride], the snapshot of
 * the newest block (first samples_per_channel entries of each of the `channels` rows are valid). */
typedef struct omx_oscilloscope_bank_update {
    uint64_t n_streams;
    uint64_t n_blocks;
    uint64_t epoch;
    uint64_t sample_stride;
    const omx_oscilloscope_block_header* d_headers;
    const float* d_samples;
} omx_oscilloscope_bank_update;
int omx_oscilloscope_bank_create(const omx_oscilloscope_config* cfg, uint32_t n_streams,
                                 omx_oscilloscope_bank** out);
void omx_oscilloscope_bank_destroy(omx_oscilloscope_bank* b);
int omx_oscilloscope_bank_update_config(omx_oscilloscope_bank* b, const omx_oscilloscope_config* cfg); /* update_config of every stream's processor (oscilloscope/processor.rs:752-759) */
int omx_oscilloscope_bank_reset_audio(omx_oscilloscope_bank* b);
int omx_oscilloscope_bank_process(omx_oscilloscope_bank* b, const float* pcm, int pcm_on_device,
                                  uint64_t block_frames, uint64_t n_blocks, uint32_t channels,
                                  float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS],
                                  void* stream, omx_oscilloscope_bank_update* out);
/* Ragged call — one OscilloscopeProcessor per capture, fed and reset on its own (registry.rs:396-418): stream s runs
 * n_blocks[s] <= max_blocks blocks of block_frames frames; its row of `pcm` (device memory) is block_frames * max_blocks frames long.
 * Streams flagged in reset_mask (may be NULL) get reset_audio() first (their d_epochs[s] advances).  d_headers is
 * [n_streams][max_blocks], of which stream s filled the first d_n_blocks[s]; d_samples holds every stream's newest snapshot.  The first
 * ragged call moves the bank to per-stream ring positions; lock-step omx_oscilloscope_bank_process calls are refused until
 * omx_oscilloscope_bank_reset_audio. */
typedef struct omx_oscilloscope_ragged_update {
    uint64_t n_streams;
    uint64_t max_blocks;
    const uint32_t* d_n_blocks;                        /* device: [n_streams] */
    const uint64_t* d_epochs;                          /* device: [n_streams] */
    const omx_oscilloscope_block_header* d_headers;    /* device: [n_streams][max_blocks] */
    const float* d_samples;                            /* device: [n_streams][2][4096] (the newest snapshot of every stream) */
} omx_oscilloscope_ragged_update;
int omx_oscilloscope_bank_process_ragged(omx_oscilloscope_bank* b, const float* pcm, uint64_t block_frames, uint64_t max_blocks,
                                         const uint32_t* n_blocks, const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                         const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_oscilloscope_ragged_update* out);
/* Chunk call — exactly what VisualManager::ingest_samples does with a batcher chunk (registry.rs:396-418; meter.rs:40-69): capture s
 * delivers ONE block of frames[s] <= frames_capacity frames (0 = nothing arrived) and gets ONE trigger evaluation — one
 * StableTrigger state update (period smoothing, reference EMA, missed-hold counter; oscilloscope/processor.rs:611-712) — however
 * many quanta the chunk holds.  `pcm` is device memory [n_streams][frames_capacity][channels].  Outputs as
 * omx_oscilloscope_bank_process_ragged with max_blocks = 1.  Keep frames_capacity constant over a bank's chunk calls (it sizes the rings). */
int omx_oscilloscope_bank_process_chunks(omx_oscilloscope_bank* b, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                             const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                             const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_oscilloscope_ragged_update* out);
int omx_oscilloscope_bank_fetch(omx_oscilloscope_bank* b, uint64_t stream_index, uint64_t block,
                                omx_oscilloscope_block_header* header, float* samples);
/* test hook (88.2 ... 192 kHz: the wide trigger pass runs on less LDS than its worst case and hands a stream's blocks over to the
 * one-workgroup-per-stream kernel from the first block whose arrays do not fit): the block at which stream `stream_index` was handed over in
 * the bank's last call — the call's block count when it never was — or -1 when the call did not run that form */
long long omx_debug_oscilloscope_bank_resume_block(const omx_oscilloscope_bank* b, uint32_t stream_index);

/* ===================================================================== *
 * State-side summary reductions — the small per-snapshot
 * reductions the reference's visual *states* run on every snapshot and that
 * feed labels / peak markers; batched here so that per-stream summary rows
 * can be produced without shipping spectra or snapshots to the host.
 * ===================================================================== */

/* peak_bin + interpolated_peak, reference src/visuals/spectrum/state.rs:320-356.
 * `found` = 0 when no interior bin inside [min_f, max_f] has a finite level. */
typedef struct omx_spectrum_peak {
    uint32_t found;
    uint32_t bin;
    float freq_hz;   /* (center + parabolic offset * bin_hz).max(0) */
    float level_db;  /* interpolated level, never below the centre bin */
} omx_spectrum_peak;
/* rows of dB levels: row r = db[r * row_stride .. + n_bins); one omx_spectrum_peak per row.
 * on_device = 0: bins/db/out are host pointers (synchronous); 1: device pointers, enqueued on `stream`. */
int omx_spectrum_peaks(const float* bins, const float* db, int on_device, uint64_t n_bins, uint64_t n_rows,
                       uint64_t row_stride, float min_f, float max_f, void* stream, omx_spectrum_peak* out);

/* MeterMode, reference src/visuals.rs:89-95 */
enum {
    OMX_METER_LUFS_SHORT_TERM = 0,
    OMX_METER_LUFS_MOMENTARY = 1,
    OMX_METER_RMS_FAST = 2,
    OMX_METER_RMS_SLOW = 3,
    OMX_METER_TRUE_PEAK = 4
};
/* PeakHold, reference src/visuals/loudness/state.rs:36-60 (2 s hold, then 60 dB/s); the wall clock
 * (`Instant`) is replaced by a caller-supplied time in seconds. */
typedef struct omx_peak_hold {
    float db;
    uint32_t _pad;
    double decay_from;
} omx_peak_hold;
/* one applied snapshot: visible_values() (:178-184, channel_side aggregation :222-246) and the three
 * peak holds after update_peak_holds (:211-217; values clamped to DB_RANGE = [-60, 4]). */
typedef struct omx_meter_row {
    float values[3]; /* left bar, right bar (left_mode aggregated per side), right_mode value */
    float peaks[3];
} omx_meter_row;
/* PeakHold::new(DB_RANGE.0, now) for `n` holds (LoudnessState::new / reset_audio / set_modes) */
int omx_peak_holds_reset(omx_peak_hold* holds, int on_device, uint64_t n, double now, void* stream);
/* snapshots [n_streams][n_blocks] applied in order, block k at time t0 + k * dt;
 * holds [n_streams][3] in/out; rows [n_streams][n_blocks] out. */
int omx_loudness_meters(const omx_loudness_snapshot* snapshots, int on_device, uint64_t n_streams,
                        uint64_t n_blocks, uint32_t left_mode, uint32_t right_mode, double t0, double dt,
                        omx_peak_hold* holds, void* stream, omx_meter_row* rows);

/* ===================================================================== *
 * Reassigned-splat accumulation + dB resolve — the consumer
 * immediately after the spectrogram path: reference
 * src/visuals/render/shaders/spectrogram.wgsl:126-147 (vs_accum_splat),
 * :215-237 (fs_accum, fs_resolve), src/util/audio/frequency.rs:15-37.
 * The reference runs this as two raster passes over an Rg16Float target; here the
 * same per-point math feeds an f32 grid (no half-float saturation channel), and
 * quad coverage follows the rasteriser's pixel-centre / top-left rule.
 * omx_spectrogram_splat takes columns oldest -> newest (age = n_columns - 1 - column);
 * omx_spectrogram_history_* keeps the renderer's slot ring (below) and splats from it.
 * ===================================================================== */
enum { OMX_FREQ_SCALE_LINEAR = 0, OMX_FREQ_SCALE_LOGARITHMIC = 1, OMX_FREQ_SCALE_ERB = 2 };
typedef struct omx_splat_view {
    float extent_x;       /* physical pixels along time (bounds.width * scale_factor), newest column at the right edge */
    float extent_y;       /* physical pixels along frequency, top = freq_max */
    float scale_factor;   /* physical pixels per column and per splat side (>= 1) */
    uint32_t freq_scale;  /* OMX_FREQ_SCALE_* */
    float freq_min;       /* display_axis(sample_rate): min(1 Hz, nyquist / 2) (spectrogram/state.rs:48-51) */
    float freq_max;       /* nyquist */
    float uv_lo, uv_hi;   /* zoom window on the normalised frequency axis ([0, 1] = everything) */
    float tilt_db;        /* dB / octave around 1 kHz, 0 = off */
    uint32_t width;       /* out: ceil(max(extent_x, 1)) — filled by omx_splat_view_size */
    uint32_t height;      /* out: ceil(max(extent_y, 1)) */
} omx_splat_view;
void omx_splat_view_size(omx_splat_view* view);
/* points [n_streams][n_columns][column_stride], counts [n_streams][n_columns] (the spectrogram bank's d_points / d_counts);
 * accum (power) and db (resolved level, -inf where nothing landed) are f32 [n_streams][width][height]: time-major, one
 * time column (all frequencies of one x) is contiguous, row 0 = freq_max.
 * on_device = 0: all four are host pointers (synchronous); 1: device pointers, enqueued on `stream`. */
int omx_spectrogram_splat(const omx_spectrogram_point* points, const uint32_t* counts, int on_device,
                          uint64_t n_streams, uint64_t n_columns, uint64_t column_stride,
                          float reassigned_power_scale, const omx_splat_view* view, void* stream,
                          float* accum, float* db);


/* ---- the column history ring between the processor and the splat passes --------------------------------------------
 * SpectrogramHistory::apply_update (`spectrogram/state.rs:53-175`: reset, capacity changes with remap_retained, one slot per
 * new column, slot_counts) together with the renderer's ring buffer it drives (`spectrogram/render.rs:457-597`: resize copy
 * plan, slot uploads) and the accumulation pass over it (`render.rs:106-160`; age = (newest_col + hl - slot) % hl,
 * `spectrogram.wgsl:141-142`).  The ring lives in device memory: [n_streams][ring_capacity][fft_size / 2 + 1] points (or u16
 * codes for classic columns) + slot_counts [n_streams][ring_capacity]; the bookkeeping integers are common to the streams of a
 * lock-step bank.  `reassigned_points_per_slot` is tracked (state.rs:131-148) but only sizes the reference's vertex buffer. */
typedef struct omx_spectrogram_history omx_spectrogram_history;
typedef struct omx_spectrogram_history_info {
    uint32_t kind;                       /* OMX_COLUMN_* of the ring */
    uint32_t ring_capacity;
    uint32_t write_slot;
    uint32_t col_count;
    uint32_t points_per_column;          /* fft_size / 2 + 1 */
    uint32_t reassigned_points_per_slot; /* of stream 0 */
    uint32_t newest_slot;                /* (write_slot + ring_capacity - 1) % ring_capacity (render.rs:221) */
    uint32_t visible_slots;              /* min(col_count, ring_capacity) (render.rs:106) */
} omx_spectrogram_history_info;
int omx_spectrogram_history_create(uint32_t n_streams, omx_spectrogram_history** out);
void omx_spectrogram_history_destroy(omx_spectrogram_history* h);
/* apply one single-stream update (host columns, n_streams must be 1): SpectrogramState::apply_snapshot -> history.apply_update */
int omx_spectrogram_history_apply(omx_spectrogram_history* h, const omx_spectrogram_update* update);
/* apply one bank update (device-resident columns of every stream), enqueued on `stream` */
int omx_spectrogram_bank_history_apply(omx_spectrogram_history* h, const omx_spectrogram_bank_update* update, void* stream);
int omx_spectrogram_history_get_info(omx_spectrogram_history* h, omx_spectrogram_history_info* out);
/* slot_counts of one stream -> out[0 .. min(capacity, ring_capacity)) ; returns ring_capacity (or < 0) */
int64_t omx_spectrogram_history_slot_counts(omx_spectrogram_history* h, uint64_t stream_index, uint32_t* out, uint64_t capacity);
/* copy one ring slot of one stream to host memory: points (12 B each) or u16 codes; n_out = valid entries of the slot */
int omx_spectrogram_history_fetch_slot(omx_spectrogram_history* h, uint64_t stream_index, uint32_t slot, void* dst,
                                       uint64_t dst_capacity_elems, uint64_t* n_out);
/* accumulation + resolve over the visible slots of the ring; accum / db as in omx_spectrogram_splat ([n_streams][width][height]);
 * on_device = 0: host outputs (synchronous), 1: device outputs, enqueued on `stream`.  A classic ring yields an empty image. */
int omx_spectrogram_history_splat(omx_spectrogram_history* h, float reassigned_power_scale, const omx_splat_view* view,
                                  int on_device, void* stream, float* accum, float* db);

/* ===================================================================== *
 * Capture group — VisualManager::ingest_samples, reference src/visuals/registry.rs:396-418:
 * one block of a capture goes to EVERY enabled visual (`for entry in &mut self.entries { if entry.enabled { module.ingest(&block) } }`),
 * and VisualManager::reset_audio (:360-365) resets every one of them.  Here the group owns one bank per enabled visual for
 * `n_streams` captures advancing in lock step; one omx_capture_group_ingest call
 *   - projects the block ONCE for the visuals that keep pending audio (Spectrogram and Spectrum are fed by a single ingest launch),
 *   - runs the Spectrogram / Spectrum banks and the Oscilloscope bank on the caller's stream and the other meter banks (Loudness,
 *     Stereometer, Waveform) on three streams of its own beside them, joined before the call returns to the caller's stream order,
 *   - and, on request (OMX_OPT_GROUP_STATS), leaves the per-stream summary rows — the table that is gathered over RCCL once per
 *     epoch — in device memory: [n_streams][OMX_STATS_COLUMNS] f32 =
 *       momentary LUFS, short-term LUFS, max true peak dBTP, rho full / low / mid / high (newest block), columns of this call,
 *       mean points per column, points of the newest column, held true-peak bar left / right (2 s hold, 60 dB/s: loudness/state.rs:36-60).
 * A multi-GPU host all-gathers `d_stats_rows` itself (ncclAllGather on its communicator; INTEGRATION.md shows the call): the library
 * does not link RCCL.
 * `pcm` is device memory [n_streams][frames][channels].  One call = ONE AudioBlock of `frames` frames to every visual, exactly as
 * ingest_samples builds it (registry.rs:407-417) from whatever chunk DspBatcher::push handed over — 256 frames at the regular cadence,
 * 512 / 768 / 1024 when catching up after a stall, each chunk as ONE call (meter.rs:61-64): one oscilloscope trigger evaluation
 * (oscilloscope/processor.rs:611-712), one true-peak take (loudness/processor.rs:287-311), one stereo_channels scan (dsp.rs:190-213)
 * over the whole chunk.  A host that queues chunks and replays several of them in one call sets cfg.block_frames = B explicitly: a call
 * whose `frames` is a multiple of B is then seen by the block-based visuals (Loudness, Stereometer, Oscilloscope) as frames / B blocks
 * of B frames, one snapshot per block, as if ingest had been called once per block (update.n_blocks / block_frames say which).
 * ===================================================================== */
enum {
    OMX_VISUAL_SPECTROGRAM = 1,
    OMX_VISUAL_SPECTRUM = 2,
    OMX_VISUAL_LOUDNESS = 4,
    OMX_VISUAL_STEREOMETER = 8,
    OMX_VISUAL_OSCILLOSCOPE = 16,
    OMX_VISUAL_WAVEFORM = 32
};
#define OMX_STATS_COLUMNS 12
enum {
    OMX_OPT_GROUP_STATS = 16,      /* value != 0: assemble the summary rows in every ingest call (needs Spectrogram, Loudness, Stereometer) */
    OMX_OPT_GROUP_SHARED_INGEST = 17 /* value == 0: every bank runs its own ingest launch (A/B against the shared one; same rings) */
};
typedef struct omx_capture_group_config {
    uint32_t n_streams;
    uint32_t visuals;                /* OMX_VISUAL_* bits */
    uint32_t block_frames;           /* 0 (default) = every ingest call is one block, the reference's partition; B > 0 = replay mode:
                                        a call of k * B frames is k blocks of B frames to the block-based visuals */
    uint32_t spectrum_emit_all_hops; /* as omx_spectrum_bank_create */
    omx_spectrogram_config spectrogram;
    omx_spectrum_config spectrum;
    omx_loudness_config loudness;
    omx_stereometer_config stereometer;
    omx_oscilloscope_config oscilloscope;
    omx_waveform_config waveform;
} omx_capture_group_config;
typedef struct omx_capture_group_update {
    uint32_t produced;               /* OMX_VISUAL_* bits: which visuals produced an update in this call */
    uint32_t ingest_launches;        /* ingest (projection) launches this call made: 1 when Spectrogram and Spectrum shared one */
    uint64_t n_blocks, block_frames; /* how the block-based visuals saw the call */
    omx_spectrogram_bank_update spectrogram;
    omx_spectrum_bank_update spectrum;
    const omx_loudness_snapshot* d_loudness; /* [n_streams][n_blocks] */
    omx_stereometer_bank_update stereometer;
    omx_oscilloscope_bank_update oscilloscope;
    omx_waveform_bank_update waveform;
    const float* d_stats_rows;       /* [n_streams][OMX_STATS_COLUMNS], or NULL (OMX_OPT_GROUP_STATS off / a visual it needs disabled) */
} omx_capture_group_update;
/* what omx_capture_group_ingest_ragged leaves behind: every enabled bank's own ragged update (per-stream counts in device memory) */
typedef struct omx_capture_group_ragged_update {
    uint32_t produced;               /* OMX_VISUAL_* bits: which visuals produced an update for at least one capture */
    uint32_t ingest_launches;        /* projection launches this call made: 1 when Spectrogram and Spectrum shared one */
    uint64_t block_frames;           /* how the block-based visuals saw the call: 0 = capture s ran ONE block of frames[s] frames
                                        (max_blocks = 1); B = cfg.block_frames: capture s ran frames[s] / B blocks ... */
    uint64_t max_blocks;             /* ... of at most frames_capacity / B */
    omx_spectrogram_ragged_update spectrogram;
    omx_spectrum_ragged_update spectrum;
    omx_loudness_ragged_update loudness;
    omx_stereometer_ragged_update stereometer;
    omx_oscilloscope_ragged_update oscilloscope;
    omx_waveform_ragged_update waveform;
    const float* d_stats_rows;       /* [n_streams][OMX_STATS_COLUMNS] (OMX_OPT_GROUP_STATS; Spectrogram, Loudness and Stereometer enabled), or NULL.
                                        Per-capture rows: a capture that delivered nothing in this call keeps its row (column 7, the
                                        columns of this call, reads 0); its peak holds run on the capture's own sample clock */
} omx_capture_group_ragged_update;
typedef struct omx_capture_group omx_capture_group;
void omx_capture_group_config_default(omx_capture_group_config* out); /* every visual's default config, none enabled, 1 stream */
int omx_capture_group_create(const omx_capture_group_config* cfg, omx_capture_group** out);
void omx_capture_group_destroy(omx_capture_group* g);
int omx_capture_group_reset_audio(omx_capture_group* g);              /* VisualManager::reset_audio (:360-365) */
int omx_capture_group_set_option(omx_capture_group* g, uint32_t option, uint64_t value); /* OMX_OPT_GROUP_*, OMX_OPT_KERNEL_TIMING */
int omx_capture_group_ingest(omx_capture_group* g, const float* pcm, uint64_t frames, uint32_t channels, float sample_rate,
                             const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_capture_group_update* out);
/* ---- the rest of VisualManager (registry.rs:266-277, :343-365, :396-418) ----
 * set_enabled: `visual` is ONE OMX_VISUAL_* bit.  A disabled visual is skipped by ingest and keeps its state (Entry.enabled gates
 *   module.ingest, :413-417); enabling prepares it (Entry::set_enabled -> module.prepare(), :272-277) — a visual that was not in
 *   cfg->visuals at creation gets its bank here, from the config the group holds for it.
 * update_config: Entry::apply_settings -> processor.update_config (:54-58, :266-270).  `config` points to the omx_<visual>_config of
 *   that visual (omx_spectrogram_config for OMX_VISUAL_SPECTROGRAM, ...); it takes effect between two ingest calls exactly as
 *   omx_<visual>_bank_update_config does (a spectrogram hop change mid-stream keeps the pending samples, `reset` is carried into the
 *   next update).  OMX_VISUAL_LOUDNESS -> OMX_ERR_INVALID: LoudnessProcessor has no update_config (loudness/processor.rs:225-253).
 *   The config of a disabled / not yet created visual is stored and used when it is enabled.
 * note_format: ingest_samples' format-generation rule (:400-406): the host passes AudioFormat.generation before the ingest of a
 *   chunk; a value different from the previous one resets every visual first (returns 1), the first value and equal values do
 *   nothing (0).  omx_capture_group_reset_audio forgets the generation (format_generation = None, :361).
 * ingest_ragged: per-capture frame counts and per-capture reset flags — one VisualManager per capture in the reference, each fed by
 *   its own batcher and reset on its own.  `pcm` is device memory [n_streams][frames_capacity][channels]; capture s delivers its
 *   first frames[s] <= frames_capacity frames (0: nothing arrived) after reset_audio() where reset_mask[s] != 0 (reset_mask may be
 *   NULL).  frames[] / reset_mask[] are HOST arrays.  What capture s delivers is ONE block of frames[s] frames — the chunk its batcher
 *   handed over, whole (meter.rs:40-69: 1 ... 4 quanta per chunk; registry.rs:407-417: one AudioBlock per call) — so captures that
 *   deliver 256, 768 and 1024 frames in the same call each get one process_block of that length (the block-based banks go through
 *   omx_<visual>_bank_process_chunks).  With cfg.block_frames = B != 0 (replay mode) the counts must be multiples of B and capture s
 *   runs frames[s] / B blocks (omx_<visual>_bank_process_ragged).  Per stream the results equal a single-stream handle fed the same
 *   sequence of blocks.  A per-capture reset reaches the banks of DISABLED visuals too (VisualManager::reset_audio resets every
 *   entry, :360-365): it is applied by the bank's next call after the visual is enabled again.  After the first ragged call the group's positions are per capture: omx_capture_group_ingest is refused
 *   (OMX_ERR_INVALID) until omx_capture_group_reset_audio. */
int omx_capture_group_set_enabled(omx_capture_group* g, uint32_t visual, int enabled);
int omx_capture_group_enabled(const omx_capture_group* g);            /* OMX_VISUAL_* bits ingest currently feeds */
int omx_capture_group_update_config(omx_capture_group* g, uint32_t visual, const void* config, void* stream);
int omx_capture_group_note_format(omx_capture_group* g, uint64_t generation);
int omx_capture_group_ingest_ragged(omx_capture_group* g, const float* pcm, uint64_t frames_capacity, const uint32_t* frames,
                                    const uint8_t* reset_mask, uint32_t channels, float sample_rate,
                                    const uint8_t positions[OMX_MAX_CHANNELS], void* stream, omx_capture_group_ragged_update* out);
/* average duration of the spectrogram bank's column kernel since the last call (OMX_OPT_KERNEL_TIMING), as
 * omx_spectrogram_bank_kernel_time */
int omx_capture_group_kernel_time(omx_capture_group* g, double* avg_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* OMX_H */
