// Host side of the spectrogram path: the integer state machine of
// reference src/visuals/spectrogram/processor.rs:170-544 (config normalisation, pending-audio
// bookkeeping, the bit-exact frame-index rule `ready = (pending - read_len)/hop + 1`, retention/skip,
// pending_skip for hop > window, reset flag) driving the HIP kernels.  All sample data stays on the
// device: per-stream mono rings, `audio_last_nonzero` as a device-side absolute position.
#include "spectrogram.hpp"

namespace omx {

constexpr size_t kDefaultFftSize = 2048;                      // :58
constexpr size_t kDefaultHopSize = 64;                        // :59
constexpr size_t kMaxHistoryColumns = 8192;                   // :60
constexpr size_t kHistoryByteBudget = 128u * 1024u * 1024u;   // :61

void spectrogram_config_default(omx_spectrogram_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->sample_rate = kDefaultSampleRate;
    c->window = OMX_WINDOW_HANN;
    c->fft_size = kDefaultFftSize;
    c->hop_size = kDefaultHopSize;
    c->history_length = 0;
    c->zero_padding_factor = 1;
    c->use_reassignment = 1;
}

static void normalize(omx_spectrogram_config& c) {  // :71-82
    c.sample_rate = sanitize_sample_rate(c.sample_rate);
    if (c.fft_size == 0) c.fft_size = kDefaultFftSize;
    if (c.hop_size == 0) c.hop_size = std::max<uint64_t>(std::min<uint64_t>(kDefaultHopSize, c.fft_size), 1);
    c.zero_padding_factor = std::max<uint64_t>(c.zero_padding_factor, 1);
    c.use_reassignment = c.use_reassignment ? 1 : 0;
    c._pad = 0;
}

uint64_t col_byte_stride(uint32_t kind, uint32_t points) {  // :144-151
    if (kind == OMX_COLUMN_REASSIGNED) return (uint64_t)points * sizeof(omx_spectrogram_point);
    return (((uint64_t)points + 1) / 2) * 4;
}
uint64_t history_columns(uint32_t kind, uint32_t points, uint64_t requested) {  // :153-158
    const uint64_t clamped = std::min<uint64_t>(std::max<uint64_t>(requested, 1), kMaxHistoryColumns);
    const uint64_t budget = (uint64_t)kHistoryByteBudget * (1 + (kind == OMX_COLUMN_REASSIGNED ? 1 : 0)) /
                            std::max<uint64_t>(col_byte_stride(kind, points), 1);
    return std::min(clamped, budget);
}
uint16_t pack_classic_db_host(float db) {  // :103-108
    const float SCALE = 65535.0f / 156.0f;
    float v = std::round((db - (-144.0f)) * SCALE);
    v = v < 0.0f ? 0.0f : (v > 65535.0f ? 65535.0f : v);
    return (v == v) ? (uint16_t)v : (uint16_t)0;
}
static size_t hilbert_len_for(size_t window_size) { return std::max<size_t>(next_pow2(window_size * 2), 2); }  // :225-227

SpectrogramBank::SpectrogramBank(const omx_spectrogram_config& cfg, uint32_t n_streams) : n_streams_(n_streams) {
    cfg_ = cfg;
    normalize(cfg_);
    last_nonzero_.reserve(n_streams_);
    OMX_HIP(hipMemset(last_nonzero_.ptr, 0xFF, n_streams_ * sizeof(long long)));  // -1 = None
}

void SpectrogramBank::reset_audio() {  // :212-217
    ragged_ = false;  // every stream drops its pending audio: the common host-side positions describe the bank again
    ragged_pending_bound_ = 0;
    tail_ = head_;
    pending_skip_ = 0;
    clear_last_nonzero(last_stream_);
    reset_ = true;
}

void SpectrogramBank::clear_last_nonzero(hipStream_t stream) {
    OMX_HIP(hipMemsetAsync(last_nonzero_.ptr, 0xFF, n_streams_ * sizeof(long long), stream));
}

void SpectrogramBank::ensure_ring(uint64_t incoming, hipStream_t stream) {
    const uint64_t pending = head_ - tail_;
    const uint64_t need = pending + incoming;
    if (need <= ring_cap_) return;
    uint64_t cap = std::max<uint64_t>(next_pow2(need), 1024);
    DeviceBuffer<float> bigger;
    bigger.reserve((size_t)(cap * n_streams_));
    if (pending > 0 && ring_.ptr) {
        // re-home the pending samples: absolute positions are kept, only the modulus changes
        for (uint32_t s = 0; s < n_streams_; ++s) {
            uint64_t pos = tail_;
            while (pos < head_) {
                const uint64_t src_off = pos & (ring_cap_ - 1), dst_off = pos & (cap - 1);
                const uint64_t run = std::min({head_ - pos, ring_cap_ - src_off, cap - dst_off});
                OMX_HIP(hipMemcpyAsync(bigger.ptr + s * cap + dst_off, ring_.ptr + s * ring_cap_ + src_off,
                                       run * sizeof(float), hipMemcpyDeviceToDevice, stream));
                pos += run;
            }
        }
        OMX_HIP(hipStreamSynchronize(stream));
    }
    std::swap(ring_.ptr, bigger.ptr);
    std::swap(ring_.count, bigger.count);
    ring_cap_ = cap;
}

void SpectrogramBank::drain(uint64_t count) {  // :397-404 (audio_last_nonzero is an absolute position here)
    tail_ += std::min<uint64_t>(count, head_ - tail_);
}
void SpectrogramBank::advance(uint64_t count) {  // :406-410
    const uint64_t len = head_ - tail_;
    const uint64_t missing = count > len ? count - len : 0;
    drain(count);
    pending_skip_ += missing;
}

// Shapes the HIP path computes.  The reference accepts any fft_size (rustfft plans any length, :71-82 only normalises);
// here a shape outside the supported set is a backend failure (OMX_ERR_UNSUPPORTED), raised BEFORE any state changes.
static void require_supported(const omx_spectrogram_config& c) {
    const size_t W = (size_t)c.fft_size, zp = (size_t)c.zero_padding_factor;
    if (W > (size_t(1) << 24) || zp > (size_t(1) << 24) || W * zp > (size_t(1) << 24))
        unsupported("spectrogram padded FFT longer than 2^24");
    // lengths that are not powers of two run Bluestein's chirp-z on the generic kernel; their derivative window is evaluated on the
    // host as a plain O(W^2) DFT pair
    if (c.use_reassignment && !is_pow2(W) && W > (size_t(1) << 16)) unsupported("spectrogram window longer than 65536 that is not a power of two");
}

void SpectrogramBank::rebuild_fft(hipStream_t stream) {  // :229-279
    require_supported(cfg_);
    prepared_ = false;  // a failure below (HIP error) must not leave tables of two configurations mixed behind a prepared flag
    const size_t W = (size_t)cfg_.fft_size;
    fft_size_ = W * (size_t)cfg_.zero_padding_factor;
    hilbert_len_ = hilbert_len_for(W);
    const bool reassign = cfg_.use_reassignment != 0;
    const size_t active_len = reassign ? hilbert_len_ : fft_size_;

    const std::vector<float> window = window_coefficients(cfg_.window, W);
    std::vector<float> bin_norm = fft_bin_normalization(window, fft_size_);
    d_window_.upload(window, stream);
    d_tw_fft_.upload(twiddle_table(fft_size_, std::max<size_t>(fft_size_ / 2, 1)), stream);
    blu_m_ = 0;
    if (!is_pow2(fft_size_)) {  // any other transform length (the reference plans it with rustfft): Bluestein's chirp-z
        const BluesteinHostTables t = bluestein_tables(fft_size_);
        blu_m_ = t.m;
        d_blu_chirp_.upload(t.chirp, stream);
        d_blu_bf_.upload(t.bf, stream);
        d_blu_tw_.upload(t.tw_m, stream);
        OMX_HIP(hipStreamSynchronize(stream));  // host vectors above go out of scope
    }
    if (reassign) {
        const float inv_h = 1.0f / (float)hilbert_len_;  // :263-266
        for (float& n : bin_norm) n *= inv_h * inv_h;
        d_tw_hilbert_.upload(twiddle_table(hilbert_len_, hilbert_len_ / 2), stream);
        // derivative window (:569-599): FFT-based, computed on the device
        d_dwindow_.reserve(W);
        if (W <= 1) {
            OMX_HIP(hipMemsetAsync(d_dwindow_.ptr, 0, std::max<size_t>(W, 1) * sizeof(float), stream));
        } else if (!is_pow2(W)) {
            d_dwindow_.upload(derivative_window_host(window), stream);
            OMX_HIP(hipStreamSynchronize(stream));
        } else {
            DeviceBuffer<float> tw_w, scratch;
            tw_w.upload(twiddle_table(W, W / 2), stream);
            scratch.reserve(2 * W);
            launch_derivative_window(d_window_.ptr, (uint32_t)W, tw_w.ptr, scratch.ptr, d_dwindow_.ptr, stream);
            OMX_HIP(hipStreamSynchronize(stream));
        }
        std::vector<float> tw(W);  // :601-608
        const float center = (float)(W ? W - 1 : 0) * 0.5f;
        for (size_t i = 0; i < W; ++i) tw[i] = ((float)i - center) * window[i];
        d_twindow_.upload(tw, stream);
        double sum = 0.0, squares = 0.0;  // :111-117
        for (float xf : window) {
            const double x = (double)xf;
            sum = sum + x;
            squares = squares + x * x;
        }
        power_scale_ = (float)(sum * sum / ((double)fft_size_ * squares));
    } else {
        power_scale_ = 1.0f;
    }
    // fused kernels: W = F in {1024, 2048, 4096} (the tuned 4096 kernel / the size-templated ones; classic and reassigned)
    // (reassigned 16384 runs as three kernels through an HBM scratch: its fused form does not fit one CU)
    fast4096_ = (W == fft_size_ && (W == 16384 || W == 8192 || W == 4096 || W == 2048 || W == 1024));
    // classic columns: any window that is zero-padded to one of the fused transform sizes
    // (windows shorter than 256 samples stay on the generic kernel: DC removal over so few samples is all cancellation, and
    // the generic kernel keeps the reference's sequential mean)
    const bool fast_classic_zp = !reassign && W < fft_size_ && W >= 256 &&
                                 (fft_size_ == 1024 || fft_size_ == 2048 || fft_size_ == 4096 || fft_size_ == 8192 || fft_size_ == 16384);
    // zero-padded reassigned shapes with a fused kernel: window 1024 / 2048 / 4096 padded to 2048 / 4096 / 8192
    // ... or, through the three-kernel form, window 1024 ... 8192 padded to 16384
    fast_zp_ = reassign && W < fft_size_ &&
               (((W == 1024 || W == 2048 || W == 4096) && (fft_size_ == 2048 || fft_size_ == 4096 || fft_size_ == 8192)) ||
                ((W == 1024 || W == 2048 || W == 4096 || W == 8192) && fft_size_ == 16384));
    // ... or, for transforms beyond 16384 points, through zp W-point transforms of modulated slices (launch_stft_reassigned_residue)
    const bool residue_shape = W < fft_size_ && fft_size_ > 16384 && fft_size_ <= (size_t(1) << 19) && (fft_size_ & (fft_size_ - 1)) == 0 &&
                               (W == 1024 || W == 2048 || W == 4096 || W == 8192 || W == 16384);
    fast_zpr_ = reassign && residue_shape;
    classic_zpr_ = !reassign && residue_shape;
    if (fast_zpr_ || classic_zpr_) fast_zp_ = true;  // (same tables; the dispatch below tells them apart)
    if (fast_classic_zp) {
        fast4096_ = true;
        d_tw256_.upload(twiddle_table(256, 256), stream);
        d_tw4096_.upload(twiddle_table(fft_size_, fft_size_), stream);  // exp(-2 pi i k / F): the only transform of the classic path
    } else if (fast4096_ || fast_zp_) {
        d_tw256_.upload(twiddle_table(256, 256), stream);
        d_tw4096_.upload(twiddle_table(W, W), stream);      // exp(-2 pi i k / W)
        std::vector<float> half_tw = twiddle_table(2 * W, W);  // exp(-2 pi i k / 2W) / 2: the real-FFT split's 1/2 folded in (exact)
        for (float& x : half_tw) x *= 0.5f;
        d_tw8192_.upload(half_tw, stream);
        if (fast_zp_) d_twF_.upload(twiddle_table(fft_size_, fft_size_), stream);  // exp(-2 pi i k / F)
    }
    d_bin_norm_.upload(bin_norm, stream);
    OMX_HIP(hipStreamSynchronize(stream));  // host vectors above go out of scope
    prepared_ = true;

    const uint64_t buffered_len = (uint64_t)active_len * 2;  // :275-278 keep the newest 2*active_len samples
    const uint64_t pending = head_ - tail_;
    drain(pending > buffered_len ? pending - buffered_len : 0);
    pending_skip_ = 0;
}

void SpectrogramBank::prepare(hipStream_t stream) {
    if (!prepared_) rebuild_fft(stream);
}

void SpectrogramBank::update_config(const omx_spectrogram_config& in, hipStream_t stream) {  // :518-543
    omx_spectrogram_config cfg = in;
    normalize(cfg);
    const omx_spectrogram_config prev = cfg_;
    const bool prepared = prepared_;
    if (prepared) require_supported(cfg);  // rejected configurations leave the handle exactly as it was (old config, old tables)
    const uint64_t old_read_len = prev.use_reassignment ? hilbert_len_ : prev.fft_size;
    cfg_ = cfg;
    const bool rate_changed = prev.sample_rate != cfg.sample_rate;
    const bool rebuild = prev.fft_size != cfg.fft_size || prev.zero_padding_factor != cfg.zero_padding_factor ||
                         prev.window != cfg.window || prev.use_reassignment != cfg.use_reassignment || rate_changed;
    if (rebuild && prepared) {
        rebuild_fft(stream);
        if (rate_changed) {
            tail_ = head_;
            clear_last_nonzero(stream);
        }
    }
    const bool hop_changed = prev.hop_size != cfg.hop_size;
    if (hop_changed) pending_skip_ = 0;
    reset_ = reset_ || rebuild || hop_changed;
    if (ragged_ && (rebuild || hop_changed)) {
        // the same steps on every stream's own positions (they live on the device once process_ragged has run): a rebuild keeps
        // the newest 2 * active_len samples (:275-278) or, on a rate change, none; pending_skip goes (:534-536, rebuild_fft); the
        // next update of every stream carries `reset`
        const uint64_t active_len = cfg_.use_reassignment ? hilbert_len_ : fft_size_;  // (as rebuild_fft has just left them)
        const uint64_t keep = rate_changed ? 0 : active_len * 2;
        const bool trim = rebuild && prepared;
        launch_spectrogram_ragged_config(n_streams_, r_head_.ptr, r_tail_.ptr, r_skip_.ptr, r_reset_flag_.ptr, last_nonzero_.ptr, trim, keep,
                                         hop_changed || trim, rate_changed && trim, stream);
        OMX_HIP(hipGetLastError());
        // what a stream can still hold when the next call sizes its ring and column count
        const uint64_t before = std::max<uint64_t>(ragged_pending_bound_, old_read_len ? old_read_len - 1 : 0);
        ragged_pending_bound_ = trim ? std::min<uint64_t>(before, keep) : before;
    }
}

// The column kernels of one call: `n_cols` = columns per stream slot of the outputs; lock-step banks pass the common tail,
// ragged banks the per-stream tails / column counts written by spectrogram_plan_kernel.
void SpectrogramBank::launch_columns(uint64_t n_cols, uint64_t tail, const uint64_t* tails, const uint32_t* cols, hipStream_t stream) {
    const uint64_t W = cfg_.fft_size, hop = cfg_.hop_size;
    const bool reassign = cfg_.use_reassignment != 0;
    const uint64_t bin_count = fft_size_ / 2 + 1;
    const uint64_t center_offset = reassign ? (hilbert_len_ - W) / 2 : 0;
    const uint64_t stride = bin_count;
    const size_t out_bytes = (size_t)(n_streams_ * n_cols * stride) * (reassign ? sizeof(omx_spectrogram_point) : sizeof(uint16_t));
    const bool pinned = host_output_limit_ != 0 && out_bytes <= host_output_limit_;
    d_counts_.reserve((size_t)(n_streams_ * n_cols), pinned);
    if (cols) OMX_HIP(hipMemsetAsync(d_counts_.ptr, 0, (size_t)(n_streams_ * n_cols) * sizeof(uint32_t), stream));  // ragged: slots past a stream's own columns
    if (reassign) d_points_.reserve((size_t)(n_streams_ * n_cols * stride), pinned);
    else d_codes_.reserve((size_t)(n_streams_ * n_cols * stride), pinned);

    const float sr = cfg_.sample_rate;
    const float bin_hz = sr / (float)fft_size_;             // :446-450
    const float max_hz = sr * 0.5f;
    const float inv_2pi = sr / kTau;
    const float inv_hop = 1.0f / (float)hop;
    const float latency_hops = (float)center_offset * inv_hop;

    timer_.begin(stream);
    // the fused kernel indexes the ring with 32-bit offsets
    const bool fast = (fast4096_ || fast_zp_) && !force_generic_ && ring_cap_ <= (uint64_t(1) << 30) && hop <= 0xFFFFFFFFull;
    if (fast) {
        StftFastArgs fa{};
        fa.ring = ring_.ptr;
        fa.cap = ring_cap_;
        fa.tail = tail;
        fa.tails = tails;
        fa.cols = cols;
        fa.hop = (uint32_t)hop;
        fa.n_streams = n_streams_;
        fa.n_cols = (uint32_t)n_cols;
        fa.column_stride = (uint32_t)stride;
        fa.window_size = (uint32_t)W;
        fa.last_nonzero = last_nonzero_.ptr;
        fa.window = d_window_.ptr;
        fa.dwindow = d_dwindow_.ptr;
        fa.twindow = d_twindow_.ptr;
        fa.bin_norm = d_bin_norm_.ptr;
        fa.tw256 = reinterpret_cast<const v2f*>(d_tw256_.ptr);
        fa.tw4096 = reinterpret_cast<const v2f*>(d_tw4096_.ptr);
        fa.tw8192 = reinterpret_cast<const v2f*>(d_tw8192_.ptr);
        fa.bin_hz = bin_hz;
        fa.max_hz = max_hz;
        fa.inv_2pi = inv_2pi;
        fa.inv_hop = inv_hop;
        fa.latency_hops = latency_hops;
        // window.rs:20-43: Hann = [0.5, -0.5], Hamming = [25/46, -21/46] (the reference's f32 constants)
        fa.win_terms = (cfg_.window == OMX_WINDOW_HANN || cfg_.window == OMX_WINDOW_HAMMING) ? 2u : 0u;
        fa.win_c0 = cfg_.window == OMX_WINDOW_HANN ? 0.5f : 25.0f / 46.0f;
        fa.win_c1 = cfg_.window == OMX_WINDOW_HANN ? -0.5f : -21.0f / 46.0f;
        {   // WindowKind::coefficients (window.rs:24-31), the reference's f32 constants
            static const float kCos[5][4] = {{1.0f, 0.0f, 0.0f, 0.0f},
                                             {0.5f, -0.5f, 0.0f, 0.0f},
                                             {25.0f / 46.0f, -21.0f / 46.0f, 0.0f, 0.0f},
                                             {0.42f, -0.5f, 0.08f, 0.0f},
                                             {0.35875f, -0.48829f, 0.14128f, -0.01168f}};
            static const uint32_t kTerms[5] = {1, 2, 2, 3, 4};
            const uint32_t wk = cfg_.window <= OMX_WINDOW_BLACKMAN_HARRIS ? cfg_.window : OMX_WINDOW_HANN;
            for (int m = 0; m < 4; ++m) fa.cos_c[m] = kCos[wk][m];
            fa.cos_terms = kTerms[wk];
        }
        fa.points = d_points_.ptr;
        fa.counts = d_counts_.ptr;
        if (!reassign) {  // classic columns: per-column sequential window sums (window.rs:76-79), written ahead of the transform kernel
            d_col_sums_.reserve((size_t)(n_streams_ * n_cols));
            fa.col_sums = d_col_sums_.ptr;
        }
        const bool cross_check = kernel_form_ == 30;  // OMX_OPT_KERNEL_FORM: 4096 through the size-templated kernel
        const bool split_check = kernel_form_ == 31;  // ... or through the three-kernel form
        if (reassign && fft_size_ == 4096 && !fast_zp_ && split_check) {
            const uint64_t total = (uint64_t)n_streams_ * n_cols, chunk = std::min<uint64_t>(total, 4096);
            d_workspace_.reserve((size_t)(chunk * stft_big_scratch_bytes_per_frame() / sizeof(float)));
            for (uint64_t first = 0; first < total; first += chunk)
                launch_stft_reassigned_4096_split(fa, d_workspace_.ptr, (uint32_t)first, (uint32_t)std::min(chunk, total - first), stream);
        } else if (reassign && fft_size_ == 16384 && !fast_zp_) {
            const uint64_t total = (uint64_t)n_streams_ * n_cols, chunk = std::min<uint64_t>(total, 1024);
            d_workspace_.reserve((size_t)(chunk * stft_big_scratch_bytes_per_frame() / sizeof(float)));
            for (uint64_t first = 0; first < total; first += chunk)
                launch_stft_reassigned_16384(fa, d_workspace_.ptr, (uint32_t)first, (uint32_t)std::min(chunk, total - first), stream);
        } else if (fast_zpr_) {
            const uint64_t total = (uint64_t)n_streams_ * n_cols, per = stft_residue_scratch_bytes_per_frame((uint32_t)W, (uint32_t)fft_size_);
            const uint64_t chunk = std::min<uint64_t>(total, std::max<uint64_t>((uint64_t(512) << 20) / per, 1));  // <= 512 MB of scratch
            d_workspace_.reserve((size_t)(chunk * per / sizeof(float)));
            for (uint64_t first = 0; first < total; first += chunk)
                (void)launch_stft_reassigned_residue(fa, (uint32_t)W, (uint32_t)(fft_size_ / W), reinterpret_cast<const v2f*>(d_twF_.ptr), d_workspace_.ptr,
                                                     (uint32_t)first, (uint32_t)std::min(chunk, total - first), stream);
        } else if (classic_zpr_) {
            (void)launch_stft_classic_residue(fa, d_codes_.ptr, (uint32_t)W, (uint32_t)(fft_size_ / W), reinterpret_cast<const v2f*>(d_twF_.ptr), stream);
        } else if (fast_zp_ && fft_size_ == 16384) {
            const uint64_t total = (uint64_t)n_streams_ * n_cols, chunk = std::min<uint64_t>(total, 1024);
            d_workspace_.reserve((size_t)(chunk * stft_big_scratch_bytes_per_frame() / sizeof(float)));
            for (uint64_t first = 0; first < total; first += chunk)
                (void)launch_stft_reassigned_zp_16384(fa, (uint32_t)W, reinterpret_cast<const v2f*>(d_twF_.ptr), d_workspace_.ptr,
                                                      (uint32_t)first, (uint32_t)std::min(chunk, total - first), stream);
        } else if (fast_zp_)
            (void)launch_stft_reassigned_zp(fa, (uint32_t)W, (uint32_t)fft_size_, reinterpret_cast<const v2f*>(d_twF_.ptr), stream);
        else if (!reassign)
            launch_stft_classic_pow2(fa, d_codes_.ptr, (uint32_t)fft_size_, stream);
        else if (fft_size_ == 4096 && !cross_check)
            launch_stft_reassigned_4096(fa, kernel_form_, stream);
        else if (fft_size_ == 8192 && fa.win_terms == 2 && kernel_form_ == 0 && !cross_check)
            launch_stft_reassigned_8192(fa, stream);  // one dual 4096-point transform per 8192-point transform (stft8192_kernels.hip)
        else
            launch_stft_reassigned_pow2(fa, (uint32_t)fft_size_, stream);
    } else {
        if (hop > 0xFFFFFFFFull) unsupported("hop_size beyond 2^32");
        const uint64_t per_wg = (reassign ? hilbert_len_ + 3 * fft_size_ : fft_size_) + blu_m_;  // (+ the chirp-z scratch)
        const uint64_t total = (uint64_t)n_streams_ * n_cols;
        // working set in LDS when it fits one CU (<= 144 KiB), else a global workspace bounded to ~1 GiB
        const bool ws_in_lds = per_wg * sizeof(v2f) <= 144 * 1024;
        uint64_t wgs = std::min<uint64_t>(total, ws_in_lds ? 4096 : 1024);
        if (!ws_in_lds) {
            while (wgs > 1 && wgs * per_wg * sizeof(v2f) > (uint64_t(1) << 30)) wgs /= 2;
            d_workspace_.reserve((size_t)(wgs * per_wg * 2));
        }
        StftGenericArgs ga{};
        ga.ring = ring_.ptr;
        ga.cap = ring_cap_;
        ga.tail = tail;
        ga.tails = tails;
        ga.cols = cols;
        ga.hop = (uint32_t)hop;
        ga.n_streams = n_streams_;
        ga.n_cols = (uint32_t)n_cols;
        ga.column_stride = (uint32_t)stride;
        ga.last_nonzero = last_nonzero_.ptr;
        ga.reassign = reassign ? 1 : 0;
        ga.window_size = (uint32_t)W;
        ga.fft_size = (uint32_t)fft_size_;
        ga.hilbert_len = (uint32_t)hilbert_len_;
        ga.log_fft = log2_exact(fft_size_);
        ga.blu = BluesteinPlan{(uint32_t)blu_m_, blu_m_ ? log2_exact(blu_m_) : 0u, reinterpret_cast<const v2f*>(d_blu_chirp_.ptr),
                               reinterpret_cast<const v2f*>(d_blu_bf_.ptr), reinterpret_cast<const v2f*>(d_blu_tw_.ptr)};
        ga.log_hilbert = log2_exact(hilbert_len_);
        ga.window = d_window_.ptr;
        ga.dwindow = d_dwindow_.ptr;
        ga.twindow = d_twindow_.ptr;
        ga.bin_norm = d_bin_norm_.ptr;
        ga.tw_fft = reinterpret_cast<const v2f*>(d_tw_fft_.ptr);
        ga.tw_hilbert = reinterpret_cast<const v2f*>(d_tw_hilbert_.ptr);
        ga.bin_hz = bin_hz;
        ga.max_hz = max_hz;
        ga.inv_2pi = inv_2pi;
        ga.inv_hop = inv_hop;
        ga.latency_hops = latency_hops;
        ga.workspace = ws_in_lds ? nullptr : reinterpret_cast<v2f*>(d_workspace_.ptr);
        ga.workspace_stride = per_wg;
        ga.points = d_points_.ptr;
        ga.counts = d_counts_.ptr;
        ga.codes = d_codes_.ptr;
        launch_stft_generic(ga, (uint32_t)wgs, stream);
    }
    timer_.end(stream);
    OMX_HIP(hipGetLastError());

}

// process_block (:490-516) in three steps, so that a capture group can feed several banks from ONE ingest launch:
// push_begin (format change, prepare, the push_audio bookkeeping; says what to project where), the ingest launch, push_end, and
// process_pushed (process_ready_windows + the update).
int SpectrogramBank::push_begin(uint64_t frames, uint32_t channels, float sample_rate_in, hipStream_t stream, IngestSlots& slots) {
    slots = IngestSlots{};
    last_stream_ = stream;
    if (ragged_) {
        set_last_error("spectrogram bank is in ragged mode (per-stream positions): use process_ragged, or reset_audio() first");
        return OMX_ERR_INVALID;
    }
    if (frames == 0) return OMX_NONE;  // block.is_empty()
    const float sample_rate = sanitize_sample_rate(sample_rate_in);
    if (cfg_.sample_rate != sample_rate) {
        cfg_.sample_rate = sample_rate;
        rebuild_fft(stream);
        tail_ = head_;
        clear_last_nonzero(stream);
        reset_ = true;
    }
    prepare(stream);
    // ---- push_audio (:412-437)
    const uint64_t skip = std::min<uint64_t>(pending_skip_, frames);
    pending_skip_ -= skip;
    if (skip != frames) {
        const uint64_t count = frames - skip;
        ensure_ring(count, stream);
        slots.n = 1;
        slots.project[0] = channels == 1 ? OMX_PROJECT_RAW : OMX_CHANNEL_MID;  // :420-431
        slots.ring[0] = ring_.ptr;
        slots.cap[0] = ring_cap_;
        slots.head[0] = head_;
        slots.skip = skip;
        slots.count = count;
        slots.last_nonzero = last_nonzero_.ptr;
        partial_nonzero_.reserve((size_t)n_streams_ * ingest_partials_per_stream(count));
        slots.partial_nonzero = partial_nonzero_.ptr;
    }
    return OMX_PRODUCED;
}
void SpectrogramBank::push_end(const IngestSlots& slots) { head_ += slots.count; }

int SpectrogramBank::process(const float* pcm, bool pcm_on_device, uint64_t frames, uint32_t channels_in,
                             float sample_rate_in, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                             omx_spectrogram_bank_update* out) {  // :490-516
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    IngestSlots slots;
    const int rc = push_begin(frames, channels, sample_rate_in, stream, slots);
    if (rc != OMX_PRODUCED) return rc;
    if (slots.count) {
        const float* d_pcm = pcm;
        if (!pcm_on_device) {
            const size_t n = (size_t)n_streams_ * frames * channels;
            d_pcm = staging_.stage(pcm, n, stream);
        }
        const IngestSlots* one[1] = {&slots};
        launch_ingest_slots(d_pcm, frames, make_format(channels, positions), one, 1, n_streams_, stream);
        push_end(slots);
    }
    return process_pushed(stream, out);
}

int SpectrogramBank::process_pushed(hipStream_t stream, omx_spectrogram_bank_update* out) {
    // ---- process_ready_windows (:281-388)
    const uint64_t W = cfg_.fft_size, hop = cfg_.hop_size;
    const bool reassign = cfg_.use_reassignment != 0;
    const uint64_t bin_count = fft_size_ / 2 + 1;
    const uint64_t read_len = reassign ? hilbert_len_ : W;
    const uint64_t pending = head_ - tail_;
    const uint64_t ready = pending >= read_len ? (pending - read_len) / hop + 1 : 0;
    const uint32_t kind = reassign ? OMX_COLUMN_REASSIGNED : OMX_COLUMN_CLASSIC;
    const uint64_t retained = history_columns(kind, (uint32_t)bin_count, cfg_.history_length);
    const uint64_t skip_cols = ready > retained ? ready - retained : 0;
    advance(skip_cols * hop);
    const uint64_t n_cols = ready - skip_cols;
    if (n_cols == 0) return OMX_NONE;
    if (n_cols > 0xFFFFFFFFull / std::max<uint64_t>(n_streams_, 1)) unsupported("too many columns in one call");

    launch_columns(n_cols, tail_, nullptr, nullptr, stream);
    const uint64_t stride = bin_count;

    for (uint64_t c = 0; c < n_cols; ++c) advance(hop);  // :384 per column (missing samples -> pending_skip)

    last_cols_ = n_cols;
    last_stride_ = stride;
    last_kind_ = kind;
    if (out) {
        std::memset(out, 0, sizeof(*out));
        out->fft_size = fft_size_;
        out->hop_size = hop;
        out->history_length = cfg_.history_length;
        out->n_streams = n_streams_;
        out->n_columns = n_cols;
        out->column_stride = stride;
        out->d_counts = d_counts_.ptr;
        out->d_points = reassign ? d_points_.ptr : nullptr;
        out->d_codes = reassign ? nullptr : d_codes_.ptr;
        out->sample_rate = cfg_.sample_rate;
        out->reassigned_power_scale = power_scale_;
        out->reset = reset_ ? 1 : 0;
        out->kind = kind;
    }
    reset_ = false;  // std::mem::take (:511)
    return OMX_PRODUCED;
}

void SpectrogramBank::enter_ragged(hipStream_t stream) {
    // the lock-step state (common head / tail / pending_skip / reset) becomes every stream's own
    std::vector<uint64_t> h(n_streams_, head_), t(n_streams_, tail_), k(n_streams_, pending_skip_);
    std::vector<uint32_t> r(n_streams_, reset_ ? 1u : 0u);
    r_head_.upload(h, stream);
    r_tail_.upload(t, stream);
    r_skip_.upload(k, stream);
    r_reset_flag_.upload(r, stream);
    for (DeviceBuffer<uint64_t>* b : {&r_ing_head_, &r_col_tail_}) b->reserve(n_streams_);
    for (DeviceBuffer<uint32_t>* b : {&r_ing_skip_, &r_ing_count_, &r_ncols_, &r_reset_out_}) b->reserve(n_streams_);
    ragged_ = true;
}

int SpectrogramBank::process_ragged(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask,
                                    uint32_t channels_in, float sample_rate_in, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                    omx_spectrogram_ragged_update* out) {
    IngestArgs ia{};
    const int rc = ragged_plan(d_pcm, frames_capacity, frames, reset_mask, channels_in, sample_rate_in, positions, stream, ia);
    if (rc < 0) return rc;
    launch_ingest(ia, n_streams_, stream);
    OMX_HIP(hipGetLastError());
    return ragged_finish(stream, out);
}

// process_ragged in two halves, so that a capture group can feed this bank's ring and the Spectrum bank's from ONE projection launch
// (launch_ingest_ragged_parts): everything up to the plan kernel, which leaves the per-stream ingest parameters in `ia` ...
int SpectrogramBank::ragged_plan(const float* d_pcm, uint64_t frames_capacity, const uint32_t* frames, const uint8_t* reset_mask,
                                 uint32_t channels_in, float sample_rate_in, const uint8_t positions[OMX_MAX_CHANNELS], hipStream_t stream,
                                 IngestArgs& ia_out) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    last_stream_ = stream;
    if (frames_capacity == 0 || frames_capacity > 0xFFFFFFFFull) {
        set_last_error("spectrogram process_ragged: frames_capacity must be 1 .. 2^32 - 1");
        return OMX_ERR_INVALID;
    }
    for (uint32_t s = 0; s < n_streams_; ++s)
        if (frames[s] > frames_capacity) {
            set_last_error("spectrogram process_ragged: frames[s] > frames_capacity");
            return OMX_ERR_INVALID;
        }
    const float sample_rate = sanitize_sample_rate(sample_rate_in);
    if (cfg_.sample_rate != sample_rate) {  // a format change concerns every stream of the bank (:493-500)
        cfg_.sample_rate = sample_rate;
        rebuild_fft(stream);
        reset_audio();
    }
    prepare(stream);
    if (!ragged_) enter_ragged(stream);
    const uint64_t W = cfg_.fft_size, hop = cfg_.hop_size;
    const bool reassign = cfg_.use_reassignment != 0;
    const uint64_t bin_count = fft_size_ / 2 + 1;
    const uint64_t read_len = reassign ? hilbert_len_ : W;
    const uint32_t kind = reassign ? OMX_COLUMN_REASSIGNED : OMX_COLUMN_CLASSIC;
    const uint64_t retained = history_columns(kind, (uint32_t)bin_count, cfg_.history_length);
    // every stream enters a call with fewer than read_len pending samples (all of its ready windows were consumed), except right
    // after the switch from lock-step mode, where the common pending count is known
    const uint64_t pending_bound = std::max<uint64_t>(std::max<uint64_t>(head_ - tail_, ragged_pending_bound_), read_len ? read_len - 1 : 0);
    const uint64_t most = pending_bound + frames_capacity;
    const uint64_t max_cols = std::min<uint64_t>(most >= read_len ? (most - read_len) / hop + 1 : 0, retained);
    if (max_cols > 0xFFFFFFFFull / std::max<uint64_t>(n_streams_, 1)) unsupported("too many columns in one call");

    // ring: room for the pending samples + this call's; growth re-homes every stream's pending samples on the device
    if (most > ring_cap_ || !ring_.ptr) {
        const uint64_t cap = std::max<uint64_t>(next_pow2(most), 1024);
        DeviceBuffer<float> bigger;
        bigger.reserve((size_t)(cap * n_streams_));
        if (ring_.ptr) {
            launch_ring_rehome(ring_.ptr, ring_cap_, bigger.ptr, cap, r_head_.ptr, r_tail_.ptr, n_streams_, stream);
            OMX_HIP(hipStreamSynchronize(stream));
        }
        std::swap(ring_.ptr, bigger.ptr);
        std::swap(ring_.count, bigger.count);
        ring_cap_ = cap;
    }
    // the call's per-stream inputs (small: through double-buffered pinned memory, no stream synchronisation)
    r_staging_.upload(frames, reset_mask, n_streams_, r_frames_, r_mask_, stream);
    SpectrogramPlanArgs pa{};
    pa.n_streams = n_streams_;
    pa.read_len = read_len;
    pa.hop = hop;
    pa.retained = retained;
    pa.max_cols = (uint32_t)max_cols;
    pa.frames = r_frames_.ptr;
    pa.reset_mask = reset_mask ? r_mask_.ptr : nullptr;
    pa.head = r_head_.ptr;
    pa.tail = r_tail_.ptr;
    pa.pending_skip = r_skip_.ptr;
    pa.reset_flag = r_reset_flag_.ptr;
    pa.last_nonzero = last_nonzero_.ptr;
    pa.ing_skip = r_ing_skip_.ptr;
    pa.ing_count = r_ing_count_.ptr;
    pa.ing_head = r_ing_head_.ptr;
    pa.col_tail = r_col_tail_.ptr;
    pa.n_cols = r_ncols_.ptr;
    pa.reset_out = r_reset_out_.ptr;
    launch_spectrogram_plan(pa, stream);

    IngestArgs ia{};
    ia.pcm = d_pcm;
    ia.frames_total = frames_capacity;
    ia.count = frames_capacity;  // grid bound; the per-stream values follow
    ia.skips = r_ing_skip_.ptr;
    ia.counts = r_ing_count_.ptr;
    ia.heads = r_ing_head_.ptr;
    ia.fmt = make_format(channels, positions);
    ia.n_out = 1;
    ia.project[0] = channels == 1 ? OMX_PROJECT_RAW : OMX_CHANNEL_MID;  // :420-431
    ia.ring[0] = ring_.ptr;
    ia.cap = ring_cap_;
    ia.last_nonzero = last_nonzero_.ptr;
    partial_nonzero_.reserve((size_t)n_streams_ * ingest_partials_per_stream(frames_capacity));
    ia.partial_nonzero = partial_nonzero_.ptr;
    ia_out = ia;
    head_ = tail_ = 0;  // from here on only the bounds above use them (pending_bound = read_len - 1)
    {   // a stream that sat this call out still holds what update_config / the lock-step calls left it
        bool all_fed = true;
        for (uint32_t s = 0; s < n_streams_; ++s) all_fed = all_fed && frames[s] != 0;
        if (all_fed) ragged_pending_bound_ = 0;
        else ragged_pending_bound_ = pending_bound;
    }
    pend_max_cols_ = max_cols;
    return OMX_PRODUCED;
}

// ... and, once the samples are in the ring, the column kernels and the update
int SpectrogramBank::ragged_finish(hipStream_t stream, omx_spectrogram_ragged_update* out) {
    const uint64_t max_cols = pend_max_cols_, hop = cfg_.hop_size;
    const bool reassign = cfg_.use_reassignment != 0;
    const uint64_t bin_count = fft_size_ / 2 + 1;
    const uint32_t kind = reassign ? OMX_COLUMN_REASSIGNED : OMX_COLUMN_CLASSIC;
    if (max_cols > 0) {
        launch_columns(max_cols, 0, r_col_tail_.ptr, r_ncols_.ptr, stream);
        OMX_HIP(hipGetLastError());
    }
    last_cols_ = max_cols;
    last_stride_ = bin_count;
    last_kind_ = kind;
    if (out) {
        std::memset(out, 0, sizeof(*out));
        out->fft_size = fft_size_;
        out->hop_size = hop;
        out->history_length = cfg_.history_length;
        out->n_streams = n_streams_;
        out->max_columns = max_cols;
        out->column_stride = bin_count;
        out->d_n_columns = r_ncols_.ptr;
        out->d_reset = r_reset_out_.ptr;
        out->d_counts = max_cols ? d_counts_.ptr : nullptr;
        out->d_points = (max_cols && reassign) ? d_points_.ptr : nullptr;
        out->d_codes = (max_cols && !reassign) ? d_codes_.ptr : nullptr;
        out->sample_rate = cfg_.sample_rate;
        out->reassigned_power_scale = power_scale_;
        out->kind = kind;
    }
    return max_cols ? OMX_PRODUCED : OMX_NONE;
}

int SpectrogramBank::fetch_column(uint64_t stream_index, uint64_t column, void* dst, uint64_t cap, uint64_t* n_out,
                                  hipStream_t stream) {
    if (stream_index >= n_streams_ || column >= last_cols_) {
        set_last_error("fetch_column: index out of range");
        return OMX_ERR_INVALID;
    }
    const uint64_t slot = stream_index * last_cols_ + column;
    if (last_kind_ == OMX_COLUMN_REASSIGNED) {
        uint32_t n = 0;
        OMX_HIP(hipMemcpyAsync(&n, d_counts_.ptr + slot, sizeof(n), hipMemcpyDeviceToHost, stream));
        OMX_HIP(hipStreamSynchronize(stream));
        const uint64_t take = std::min<uint64_t>(n, cap);
        if (take) {
            OMX_HIP(hipMemcpyAsync(dst, d_points_.ptr + slot * last_stride_, take * sizeof(omx_spectrogram_point),
                                   hipMemcpyDeviceToHost, stream));
            OMX_HIP(hipStreamSynchronize(stream));
        }
        if (n_out) *n_out = n;
    } else {
        const uint64_t take = std::min<uint64_t>(last_stride_, cap);
        OMX_HIP(hipMemcpyAsync(dst, d_codes_.ptr + slot * last_stride_, take * sizeof(uint16_t), hipMemcpyDeviceToHost,
                               stream));
        OMX_HIP(hipStreamSynchronize(stream));
        if (n_out) *n_out = last_stride_;
    }
    return OMX_NONE;
}

// ------------------------------------------------------------------ single-stream handle (host in/out)
int SpectrogramSingle::process_block(const omx_block* block, omx_spectrogram_update* out) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(block->channels, 1), OMX_MAX_CHANNELS);
    if (block->n_samples < channels) return OMX_NONE;  // AudioBlock::is_empty (dsp.rs:259-261)
    const uint64_t frames = block->n_samples / channels;
    // The bank stages the block in pinned host memory that the ingest kernel reads directly (HostStage), and small results come
    // back through pinned memory too: one synchronisation per producing call instead of a blocking copy each way.
    omx_spectrogram_bank_update bu;
    const int rc = bank.process(block->samples, false, frames, channels, block->sample_rate, block->positions, nullptr, &bu);
    if (rc != OMX_PRODUCED) return rc;
    const uint64_t n_cols = bu.n_columns, stride = bu.column_stride;
    offsets.assign(1, 0);
    points.clear();
    codes.clear();
    const bool on_host = bank.outputs_on_host();
    if (on_host) OMX_HIP(hipStreamSynchronize(nullptr));  // the kernels wrote straight into host memory
    if (bu.kind == OMX_COLUMN_REASSIGNED) {
        std::vector<uint32_t> counts_copy;
        std::vector<omx_spectrogram_point> all_copy;
        const uint32_t* counts = bu.d_counts;
        const omx_spectrogram_point* all = bu.d_points;
        if (!on_host) {
            counts_copy.resize(n_cols);
            OMX_HIP(hipMemcpy(counts_copy.data(), bu.d_counts, n_cols * sizeof(uint32_t), hipMemcpyDeviceToHost));
            all_copy.resize(n_cols * stride);
            OMX_HIP(hipMemcpy(all_copy.data(), bu.d_points, all_copy.size() * sizeof(omx_spectrogram_point), hipMemcpyDeviceToHost));
            counts = counts_copy.data();
            all = all_copy.data();
        }
        for (uint64_t c = 0; c < n_cols; ++c) {
            points.insert(points.end(), all + c * stride, all + c * stride + counts[c]);
            offsets.push_back(points.size());
        }
    } else {
        codes.resize(n_cols * stride);
        if (on_host) std::memcpy(codes.data(), bu.d_codes, codes.size() * sizeof(uint16_t));
        else OMX_HIP(hipMemcpy(codes.data(), bu.d_codes, codes.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
        for (uint64_t c = 0; c < n_cols; ++c) offsets.push_back((c + 1) * stride);
    }
    std::memset(out, 0, sizeof(*out));
    out->fft_size = bu.fft_size;
    out->hop_size = bu.hop_size;
    out->history_length = bu.history_length;
    out->n_columns = n_cols;
    out->column_offsets = offsets.data();
    out->points = points.data();
    out->codes = codes.data();
    out->sample_rate = bu.sample_rate;
    out->reassigned_power_scale = bu.reassigned_power_scale;
    out->reset = bu.reset;
    out->kind = bu.kind;
    return OMX_PRODUCED;
}

}  // namespace omx
