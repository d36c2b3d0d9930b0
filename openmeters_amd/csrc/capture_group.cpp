// CaptureGroup — see capture_group.hpp / include/omx.h.  Reference: VisualManager::{ingest_samples, reset_audio}
// (src/visuals/registry.rs:360-365, :396-418); the block partition of the block-based visuals is the batcher's (src/meter.rs:16-25).
#include "capture_group.hpp"

namespace omx {

void capture_group_config_default(omx_capture_group_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->n_streams = 1;
    spectrogram_config_default(&c->spectrogram);
    spectrum_config_default(&c->spectrum);
    loudness_config_default(&c->loudness);
    stereometer_config_default(&c->stereometer);
    oscilloscope_config_default(&c->oscilloscope);
    waveform_config_default(&c->waveform);
}

CaptureGroup::CaptureGroup(const omx_capture_group_config& cfg) : cfg_(cfg) {
    const uint32_t S = cfg.n_streams;
    if (cfg.visuals & OMX_VISUAL_SPECTROGRAM) spectrogram_.reset(new SpectrogramBank(cfg.spectrogram, S));
    if (cfg.visuals & OMX_VISUAL_SPECTRUM) spectrum_.reset(new SpectrumBank(cfg.spectrum, S, cfg.spectrum_emit_all_hops != 0));
    if (cfg.visuals & OMX_VISUAL_LOUDNESS) loudness_.reset(new LoudnessBank(cfg.loudness, S));
    if (cfg.visuals & OMX_VISUAL_STEREOMETER) stereometer_.reset(new StereometerBank(cfg.stereometer, S));
    if (cfg.visuals & OMX_VISUAL_OSCILLOSCOPE) oscilloscope_.reset(new OscilloscopeBank(cfg.oscilloscope, S));
    if (cfg.visuals & OMX_VISUAL_WAVEFORM) waveform_.reset(new WaveformBank(cfg.waveform, S));
    for (int i = 0; i < 2; ++i) {
        OMX_HIP(hipStreamCreateWithFlags(&side_[i], hipStreamNonBlocking));
        OMX_HIP(hipEventCreateWithFlags(&join_[i], hipEventDisableTiming));
    }
    OMX_HIP(hipEventCreateWithFlags(&fork_, hipEventDisableTiming));
}

CaptureGroup::~CaptureGroup() {
    for (int i = 0; i < 2; ++i) {
        if (side_[i]) {
            (void)hipStreamSynchronize(side_[i]);
            (void)hipStreamDestroy(side_[i]);
        }
        if (join_[i]) (void)hipEventDestroy(join_[i]);
    }
    if (fork_) (void)hipEventDestroy(fork_);
}

void CaptureGroup::reset_audio() {  // registry.rs:360-365: every module's reset_audio
    if (spectrogram_) spectrogram_->reset_audio();
    if (spectrum_) spectrum_->reset_audio();
    if (loudness_) loudness_->reset_audio();
    if (stereometer_) stereometer_->reset_audio();
    if (oscilloscope_) oscilloscope_->reset_audio();
    if (waveform_) waveform_->reset_audio();
    holds_valid_ = false;  // LoudnessState::reset_audio: fresh PeakHolds (loudness/state.rs:153-160)
}

void CaptureGroup::set_timing(bool on) {
    if (spectrogram_) spectrogram_->timer().enabled = on;
}

int CaptureGroup::ingest(const float* d_pcm, uint64_t frames, uint32_t channels_in, float sample_rate, const uint8_t positions[OMX_MAX_CHANNELS],
                         hipStream_t stream, omx_capture_group_update* out) {
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    const uint32_t S = cfg_.n_streams;
    omx_capture_group_update up;
    std::memset(&up, 0, sizeof(up));
    if (frames == 0) {  // block.is_empty(): nothing happens in any visual
        if (out) *out = up;
        return OMX_NONE;
    }
    // how the block-based visuals see the call (meter.rs:16-25: the batcher hands out blocks of round(256 fs / 48000) frames)
    const float sr = sanitize_sample_rate(sample_rate);
    uint64_t block = cfg_.block_frames ? cfg_.block_frames : (uint64_t)std::max(1.0, std::round(256.0 * (double)sr / 48000.0));
    uint64_t n_blocks = frames / block;
    if (n_blocks == 0 || n_blocks * block != frames) {
        block = frames;
        n_blocks = 1;
    }
    up.n_blocks = n_blocks;
    up.block_frames = block;
    const bool stats = stats_ && spectrogram_ && loudness_ && stereometer_;
    if (stats) {
        rows_.reserve((size_t)S * OMX_STATS_COLUMNS);
        OMX_HIP(hipMemsetAsync(rows_.ptr, 0, (size_t)S * OMX_STATS_COLUMNS * sizeof(float), stream));  // ahead of the fork
    }
    OMX_HIP(hipEventRecord(fork_, stream));
    int worst = OMX_NONE;
    auto note = [&](int rc, uint32_t bit) {
        if (rc < 0) worst = worst < 0 ? worst : rc;
        else if (rc == OMX_PRODUCED) up.produced |= bit;
    };

    // ---- the caller's stream: the banks that keep pending audio, fed by one projection of the block
    {
        IngestSlots sg, sp;
        int rc_sg = OMX_NONE, rc_sp = OMX_NONE;
        if (spectrogram_) rc_sg = spectrogram_->push_begin(frames, channels, sample_rate, stream, sg);
        if (spectrum_) rc_sp = spectrum_->push_begin(frames, channels, sample_rate, stream, sp);
        if (rc_sg < 0) note(rc_sg, 0);
        if (rc_sp < 0) note(rc_sp, 0);
        const AudioFormatArgs fmt = make_format(channels, positions);
        const bool both = rc_sg == OMX_PRODUCED && rc_sp == OMX_PRODUCED && sg.count && sp.count;
        if (both && shared_ingest_ && sg.skip == sp.skip && sg.count == sp.count && sg.n + sp.n <= OMX_INGEST_MAX_OUT) {
            const IngestSlots* two[2] = {&sg, &sp};
            launch_ingest_slots(d_pcm, frames, fmt, two, 2, S, stream);
            up.ingest_launches += 1;
        } else {
            if (rc_sg == OMX_PRODUCED && sg.count) {
                const IngestSlots* one[1] = {&sg};
                launch_ingest_slots(d_pcm, frames, fmt, one, 1, S, stream);
                up.ingest_launches += 1;
            }
            if (rc_sp == OMX_PRODUCED && sp.count) {
                const IngestSlots* one[1] = {&sp};
                launch_ingest_slots(d_pcm, frames, fmt, one, 1, S, stream);
                up.ingest_launches += 1;
            }
        }
        if (rc_sg == OMX_PRODUCED) {
            spectrogram_->push_end(sg);
            note(spectrogram_->process_pushed(stream, &up.spectrogram), OMX_VISUAL_SPECTROGRAM);
        }
        if (rc_sp == OMX_PRODUCED) {
            spectrum_->push_end(sp);
            note(spectrum_->process_pushed(stream, &up.spectrum), OMX_VISUAL_SPECTRUM);
        }
    }

    // ---- side stream 0: loudness (+ its summary columns), waveform
    bool used[2] = {false, false};
    if (loudness_ || waveform_) {
        used[0] = true;
        OMX_HIP(hipStreamWaitEvent(side_[0], fork_, 0));
        if (loudness_) {
            const int rc = loudness_->process(d_pcm, true, block, n_blocks, channels, sample_rate, positions, side_[0], &up.d_loudness);
            note(rc, OMX_VISUAL_LOUDNESS);
            if (stats && rc == OMX_PRODUCED && up.d_loudness) {
                // K9: true-peak bars + their peak holds on the sample clock (loudness/state.rs:36-60, 178-217)
                if (!holds_valid_) {
                    holds_.reserve((size_t)S * 3);
                    launch_peak_holds_reset(holds_.ptr, (uint64_t)S * 3, clock_, side_[0]);
                    holds_valid_ = true;
                }
                meters_.reserve((size_t)S * n_blocks);
                const double dt = (double)block / (double)sr;
                launch_loudness_meters(up.d_loudness, S, n_blocks, OMX_METER_TRUE_PEAK, OMX_METER_LUFS_SHORT_TERM, clock_, dt, holds_.ptr,
                                       meters_.ptr, side_[0]);
                clock_ += (double)n_blocks * dt;
                launch_stats_loudness(up.d_loudness, meters_.ptr, S, n_blocks, channels, rows_.ptr, side_[0]);
            }
        }
        if (waveform_) note(waveform_->process(d_pcm, true, frames, channels, sample_rate, positions, side_[0], &up.waveform), OMX_VISUAL_WAVEFORM);
        OMX_HIP(hipGetLastError());
        OMX_HIP(hipEventRecord(join_[0], side_[0]));
    }
    // ---- side stream 1: stereometer (+ its summary columns), oscilloscope
    if (stereometer_ || oscilloscope_) {
        used[1] = true;
        OMX_HIP(hipStreamWaitEvent(side_[1], fork_, 0));
        if (stereometer_) {
            const int rc = stereometer_->process(d_pcm, true, block, n_blocks, channels, sample_rate, positions, side_[1], &up.stereometer);
            note(rc, OMX_VISUAL_STEREOMETER);
            if (stats && rc == OMX_PRODUCED && up.stereometer.d_correlations)
                launch_stats_stereometer(up.stereometer.d_correlations, S, n_blocks, rows_.ptr, side_[1]);
        }
        if (oscilloscope_) {
            const int rc = oscilloscope_->process(d_pcm, true, block, n_blocks, channels, sample_rate, positions, side_[1]);
            note(rc, OMX_VISUAL_OSCILLOSCOPE);
            if (rc == OMX_PRODUCED) {
                up.oscilloscope.n_streams = S;
                up.oscilloscope.n_blocks = n_blocks;
                up.oscilloscope.epoch = oscilloscope_->epoch();
                up.oscilloscope.sample_stride = kScopeTarget;
                up.oscilloscope.d_headers = reinterpret_cast<const omx_oscilloscope_block_header*>(oscilloscope_->d_headers());
                up.oscilloscope.d_samples = oscilloscope_->d_samples();
            }
        }
        OMX_HIP(hipGetLastError());
        OMX_HIP(hipEventRecord(join_[1], side_[1]));
    }
    // ---- join; the spectrogram's summary columns follow its kernel on the caller's stream
    for (int i = 0; i < 2; ++i)
        if (used[i]) OMX_HIP(hipStreamWaitEvent(stream, join_[i], 0));
    if (stats) {
        if ((up.produced & OMX_VISUAL_SPECTROGRAM) && up.spectrogram.d_counts)
            launch_stats_spectrogram(up.spectrogram.d_counts, S, up.spectrogram.n_columns, rows_.ptr, stream);
        OMX_HIP(hipGetLastError());
        up.d_stats_rows = rows_.ptr;
    }
    if (out) *out = up;
    if (worst < 0) return worst;
    return up.produced ? OMX_PRODUCED : OMX_NONE;
}

}  // namespace omx
