/* Plain C99 host of the capture group as VisualManager (include/omx.h: omx_capture_group_set_enabled / _update_config / _note_format /
 * _ingest_ragged; reference src/visuals/registry.rs:266-277, :343-365, :396-418).  Four captures, each with its own frame counts per
 * call (a batcher chunk of 0 ... 4 quanta, uneven, delivered WHOLE as one block: meter.rs:61-64, registry.rs:407-417), a visual toggled off and on again, a second visual created by set_enabled, the spectrogram's
 * hop changed mid-stream, one capture reset on its own and a format-generation change that resets all of them.  Every capture is
 * compared, call by call, with single-stream handles of the CPU ORACLE (libomx_oracle.so, prefix omxo_, declared by hand below:
 * the public header only declares the product) fed exactly the same sequence:
 *   spectrogram  column count and `reset` flag per call exact, point counts within 4 per column
 *   loudness     momentary / short-term LUFS and true peak of every chunk (= block) within 1e-4 dB
 *   stereometer  the four correlations of every chunk within 1e-6 (chunk calls run the reference's operation order)
 * Exit code 0 = every call succeeded and every comparison held; prints the largest differences. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "omx.h"

extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t size, int kind); /* 1 = host to device, 2 = device to host */
extern int hipDeviceSynchronize(void);

int omxo_spectrogram_create(const omx_spectrogram_config*, omx_spectrogram**);
void omxo_spectrogram_destroy(omx_spectrogram*);
int omxo_spectrogram_update_config(omx_spectrogram*, const omx_spectrogram_config*);
int omxo_spectrogram_reset_audio(omx_spectrogram*);
int omxo_spectrogram_process_block(omx_spectrogram*, const omx_block*, omx_spectrogram_update*);
int omxo_loudness_create(const omx_loudness_config*, omx_loudness**);
void omxo_loudness_destroy(omx_loudness*);
int omxo_loudness_reset_audio(omx_loudness*);
int omxo_loudness_process_block(omx_loudness*, const omx_block*, omx_loudness_snapshot*);
int omxo_stereometer_create(const omx_stereometer_config*, omx_stereometer**);
void omxo_stereometer_destroy(omx_stereometer*);
int omxo_stereometer_reset_audio(omx_stereometer*);
int omxo_stereometer_process_block(omx_stereometer*, const omx_block*, omx_stereometer_snapshot*);

#define CHECK(expr)                                                           \
    do {                                                                      \
        int rc_ = (expr);                                                     \
        if (rc_ < 0) {                                                        \
            fprintf(stderr, "%s -> %d (%s)\n", #expr, rc_, omx_last_error()); \
            return 1;                                                         \
        }                                                                     \
    } while (0)
#define EXPECT(cond)                                              \
    do {                                                          \
        if (!(cond)) {                                            \
            fprintf(stderr, "line %d: %s\n", __LINE__, #cond);    \
            return 4;                                             \
        }                                                         \
    } while (0)

enum { S = 4, BLOCK = 256, MAXB = 4, CAP = BLOCK * MAXB, CALLS = 14, CH = 2 };

static void fill_block(omx_block* blk, const float* samples, size_t frames, const uint8_t* positions) {
    memset(blk, 0, sizeof(*blk));
    blk->samples = samples;
    blk->n_samples = frames * CH;
    blk->channels = CH;
    blk->sample_rate = 48000.0f;
    memcpy(blk->positions, positions, OMX_MAX_CHANNELS);
}

int main(void) {
    if (!omx_device_available()) {
        printf("no device\n");
        return 0;
    }
    omx_capture_group_config cfg;
    omx_capture_group_config_default(&cfg);
    cfg.n_streams = S;
    cfg.visuals = OMX_VISUAL_SPECTROGRAM | OMX_VISUAL_LOUDNESS; /* the stereometer joins later through set_enabled */
    cfg.block_frames = 0;  /* the reference's partition: what a capture delivers in one call is one AudioBlock */
    cfg.spectrogram.fft_size = 1024;
    cfg.spectrogram.hop_size = 256;
    cfg.spectrogram.use_reassignment = 1;
    cfg.spectrogram.history_length = 8192;
    cfg.stereometer.analyze_bands = 1;
    omx_capture_group* g = NULL;
    CHECK(omx_capture_group_create(&cfg, &g));
    EXPECT(omx_capture_group_enabled(g) == (OMX_VISUAL_SPECTROGRAM | OMX_VISUAL_LOUDNESS));
    EXPECT(omx_capture_group_update_config(g, OMX_VISUAL_LOUDNESS, &cfg.loudness, NULL) == OMX_ERR_INVALID); /* no such thing in the reference */
    EXPECT(omx_capture_group_set_enabled(g, OMX_VISUAL_LOUDNESS | OMX_VISUAL_SPECTRUM, 1) == OMX_ERR_INVALID); /* one bit at a time */

    omx_spectrogram* sg[S];
    omx_loudness* ld[S];
    omx_stereometer* st[S];
    for (int s = 0; s < S; ++s) {
        CHECK(omxo_spectrogram_create(&cfg.spectrogram, &sg[s]));
        CHECK(omxo_loudness_create(&cfg.loudness, &ld[s]));
        CHECK(omxo_stereometer_create(&cfg.stereometer, &st[s]));
    }
    uint8_t positions[OMX_MAX_CHANNELS];
    omx_positions_fallback(CH, positions);

    float* pcm = (float*)calloc((size_t)S * CAP * CH, sizeof(float));
    omx_loudness_snapshot* snaps = (omx_loudness_snapshot*)malloc(sizeof(omx_loudness_snapshot) * S * MAXB);
    float* rho = (float*)malloc(sizeof(float) * S * MAXB * 4);
    uint32_t ncols[S], resets[S], counts[S * 16];
    void* d_pcm = NULL;
    if (hipMalloc(&d_pcm, sizeof(float) * S * CAP * CH) != 0) return 2;
    size_t at[S] = {0, 0, 0, 0};
    int stereo_on = 0;
    uint64_t generation = 7;
    double worst_lufs = 0.0, worst_rho = 0.0;
    unsigned long long columns = 0, blocks = 0, point_diff = 0, rho_checks = 0;
    for (int call = 0; call < CALLS; ++call) {
        /* ---- what the host does between two chunks */
        if (call == 2) {  /* a visual that was not in cfg.visuals: created (prepared) by set_enabled */
            CHECK(omx_capture_group_set_enabled(g, OMX_VISUAL_STEREOMETER, 1));
            stereo_on = 1;
        }
        if (call == 5) {  /* off: skipped by ingest, keeps its state */
            CHECK(omx_capture_group_set_enabled(g, OMX_VISUAL_STEREOMETER, 0));
            stereo_on = 0;
        }
        if (call == 8) {
            CHECK(omx_capture_group_set_enabled(g, OMX_VISUAL_STEREOMETER, 1));
            stereo_on = 1;
        }
        if (call == 6) {  /* the spectrogram's hop, mid-stream (:518-543): pending samples stay, `reset` goes into the next update */
            cfg.spectrogram.hop_size = 128;
            CHECK(omx_capture_group_update_config(g, OMX_VISUAL_SPECTROGRAM, &cfg.spectrogram, NULL));
            for (int s = 0; s < S; ++s) CHECK(omxo_spectrogram_update_config(sg[s], &cfg.spectrogram));
        }
        uint8_t mask[S] = {0, 0, 0, 0};
        if (call == 9) mask[1] = 1;  /* one capture reset on its own */
        if (call == 11) generation = 8;  /* a format change of the capture source */
        const int was_reset = omx_capture_group_note_format(g, generation);
        CHECK(was_reset);
        EXPECT(was_reset == (call == 11));
        /* (note_format resets every visual and returns the group to lock-step positions; the next ragged call moves it back) */
        uint32_t frames[S];
        for (int s = 0; s < S; ++s) {
            frames[s] = (uint32_t)(BLOCK * ((call * 3 + s * 5 + (call * s) % 3) % (MAXB + 1)));
            if (call == 0) frames[s] = CAP;  /* everyone starts with a full chunk */
            for (uint32_t f = 0; f < frames[s]; ++f) {
                const double t = (double)(at[s] + f) / 48000.0;
                const float v = (float)(0.5 * sin(2.0 * 3.14159265358979323846 * (330.0 + 170.0 * s) * t) + 0.05 * sin(2.0 * 3.14159265358979323846 * 3100.0 * t));
                pcm[((size_t)s * CAP + f) * CH] = v;
                pcm[((size_t)s * CAP + f) * CH + 1] = (s % 2 ? 0.7f : -0.6f) * v;
            }
        }
        if (hipMemcpy(d_pcm, pcm, sizeof(float) * S * CAP * CH, 1) != 0) return 2;
        omx_capture_group_ragged_update up;
        CHECK(omx_capture_group_ingest_ragged(g, (const float*)d_pcm, CAP, frames, mask, CH, 48000.0f, positions, NULL, &up));
        if (hipDeviceSynchronize() != 0) return 2;
        EXPECT(up.block_frames == 0 && up.max_blocks == 1);
        EXPECT(omx_capture_group_ingest(g, (const float*)d_pcm, CAP, CH, 48000.0f, positions, NULL, NULL) == OMX_ERR_INVALID);
        if (hipMemcpy(ncols, up.spectrogram.d_n_columns, sizeof(ncols), 2) != 0) return 2;
        if (hipMemcpy(resets, up.spectrogram.d_reset, sizeof(resets), 2) != 0) return 2;
        EXPECT(up.spectrogram.max_columns <= 16);
        if (up.spectrogram.max_columns && hipMemcpy(counts, up.spectrogram.d_counts, sizeof(uint32_t) * S * up.spectrogram.max_columns, 2) != 0) return 2;
        if (hipMemcpy(snaps, up.loudness.d_snapshots, sizeof(omx_loudness_snapshot) * S, 2) != 0) return 2;
        if (stereo_on && hipMemcpy(rho, up.stereometer.d_correlations, sizeof(float) * S * 4, 2) != 0) return 2;
        EXPECT(((up.produced & OMX_VISUAL_STEREOMETER) != 0) <= stereo_on);
        /* ---- the same sequence through one oracle handle per capture and visual */
        for (int s = 0; s < S; ++s) {
            if (was_reset == 1 || mask[s]) {  /* VisualManager::reset_audio: every module, enabled or not (:360-365) */
                CHECK(omxo_spectrogram_reset_audio(sg[s]));
                CHECK(omxo_loudness_reset_audio(ld[s]));
                CHECK(omxo_stereometer_reset_audio(st[s]));
            }
            if (frames[s] == 0) {
                EXPECT(ncols[s] == 0);
                continue;
            }
            const float* mine = pcm + (size_t)s * CAP * CH;
            omx_block blk;
            omx_spectrogram_update su;
            memset(&su, 0, sizeof(su));
            fill_block(&blk, mine, frames[s], positions);
            const int rc = omxo_spectrogram_process_block(sg[s], &blk, &su);
            CHECK(rc);
            const uint32_t want_cols = rc > 0 ? (uint32_t)su.n_columns : 0u;
            EXPECT(ncols[s] == want_cols);
            if (want_cols) EXPECT((resets[s] != 0) == (su.reset != 0));
            for (uint32_t c = 0; c < want_cols; ++c) {
                const long long a = (long long)counts[(size_t)s * up.spectrogram.max_columns + c];
                const long long b = (long long)(su.column_offsets[c + 1] - su.column_offsets[c]);
                point_diff += (unsigned long long)llabs(a - b);
                EXPECT(llabs(a - b) <= 4);
            }
            columns += want_cols;
            {   /* the chunk, whole, to the block-based visuals: one process_block each */
                omx_loudness_snapshot ls;
                omx_stereometer_snapshot ss;
                memset(&ls, 0, sizeof(ls));
                memset(&ss, 0, sizeof(ss));
                fill_block(&blk, mine, frames[s], positions);
                CHECK(omxo_loudness_process_block(ld[s], &blk, &ls));
                const omx_loudness_snapshot* got = snaps + (size_t)s;
                double d = fabs((double)got->momentary_loudness - (double)ls.momentary_loudness);
                if (d > worst_lufs) worst_lufs = d;
                d = fabs((double)got->short_term_loudness - (double)ls.short_term_loudness);
                if (d > worst_lufs) worst_lufs = d;
                for (int k = 0; k < CH; ++k) {  /* ONE true-peak take over the whole chunk (loudness/processor.rs:301) */
                    d = fabs((double)got->true_peak_db[k] - (double)ls.true_peak_db[k]);
                    if (d > worst_lufs) worst_lufs = d;
                }
                ++blocks;
                if (stereo_on) {  /* a disabled visual is not fed: its oracle twin sits the call out as well */
                    const int produced = omxo_stereometer_process_block(st[s], &blk, &ss);
                    CHECK(produced);
                    if (produced > 0)
                        for (int k = 0; k < 4; ++k) {
                            d = fabs((double)rho[(size_t)s * 4 + k] - (double)ss.correlations[k]);
                            if (d > worst_rho) worst_rho = d;
                            ++rho_checks;
                        }
                }
            }
            at[s] += frames[s];
        }
    }
    printf("columns %llu blocks %llu point_diff %llu rho_checks %llu worst_lufs %.3g worst_rho %.3g\n", columns, blocks, point_diff, rho_checks,
           worst_lufs, worst_rho);
    for (int s = 0; s < S; ++s) {
        omxo_spectrogram_destroy(sg[s]);
        omxo_loudness_destroy(ld[s]);
        omxo_stereometer_destroy(st[s]);
    }
    omx_capture_group_destroy(g);
    hipFree(d_pcm);
    free(pcm);
    free(snaps);
    free(rho);
    return worst_lufs < 1e-4 && worst_rho < 1e-6 ? 0 : 5;
}
