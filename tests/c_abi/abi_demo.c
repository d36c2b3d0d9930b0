/* Plain C99 consumer of include/omx.h: what a non-Python host (the reference's Rust FFI, a C service) does.
 * Links libomx_hip.so (the product) or libomx_oracle.so with -DOMX_PREFIX_ORACLE (test infrastructure) — same calls.
 * Feeds a 1 kHz tone in 256-frame blocks to a spectrogram, a spectrum and a loudness processor and prints a few numbers;
 * tests/test_c_abi.py compares the two outputs.  Exit code 0 = every call returned a non-negative status. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "omx.h"

#ifdef OMX_PREFIX_ORACLE
#define F(name) omxo_##name
/* the oracle mirrors the ABI with an omxo_ prefix (declared here: the public header only declares the product) */
void omxo_spectrogram_config_default(omx_spectrogram_config*);
int omxo_spectrogram_create(const omx_spectrogram_config*, omx_spectrogram**);
void omxo_spectrogram_destroy(omx_spectrogram*);
int omxo_spectrogram_process_block(omx_spectrogram*, const omx_block*, omx_spectrogram_update*);
void omxo_spectrum_config_default(omx_spectrum_config*);
int omxo_spectrum_create(const omx_spectrum_config*, omx_spectrum**);
void omxo_spectrum_destroy(omx_spectrum*);
int omxo_spectrum_process_block(omx_spectrum*, const omx_block*, omx_spectrum_snapshot*);
void omxo_loudness_config_default(omx_loudness_config*);
int omxo_loudness_create(const omx_loudness_config*, omx_loudness**);
void omxo_loudness_destroy(omx_loudness*);
int omxo_loudness_process_block(omx_loudness*, const omx_block*, omx_loudness_snapshot*);
void omxo_positions_fallback(uint32_t, uint8_t*);
#else
#define F(name) omx_##name
#endif

#define CHECK(expr)                                                      \
    do {                                                                 \
        int rc_ = (expr);                                                \
        if (rc_ < 0) {                                                   \
            fprintf(stderr, "%s -> %d\n", #expr, rc_);                   \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(void) {
    enum { BLOCK = 256, BLOCKS = 200, CH = 2 };
    omx_spectrogram_config sg_cfg;
    omx_spectrum_config sp_cfg;
    omx_loudness_config ld_cfg;
    F(spectrogram_config_default)(&sg_cfg);
    F(spectrum_config_default)(&sp_cfg);
    F(loudness_config_default)(&ld_cfg);
    sg_cfg.fft_size = 4096;
    sg_cfg.hop_size = 256;
    sg_cfg.use_reassignment = 1;
    sp_cfg.fft_size = 4096;
    sp_cfg.hop_size = 1024;
    omx_spectrogram* sg = NULL;
    omx_spectrum* sp = NULL;
    omx_loudness* ld = NULL;
    CHECK(F(spectrogram_create)(&sg_cfg, &sg));
    CHECK(F(spectrum_create)(&sp_cfg, &sp));
    CHECK(F(loudness_create)(&ld_cfg, &ld));

    float* pcm = (float*)malloc(sizeof(float) * BLOCK * CH);
    omx_block block;
    block.samples = pcm;
    block.n_samples = BLOCK * CH;
    block.channels = CH;
    block.sample_rate = 48000.0f;
    F(positions_fallback)(CH, block.positions);

    unsigned long long columns = 0, points = 0;
    double peak_power = 0.0, peak_freq = 0.0;
    float spectrum_peak_db = -1000.0f, spectrum_peak_hz = 0.0f, momentary = 0.0f, true_peak = 0.0f;
    for (int b = 0; b < BLOCKS; ++b) {
        for (int i = 0; i < BLOCK; ++i) {
            const double t = (double)(b * BLOCK + i) / 48000.0;
            const float x = (float)(0.5 * sin(2.0 * 3.14159265358979323846 * 1000.0 * t));
            pcm[2 * i] = x;
            pcm[2 * i + 1] = 0.8f * x;
        }
        omx_spectrogram_update up;
        int rc = F(spectrogram_process_block)(sg, &block, &up);
        CHECK(rc);
        if (rc == OMX_PRODUCED) {
            columns += up.n_columns;
            for (uint64_t c = 0; c < up.n_columns; ++c)
                for (uint64_t k = up.column_offsets[c]; k < up.column_offsets[c + 1]; ++k) {
                    ++points;
                    if (up.points[k].power > peak_power) {
                        peak_power = up.points[k].power;
                        peak_freq = up.points[k].freq_hz;
                    }
                }
        }
        omx_spectrum_snapshot snap;
        rc = F(spectrum_process_block)(sp, &block, &snap);
        CHECK(rc);
        if (rc == OMX_PRODUCED && b == BLOCKS - 1)
            for (uint64_t k = 0; k < snap.bins; ++k)
                if (snap.traces[0][1][k] > spectrum_peak_db) {
                    spectrum_peak_db = snap.traces[0][1][k];
                    spectrum_peak_hz = snap.frequency_bins[k];
                }
        omx_loudness_snapshot ls;
        rc = F(loudness_process_block)(ld, &block, &ls);
        CHECK(rc);
        if (rc == OMX_PRODUCED) {
            momentary = ls.momentary_loudness;
            true_peak = ls.true_peak_db[0];
        }
    }
    printf("columns %llu points %llu peak_freq %.3f peak_power %.6e spectrum_peak_hz %.3f spectrum_peak_db %.3f momentary %.4f true_peak %.4f\n",
           columns, points, peak_freq, peak_power, (double)spectrum_peak_hz, (double)spectrum_peak_db, (double)momentary, (double)true_peak);
    F(spectrogram_destroy)(sg);
    F(spectrum_destroy)(sp);
    F(loudness_destroy)(ld);
    free(pcm);
    return 0;
}
