/* Plain C99 host of the capture group (include/omx.h: omx_capture_group_*): what the reference's VisualManager::ingest_samples
 * (src/visuals/registry.rs:396-418) becomes for a Rust / C host — one ingest call per block feeds every enabled visual of S captures.
 * Drives a 3-visual group (Spectrogram + Loudness + Stereometer, 4 streams, summary rows on) with device-resident PCM and checks the
 * summary rows against the single-stream handles of the same library fed the same blocks.  The HIP runtime's C entry points are
 * declared by hand: a C host needs no HIP headers.  Exit code 0 = every call succeeded; prints the largest differences. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "omx.h"

extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t size, int kind); /* 1 = host to device, 2 = device to host */
extern int hipDeviceSynchronize(void);

#define CHECK(expr)                                    \
    do {                                               \
        int rc_ = (expr);                              \
        if (rc_ < 0) {                                 \
            fprintf(stderr, "%s -> %d (%s)\n", #expr, rc_, omx_last_error()); \
            return 1;                                  \
        }                                              \
    } while (0)

int main(void) {
    enum { S = 4, BLOCK = 256, BLOCKS_PER_CALL = 8, CALLS = 9, CH = 2 };
    const size_t frames = (size_t)BLOCK * BLOCKS_PER_CALL;
    if (!omx_device_available()) {
        printf("no device\n");
        return 0;
    }
    omx_capture_group_config cfg;
    omx_capture_group_config_default(&cfg);
    cfg.n_streams = S;
    cfg.visuals = OMX_VISUAL_SPECTROGRAM | OMX_VISUAL_LOUDNESS | OMX_VISUAL_STEREOMETER;
    cfg.block_frames = BLOCK;
    cfg.spectrogram.fft_size = 1024;
    cfg.spectrogram.hop_size = 256;
    cfg.spectrogram.use_reassignment = 1;
    cfg.spectrogram.history_length = 8192; /* (the view pushes its width every ingest, registry.rs:181-188; 0 keeps one column) */
    cfg.stereometer.analyze_bands = 1;
    omx_capture_group* g = NULL;
    CHECK(omx_capture_group_create(&cfg, &g));
    CHECK(omx_capture_group_set_option(g, OMX_OPT_GROUP_STATS, 1));

    omx_loudness* ld[S];
    omx_stereometer* st[S];
    for (int s = 0; s < S; ++s) {
        CHECK(omx_loudness_create(&cfg.loudness, &ld[s]));
        CHECK(omx_stereometer_create(&cfg.stereometer, &st[s]));
    }
    uint8_t positions[OMX_MAX_CHANNELS];
    omx_positions_fallback(CH, positions);

    float* pcm = (float*)malloc(sizeof(float) * S * frames * CH);
    float* rows = (float*)malloc(sizeof(float) * S * OMX_STATS_COLUMNS);
    void* d_pcm = NULL;
    if (hipMalloc(&d_pcm, sizeof(float) * S * frames * CH) != 0) return 2;
    double worst_lufs = 0.0, worst_rho = 0.0;
    unsigned long long columns = 0;
    for (int call = 0; call < CALLS; ++call) {
        for (int s = 0; s < S; ++s)
            for (size_t f = 0; f < frames; ++f) {
                const double t = (double)(call * frames + f) / 48000.0;
                const float v = (float)(0.5 * sin(2.0 * 3.14159265358979323846 * (440.0 + 110.0 * s) * t));
                pcm[(s * frames + f) * CH] = v;
                pcm[(s * frames + f) * CH + 1] = -0.6f * v;
            }
        if (hipMemcpy(d_pcm, pcm, sizeof(float) * S * frames * CH, 1) != 0) return 2;
        omx_capture_group_update up;
        CHECK(omx_capture_group_ingest(g, (const float*)d_pcm, frames, CH, 48000.0f, positions, NULL, &up));
        if (hipDeviceSynchronize() != 0) return 2;
        if (up.ingest_launches != 1 || up.n_blocks != BLOCKS_PER_CALL || !up.d_stats_rows) return 3;
        if (hipMemcpy(rows, up.d_stats_rows, sizeof(float) * S * OMX_STATS_COLUMNS, 2) != 0) return 2;
        if (up.produced & OMX_VISUAL_SPECTROGRAM) columns += up.spectrogram.n_columns;
        for (int s = 0; s < S; ++s) {
            omx_loudness_snapshot ls;
            omx_stereometer_snapshot ss;
            memset(&ls, 0, sizeof(ls));
            memset(&ss, 0, sizeof(ss));
            int produced = 0;
            for (int b = 0; b < BLOCKS_PER_CALL; ++b) {
                omx_block blk;
                memset(&blk, 0, sizeof(blk));
                blk.samples = pcm + (s * frames + (size_t)b * BLOCK) * CH;
                blk.n_samples = (size_t)BLOCK * CH;
                blk.channels = CH;
                blk.sample_rate = 48000.0f;
                memcpy(blk.positions, positions, sizeof(positions));
                CHECK(omx_loudness_process_block(ld[s], &blk, &ls));
                int rc = omx_stereometer_process_block(st[s], &blk, &ss);
                CHECK(rc);
                produced = rc;
            }
            const float* r = rows + s * OMX_STATS_COLUMNS;
            double d = fabs((double)r[0] - (double)ls.momentary_loudness);
            if (d > worst_lufs) worst_lufs = d;
            d = fabs((double)r[1] - (double)ls.short_term_loudness);
            if (d > worst_lufs) worst_lufs = d;
            if (produced > 0)
                for (int k = 0; k < 4; ++k) {
                    d = fabs((double)r[3 + k] - (double)ss.correlations[k]);
                    if (d > worst_rho) worst_rho = d;
                }
        }
    }
    printf("columns %llu worst_lufs %.3g worst_rho %.3g rho_full %.4f\n", columns, worst_lufs, worst_rho, (double)rows[3]);
    for (int s = 0; s < S; ++s) {
        omx_loudness_destroy(ld[s]);
        omx_stereometer_destroy(st[s]);
    }
    omx_capture_group_destroy(g);
    hipFree(d_pcm);
    free(pcm);
    free(rows);
    return 0;
}
