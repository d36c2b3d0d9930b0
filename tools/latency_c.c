/* Per-block latency of the single-stream handles measured from plain C (what a Rust / C host pays per 256-frame block at 48 kHz):
 * host PCM in, host snapshot out, one call per block.  Build + run on the GPU box:  bash tools/latency_c.sh */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "omx.h"

enum { BLOCK = 256, BLOCKS = 1200, WARM = 200, CH = 2 };
static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e6 + (double)t.tv_nsec * 1e-3;
}
static int cmp(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static void report(const char* name, double* lat, int n) {
    qsort(lat, (size_t)n, sizeof(double), cmp);
    printf("%-52s median %7.1f us   p99 %7.1f us\n", name, lat[n / 2], lat[(int)(n * 0.99)]);
}

int main(void) {
    static float pcm[BLOCKS][BLOCK * CH];
    unsigned x = 12345u;
    for (int b = 0; b < BLOCKS; ++b)
        for (int i = 0; i < BLOCK; ++i) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            const float l = 0.4f * sinf(6.2831853f * 440.0f * (float)(b * BLOCK + i) / 48000.0f) + 1e-3f * ((float)x / 4294967296.0f - 0.5f);
            pcm[b][2 * i] = l;
            pcm[b][2 * i + 1] = -0.7f * l;
        }
    static double lat[BLOCKS];
    omx_block blk;
    memset(&blk, 0, sizeof(blk));
    blk.n_samples = BLOCK * CH;
    blk.channels = CH;
    blk.sample_rate = 48000.0f;
    omx_positions_fallback(CH, blk.positions);
#define RUN(NAME, CREATE, CALL)                                   \
    do {                                                          \
        CREATE;                                                   \
        for (int b = 0; b < BLOCKS; ++b) {                        \
            blk.samples = pcm[b];                                 \
            const double t0 = now_us();                           \
            if ((CALL) < 0) { fprintf(stderr, "%s failed: %s\n", NAME, omx_last_error()); return 1; } \
            lat[b] = now_us() - t0;                               \
        }                                                         \
        report(NAME, lat + WARM, BLOCKS - WARM);                  \
    } while (0)

    { omx_spectrogram_config c; omx_spectrogram_config_default(&c); omx_spectrogram* h = NULL; omx_spectrogram_update u;
      RUN("spectrogram 2048/64 reassigned (reference default)", omx_spectrogram_create(&c, &h), omx_spectrogram_process_block(h, &blk, &u)); omx_spectrogram_destroy(h); }
    { omx_spectrogram_config c; omx_spectrogram_config_default(&c); c.fft_size = 4096; c.hop_size = 256; omx_spectrogram* h = NULL; omx_spectrogram_update u;
      RUN("spectrogram 4096/256 reassigned", omx_spectrogram_create(&c, &h), omx_spectrogram_process_block(h, &blk, &u)); omx_spectrogram_destroy(h); }
    { omx_spectrum_config c; omx_spectrum_config_default(&c); c.hop_size = 256; c.fft_size = 4096; omx_spectrum* h = NULL; omx_spectrum_snapshot u;
      RUN("spectrum 4096/256", omx_spectrum_create(&c, &h), omx_spectrum_process_block(h, &blk, &u)); omx_spectrum_destroy(h); }
    { omx_loudness_config c; omx_loudness_config_default(&c); omx_loudness* h = NULL; omx_loudness_snapshot u;
      RUN("loudness", omx_loudness_create(&c, &h), omx_loudness_process_block(h, &blk, &u)); omx_loudness_destroy(h); }
    { omx_stereometer_config c; omx_stereometer_config_default(&c); c.analyze_bands = 1; omx_stereometer* h = NULL; omx_stereometer_snapshot u;
      RUN("stereometer (bands)", omx_stereometer_create(&c, &h), omx_stereometer_process_block(h, &blk, &u)); omx_stereometer_destroy(h); }
    { omx_oscilloscope_config c; omx_oscilloscope_config_default(&c); omx_oscilloscope* h = NULL; omx_oscilloscope_snapshot u;
      RUN("oscilloscope", omx_oscilloscope_create(&c, &h), omx_oscilloscope_process_block(h, &blk, &u)); omx_oscilloscope_destroy(h); }
    { omx_waveform_config c; omx_waveform_config_default(&c); omx_waveform* h = NULL; omx_waveform_update u;
      RUN("waveform", omx_waveform_create(&c, &h), omx_waveform_process_block(h, &blk, &u)); omx_waveform_destroy(h); }
    return 0;
}
