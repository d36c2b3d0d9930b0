/* Plain C99 host of the device-side batchers (include/omx.h: omx_batcher_bank_*): MeterEngine's DspBatcher (src/meter.rs:27-80) for S
 * captures whose packets sit in device memory.  Every capture's chunks — lengths, order, samples — are compared with what the host
 * batcher of the same library (omx_batcher_push, the reference's structure for one capture) emits for the same packets.  The HIP
 * runtime's C entry points are declared by hand: a C host needs no HIP headers.  Exit code 0 = every call succeeded and every chunk
 * matched; prints the counts. */
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

enum { S = 5, CH = 2, MAX_PACKET = 2048, PUSHES = 30, MAX_CHUNKS = 16 };

typedef struct chunk_log {   /* what one capture's host batcher emitted in one push */
    int n;
    uint64_t len[MAX_CHUNKS];        /* samples */
    float data[MAX_CHUNKS][1024 * CH];
} chunk_log;

static void on_ingest(void* user, const float* samples, uint64_t n_samples, const omx_audio_format* format) {
    chunk_log* log = (chunk_log*)user;
    (void)format;
    if (log->n < MAX_CHUNKS && n_samples <= 1024 * CH) {
        log->len[log->n] = n_samples;
        memcpy(log->data[log->n], samples, sizeof(float) * n_samples);
    }
    log->n += 1;
}

int main(void) {
    if (!omx_device_available()) {
        printf("no device\n");
        return 0;
    }
    omx_audio_format fmt;
    memset(&fmt, 0, sizeof(fmt));
    fmt.generation = 1;
    fmt.sample_rate = 48000.0f;
    fmt.channels = CH;
    omx_positions_fallback(CH, fmt.positions);

    omx_batcher_bank* bank = NULL;
    CHECK(omx_batcher_bank_create(S, MAX_PACKET, &bank));
    omx_batcher* host[S];
    for (int s = 0; s < S; ++s) CHECK(omx_batcher_create(&host[s]));

    float* packets = (float*)malloc(sizeof(float) * S * MAX_PACKET * CH);
    float* round_host = (float*)malloc(sizeof(float) * S * 1024 * CH);
    chunk_log* logs = (chunk_log*)malloc(sizeof(chunk_log) * S);
    void* d_packets = NULL;
    if (!packets || !round_host || !logs || hipMalloc(&d_packets, sizeof(float) * S * MAX_PACKET * CH) != 0) return 2;

    uint32_t lcg = 12345u;
    long chunks = 0, samples = 0, mismatches = 0, multi = 0;
    for (int push = 0; push < PUSHES; ++push) {
        uint32_t lengths[S];
        for (int s = 0; s < S; ++s) {
            lcg = lcg * 1664525u + 1013904223u;
            const uint32_t kind = (lcg >> 28) & 7u;   /* quanta, odd sizes, a stall's worth, nothing */
            lengths[s] = kind == 0 ? 0u : (kind < 4 ? 256u : (kind < 6 ? 1u + (lcg >> 8) % 700u : 700u + (lcg >> 8) % (MAX_PACKET - 699u)));
        }
        for (size_t i = 0; i < (size_t)S * MAX_PACKET * CH; ++i) {
            lcg = lcg * 1664525u + 1013904223u;
            packets[i] = (float)(int32_t)lcg * (1.0f / 2147483648.0f);
        }
        if (hipMemcpy(d_packets, packets, sizeof(float) * S * MAX_PACKET * CH, 1) != 0) return 2;
        uint32_t rounds = 0;
        CHECK(omx_batcher_bank_push(bank, (const float*)d_packets, MAX_PACKET, lengths, NULL, &fmt, NULL, &rounds));
        for (int s = 0; s < S; ++s) {   /* the reference's structure: one DspBatcher per capture, samples through the host */
            logs[s].n = 0;
            (void)omx_batcher_push(host[s], packets + (size_t)s * MAX_PACKET * CH, (uint64_t)lengths[s] * CH, &fmt, on_ingest, &logs[s]);
            if (logs[s].n > MAX_CHUNKS) return 3;
            multi += logs[s].n > 1;
        }
        uint32_t most = 0;
        for (int s = 0; s < S; ++s) most = (uint32_t)logs[s].n > most ? (uint32_t)logs[s].n : most;
        if (most != rounds) {
            fprintf(stderr, "push %d: %u rounds, the host batchers emitted up to %u chunks\n", push, rounds, most);
            return 4;
        }
        for (uint32_t r = 0; r < rounds; ++r) {
            const float* d_pcm = NULL;
            uint64_t cap = 0;
            const uint32_t* frames = NULL;
            CHECK(omx_batcher_bank_round(bank, r, &d_pcm, &cap, &frames));
            if (cap != 1024 || hipDeviceSynchronize() != 0 || hipMemcpy(round_host, d_pcm, sizeof(float) * S * cap * CH, 2) != 0) return 2;
            for (int s = 0; s < S; ++s) {
                const uint64_t want = r < (uint32_t)logs[s].n ? logs[s].len[r] : 0;
                if ((uint64_t)frames[s] * CH != want) {
                    mismatches += 1;
                    continue;
                }
                if (want && memcmp(round_host + (size_t)s * cap * CH, logs[s].data[r], sizeof(float) * want) != 0) mismatches += 1;
                chunks += want != 0;
                samples += (long)want;
            }
        }
        for (int s = 0; s < S; ++s) {   /* the remainders */
            float a[256 * CH], b[256 * CH];
            const uint64_t na = omx_batcher_bank_pending(bank, (uint32_t)s, a, 256 * CH, NULL), nb = omx_batcher_pending(host[s], b, 256 * CH);
            if (na != nb || (na && memcmp(a, b, sizeof(float) * na) != 0)) mismatches += 1;
        }
    }
    printf("chunks %ld samples %ld multi %ld mismatches %ld\n", chunks, samples, multi, mismatches);
    for (int s = 0; s < S; ++s) omx_batcher_destroy(host[s]);
    omx_batcher_bank_destroy(bank);
    hipFree(d_packets);
    free(packets);
    free(round_host);
    free(logs);
    return mismatches == 0 ? 0 : 5;
}
