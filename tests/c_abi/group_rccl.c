/* A native N-rank host of the capture group: plain C99 + librccl, one process per GPU, no Python and no torch.
 *   north_star: "independent capture streams shard embarrassingly across the 8 GPUs with RCCL over xGMI used only to gather summary stats".
 * Every rank owns a contiguous shard of `total` captures (the static partition of openmeters_amd/sharding.py restated below), runs the
 * full per-capture pipeline on it through ONE omx_capture_group_ingest per step (Spectrogram + Loudness + Stereometer, summary rows on),
 * and all-gathers the rows [shard][OMX_STATS_COLUMNS] with ncclAllGather on its own communicator — the library does not link RCCL
 * (include/omx.h, capture group).  The gathered table's checksum is printed by every rank; streams are independent and the rows are a
 * deterministic function of a capture's own samples, so an N-rank run and a 1-rank run of the same `total` print the same checksum.
 *
 *   usage: RANK=r WORLD_SIZE=n LOCAL_RANK=l OMX_RCCL_ID_FILE=/path/shared/by/the/ranks  group_rccl [total captures] [steps]
 *          group_rccl --shards <total> <world>      (no device: prints the partition, one "rank first count" line per rank)
 * The ncclUniqueId travels through a file: rank 0 writes it (tmp + rename), the others wait for it.  HIP and RCCL entry points are
 * declared by hand: a C host needs neither header set. */
#define _POSIX_C_SOURCE 200809L /* nanosleep */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "omx.h"

extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t size, int kind); /* 1 = host to device, 2 = device to host, 3 = device to device */
extern int hipMemset(void* dst, int value, size_t size);
extern int hipDeviceSynchronize(void);
extern int hipGetDeviceCount(int* count);

typedef struct {
    char internal[128];
} ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
extern int ncclGetUniqueId(ncclUniqueId* id);
extern int ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank);
extern int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, ncclComm_t comm, void* stream);
extern int ncclCommDestroy(ncclComm_t comm);
extern const char* ncclGetErrorString(int result);
enum { NCCL_FLOAT32 = 7 };

#define CHECK(expr)                                                                       \
    do {                                                                                  \
        int rc_ = (expr);                                                                 \
        if (rc_ < 0) {                                                                    \
            fprintf(stderr, "rank %d: %s -> %d (%s)\n", rank, #expr, rc_, omx_last_error()); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)
#define HIPCHECK(expr)                                                     \
    do {                                                                   \
        int rc_ = (expr);                                                  \
        if (rc_ != 0) {                                                    \
            fprintf(stderr, "rank %d: %s -> hip error %d\n", rank, #expr, rc_); \
            return 2;                                                      \
        }                                                                  \
    } while (0)
#define NCCLCHECK(expr)                                                                          \
    do {                                                                                         \
        int rc_ = (expr);                                                                        \
        if (rc_ != 0) {                                                                          \
            fprintf(stderr, "rank %d: %s -> %d (%s)\n", rank, #expr, rc_, ncclGetErrorString(rc_)); \
            return 3;                                                                            \
        }                                                                                        \
    } while (0)

/* openmeters_amd/sharding.py::shard_streams: capture s lives on rank s / ceil(total / world) */
static void shard_streams(unsigned total, unsigned rank, unsigned world, unsigned* first, unsigned* count) {
    const unsigned per = (total + world - 1) / world;
    *first = rank * per < total ? rank * per : total;
    *count = total - *first < per ? total - *first : per;
}

static int env_int(const char* name, int fallback) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : fallback;
}

static int exchange_id(ncclUniqueId* id, int rank, const char* path) {
    if (rank == 0) {
        char tmp[1024];
        FILE* f;
        if (ncclGetUniqueId(id) != 0) return -1;
        snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        f = fopen(tmp, "wb");
        if (!f || fwrite(id, sizeof(*id), 1, f) != 1) return -1;
        fclose(f);
        return rename(tmp, path);
    }
    for (int tries = 0; tries < 6000; ++tries) { /* up to 60 s */
        FILE* f = fopen(path, "rb");
        if (f) {
            const size_t n = fread(id, sizeof(*id), 1, f);
            fclose(f);
            if (n == 1) return 0;
        }
        {
            struct timespec ts;
            ts.tv_sec = 0;
            ts.tv_nsec = 10 * 1000 * 1000;
            nanosleep(&ts, NULL);
        }
    }
    return -1;
}

int main(int argc, char** argv) {
    enum { BLOCK = 256, BLOCKS_PER_CALL = 8, CH = 2 };
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local_rank = env_int("LOCAL_RANK", rank);
    if (argc == 4 && strcmp(argv[1], "--shards") == 0) {
        const unsigned t = (unsigned)atoi(argv[2]), w = (unsigned)atoi(argv[3]);
        for (unsigned r = 0; w > 0 && r < w; ++r) {
            unsigned f0, c0;
            shard_streams(t, r, w, &f0, &c0);
            printf("%u %u %u\n", r, f0, c0);
        }
        return 0;
    }
    const unsigned total = argc > 1 ? (unsigned)atoi(argv[1]) : 16u;
    const int steps = argc > 2 ? atoi(argv[2]) : 9;
    const char* id_file = getenv("OMX_RCCL_ID_FILE");
    const size_t frames = (size_t)BLOCK * BLOCKS_PER_CALL;
    unsigned first, count, per;
    int n_dev = 0;
    if (world < 1 || rank < 0 || rank >= world || total == 0 || steps < 1) {
        fprintf(stderr, "bad RANK / WORLD_SIZE / arguments\n");
        return 1;
    }
    if (!omx_device_available()) {
        printf("no device\n");
        return 0;
    }
    HIPCHECK(hipGetDeviceCount(&n_dev));
    CHECK(omx_set_device(local_rank % (n_dev > 0 ? n_dev : 1))); /* one process per GPU */
    shard_streams(total, (unsigned)rank, (unsigned)world, &first, &count);
    per = (total + (unsigned)world - 1) / (unsigned)world;

    /* ---- the communicator: this process's, not the library's */
    ncclUniqueId id;
    ncclComm_t comm = NULL;
    memset(&id, 0, sizeof(id));
    if (world > 1 && !id_file) {
        fprintf(stderr, "OMX_RCCL_ID_FILE is needed for WORLD_SIZE > 1\n");
        return 1;
    }
    if (world > 1) {
        if (exchange_id(&id, rank, id_file) != 0) {
            fprintf(stderr, "rank %d: could not exchange the ncclUniqueId through %s\n", rank, id_file);
            return 3;
        }
    } else {
        NCCLCHECK(ncclGetUniqueId(&id));
    }
    NCCLCHECK(ncclCommInitRank(&comm, world, id, rank));

    /* ---- this rank's shard through one capture group */
    omx_capture_group* g = NULL;
    float *pcm = NULL, *table = NULL;
    void *d_pcm = NULL, *d_send = NULL, *d_recv = NULL;
    uint8_t positions[OMX_MAX_CHANNELS];
    const size_t row_bytes = sizeof(float) * OMX_STATS_COLUMNS;
    omx_positions_fallback(CH, positions);
    HIPCHECK(hipMalloc(&d_send, row_bytes * per));
    HIPCHECK(hipMalloc(&d_recv, row_bytes * per * (size_t)world));
    HIPCHECK(hipMemset(d_send, 0, row_bytes * per)); /* padding rows of an uneven shard */
    if (count > 0) {
        omx_capture_group_config cfg;
        omx_capture_group_config_default(&cfg);
        cfg.n_streams = count;
        cfg.visuals = OMX_VISUAL_SPECTROGRAM | OMX_VISUAL_LOUDNESS | OMX_VISUAL_STEREOMETER;
        cfg.block_frames = BLOCK;
        cfg.spectrogram.fft_size = 4096;
        cfg.spectrogram.hop_size = 256;
        cfg.spectrogram.use_reassignment = 1;
        cfg.spectrogram.history_length = 8192;
        cfg.stereometer.analyze_bands = 1;
        CHECK(omx_capture_group_create(&cfg, &g));
        CHECK(omx_capture_group_set_option(g, OMX_OPT_GROUP_STATS, 1));
        pcm = (float*)malloc(sizeof(float) * count * frames * CH);
        HIPCHECK(hipMalloc(&d_pcm, sizeof(float) * count * frames * CH));
    }
    table = (float*)malloc(row_bytes * per * (size_t)world);
    if (!table || (count > 0 && !pcm)) return 4;

    for (int step = 0; step < steps; ++step) {
        if (count > 0) {
            omx_capture_group_update up;
            for (unsigned s = 0; s < count; ++s) { /* a capture's samples depend on its GLOBAL index only */
                const unsigned gs = first + s;
                const double hz = 440.0 * pow(2.0, (double)(gs % 24u) / 12.0), side = 0.2 + 0.05 * (double)(gs % 7u);
                for (size_t f = 0; f < frames; ++f) {
                    const double t = (double)((size_t)step * frames + f) / 48000.0;
                    const float v = (float)(0.5 * sin(2.0 * 3.14159265358979323846 * hz * t));
                    pcm[(s * frames + f) * CH] = v;
                    pcm[(s * frames + f) * CH + 1] = (float)(-side) * v;
                }
            }
            HIPCHECK(hipMemcpy(d_pcm, pcm, sizeof(float) * count * frames * CH, 1));
            CHECK(omx_capture_group_ingest(g, (const float*)d_pcm, frames, CH, 48000.0f, positions, NULL, &up));
            HIPCHECK(hipDeviceSynchronize());
            if (!up.d_stats_rows) {
                fprintf(stderr, "rank %d: no summary rows\n", rank);
                return 5;
            }
            HIPCHECK(hipMemcpy(d_send, up.d_stats_rows, row_bytes * count, 3));
        }
        /* K8: the only exchange of the path — 48 B per capture, once per step */
        NCCLCHECK(ncclAllGather(d_send, d_recv, (size_t)per * OMX_STATS_COLUMNS, NCCL_FLOAT32, comm, NULL));
        HIPCHECK(hipDeviceSynchronize());
    }
    HIPCHECK(hipMemcpy(table, d_recv, row_bytes * per * (size_t)world, 2));

    /* ---- the gathered table, padding rows dropped: rows of capture s for s = 0 ... total - 1 */
    {
        double checksum = 0.0, lufs = 0.0, rho = 0.0;
        unsigned rows_seen = 0, finite = 1;
        for (int r = 0; r < world; ++r) {
            unsigned f0, c0;
            shard_streams(total, (unsigned)r, (unsigned)world, &f0, &c0);
            for (unsigned s = 0; s < c0; ++s) {
                const float* row = table + ((size_t)r * per + s) * OMX_STATS_COLUMNS;
                for (int k = 0; k < OMX_STATS_COLUMNS; ++k) {
                    if (!isfinite(row[k])) finite = 0;
                    checksum += (double)row[k] * (double)(1 + ((f0 + s) * 31u + (unsigned)k * 7u) % 97u);
                }
                lufs += row[0];
                rho += row[3];
                ++rows_seen;
            }
        }
        printf("rank %d world %d total %u shard_first %u shard_count %u rows %u finite %u mean_momentary %.6f mean_rho %.6f checksum %.9e\n", rank, world,
               total, first, count, rows_seen, finite, lufs / rows_seen, rho / rows_seen, checksum);
    }
    if (g) omx_capture_group_destroy(g);
    NCCLCHECK(ncclCommDestroy(comm));
    if (d_pcm) hipFree(d_pcm);
    hipFree(d_send);
    hipFree(d_recv);
    free(pcm);
    free(table);
    return 0;
}
