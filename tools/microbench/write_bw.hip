// HBM write bandwidth of the row-store patterns the spectrum / spectrogram kernels use (gfx950).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/write_bw tools/microbench/write_bw.hip
//   run  : tools/microbench/write_bw          (on the GPU box)
// Patterns, 1 GiB per launch, 256 threads per workgroup, every lane one 16-byte store per row piece:
//   aligned    rows of 2048 floats (16-byte aligned starts)
//   odd        rows of 2049 floats (4-byte aligned starts: every 16-byte store straddles — the spectrum rows, bins = 2049)
//   odd nt     the same with non-temporal stores
//   read+write one 16-byte load per store from a second buffer (copy): the read side's share of a mixed stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                \
            return 1;                                                             \
        }                                                                         \
    } while (0)

template <bool NT, bool COPY>
__global__ __launch_bounds__(256) void rows_kernel(float* out, const float* in, uint32_t row_floats, uint32_t rows_per_wg) {
    const uint32_t j = threadIdx.x;
    for (uint32_t r = 0; r < rows_per_wg; ++r) {
        const uint64_t row = (uint64_t)blockIdx.x * rows_per_wg + r;
        float* p = out + row * row_floats;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t c = (uint32_t)k * 1024u + 4u * j;
            f4u v{(float)j, (float)r, (float)k, 1.0f};
            if (COPY) v = *reinterpret_cast<const f4u*>(in + row * row_floats + c);
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4u*>(p + c));
            else *reinterpret_cast<f4u*>(p + c) = v;
        }
    }
}

int main() {
    const uint64_t bytes = 1ull << 30;
    float *out = nullptr, *in = nullptr;
    CHECK(hipMalloc(&out, bytes + (1 << 20)));
    CHECK(hipMalloc(&in, bytes + (1 << 20)));
    CHECK(hipMemset(out, 0, bytes + (1 << 20)));
    CHECK(hipMemset(in, 0, bytes + (1 << 20)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Case { const char* name; uint32_t row_floats; int nt, copy; uint32_t rows_per_wg; };
    const Case cases[] = {{"aligned rows, 4 rows per workgroup", 2048, 0, 0, 4},   {"odd rows (2049), 4 rows per workgroup", 2049, 0, 0, 4},
                          {"odd rows, non-temporal", 2049, 1, 0, 4},              {"aligned rows, non-temporal", 2048, 1, 0, 4},
                          {"odd rows, 32 rows per workgroup", 2049, 0, 0, 32},    {"copy, aligned rows", 2048, 0, 1, 4},
                          {"copy, odd rows, non-temporal", 2049, 1, 1, 4}};
    for (const Case& c : cases) {
        const uint64_t rows = bytes / (2048ull * 4ull);
        const uint32_t wgs = (uint32_t)(rows / c.rows_per_wg);
        auto launch = [&] {
            if (c.nt && c.copy) hipLaunchKernelGGL((rows_kernel<true, true>), dim3(wgs), dim3(256), 0, 0, out, in, c.row_floats, c.rows_per_wg);
            else if (c.nt) hipLaunchKernelGGL((rows_kernel<true, false>), dim3(wgs), dim3(256), 0, 0, out, in, c.row_floats, c.rows_per_wg);
            else if (c.copy) hipLaunchKernelGGL((rows_kernel<false, true>), dim3(wgs), dim3(256), 0, 0, out, in, c.row_floats, c.rows_per_wg);
            else hipLaunchKernelGGL((rows_kernel<false, false>), dim3(wgs), dim3(256), 0, 0, out, in, c.row_floats, c.rows_per_wg);
        };
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipDeviceSynchronize());
        const int reps = 20;
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double written = (double)rows * 2048.0 * 4.0;
        printf("%-44s %.3f ms per GiB  -> %.2f TB/s written%s\n", c.name, ms, written / ms / 1e9, c.copy ? " (+ the same read)" : "");
    }
    // hipMemsetAsync for reference
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) CHECK(hipMemsetAsync(out, 0, bytes, 0));
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %.3f ms per GiB  -> %.2f TB/s\n", "hipMemsetAsync", ms / 10, (double)bytes / (ms / 10) / 1e9);
    return 0;
}
