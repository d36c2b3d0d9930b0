// Single-wavefront VALU issue cadence on gfx950: cycles per instruction for dependent / independent chains of packed-f32,
// scalar-f32 and f64 operations, one wave per SIMD.  hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, UNROLL = 16;

template <int KIND, int CHAINS>
__global__ __launch_bounds__(64) void chain_kernel(float seed, unsigned long long* cycles, float* sink) {
    v2f p[CHAINS];
    double d[CHAINS];
    float f[CHAINS];
    for (int c = 0; c < CHAINS; ++c) {
        p[c] = v2f{seed + c, seed - c};
        d[c] = (double)seed + c;
        f[c] = seed + c;
    }
    const v2f pm{1.0000001f, 0.9999999f};
    const double dm = 1.0000001;
    const long long t0 = clock64();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if constexpr (KIND == 0) p[c] = __builtin_elementwise_fma(p[c], pm, pm);   // v_pk_fma_f32
                if constexpr (KIND == 1) f[c] = __builtin_fmaf(f[c], 1.0000001f, 0.5f);     // v_fma_f32
                if constexpr (KIND == 2) d[c] = __builtin_fma(d[c], dm, 0.5);               // v_fma_f64
                if constexpr (KIND == 3) d[c] = d[c] + dm;                                  // v_add_f64
                if constexpr (KIND == 4) p[c] = p[c] * pm;                                  // v_pk_mul_f32
                if constexpr (KIND == 5) d[c] = (d[c] >= dm) ? d[c] - dm : d[c] + dm;       // cmp + 2 cndmask + add/sub mix
            }
        }
    }
    const long long t1 = clock64();
    float acc = 0.0f;
    for (int c = 0; c < CHAINS; ++c) acc += p[c].x + p[c].y + (float)d[c] + f[c];
    if (threadIdx.x == 0) cycles[blockIdx.x] = (unsigned long long)(t1 - t0);
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int KIND, int CHAINS>
void run(const char* name, int blocks) {
    unsigned long long* dc;
    float* ds;
    hipMalloc(&dc, blocks * sizeof(unsigned long long));
    hipMalloc(&ds, blocks * 64 * sizeof(float));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((chain_kernel<KIND, CHAINS>), dim3(blocks), dim3(64), 0, 0, 1.0f, dc, ds);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), dc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks;
    const double n = (double)ITERS * UNROLL * CHAINS;
    printf("%-28s chains=%d blocks=%4d : %.2f clock64 ticks / instruction\n", name, CHAINS, blocks, mean / n);
    hipFree(dc);
    hipFree(ds);
}

int main() {
    // calibrate clock64 against wall time
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        unsigned long long* dc;
        float* ds;
        hipMalloc(&dc, 8);
        hipMalloc(&ds, 256);
        hipLaunchKernelGGL((chain_kernel<0, 1>), dim3(1), dim3(64), 0, 0, 1.0f, dc, ds);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((chain_kernel<0, 1>), dim3(1), dim3(64), 0, 0, 1.0f, dc, ds);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c;
        hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("clock64: %.1f MHz (ticks %llu in %.3f ms); dependent pk_fma: %.1f ns / instruction\n", c / (ms * 1e3), c, ms,
               ms * 1e6 / ((double)ITERS * UNROLL));
    }
    for (int blocks : {256, 1024, 2048}) {
        run<0, 1>("v_pk_fma_f32 dependent", blocks);
        run<0, 4>("v_pk_fma_f32 4 chains", blocks);
        run<4, 1>("v_pk_mul_f32 dependent", blocks);
        run<4, 4>("v_pk_mul_f32 4 chains", blocks);
        run<1, 1>("v_fma_f32 dependent", blocks);
        run<1, 4>("v_fma_f32 4 chains", blocks);
        run<2, 1>("v_fma_f64 dependent", blocks);
        run<2, 4>("v_fma_f64 4 chains", blocks);
        run<3, 1>("v_add_f64 dependent", blocks);
        run<3, 4>("v_add_f64 4 chains", blocks);
        run<5, 1>("f64 cmp+select+add dependent", blocks);
        run<5, 4>("f64 cmp+select+add 4 chains", blocks);
    }
    return 0;
}
