// One wavefront alone on a CU: cycles per VALU instruction for a dependent chain and for 2 / 4 / 8 independent chains (f32 fma, f64 add,
// v_cndmask).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, int KIND>
__global__ void k(float* out, long long* cycles, float seed, int waves) {
    float x[CHAINS];
    double d[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { x[c] = seed + c; d[c] = seed + c; }
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[c]) : "v"(seed));
                if (KIND == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"((double)seed));
                if (KIND == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(seed));
                if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(seed));
            }
    }
    const long long t1 = clock64();
    float acc = 0;
    for (int c = 0; c < CHAINS; ++c) acc += x[c] + (float)d[c];
    out[threadIdx.x + blockIdx.x * blockDim.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
template <int CHAINS, int KIND>
void run(const char* name, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
    k<CHAINS, KIND><<<1, threads>>>(out, cyc, 1.0f, 1);
    k<CHAINS, KIND><<<1, threads>>>(out, cyc, 1.0f, 1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-10s chains %d threads %4d: %.2f cycles / instruction (per wave)\n", name, CHAINS, threads, (double)h / (1000.0 * 16 * CHAINS));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1, 0>("fma_f32", 64); run<2, 0>("fma_f32", 64); run<4, 0>("fma_f32", 64); run<8, 0>("fma_f32", 64);
    run<1, 3>("add_f32", 64); run<4, 3>("add_f32", 64);
    run<1, 1>("add_f64", 64); run<2, 1>("add_f64", 64); run<4, 1>("add_f64", 64);
    run<1, 2>("cndmask", 64); run<4, 2>("cndmask", 64);
    run<1, 0>("fma_f32", 256); run<4, 0>("fma_f32", 256); run<1, 0>("fma_f32", 512); run<4, 0>("fma_f32", 512);
    run<1, 1>("add_f64", 256); run<4, 1>("add_f64", 256);
    return 0;
}
