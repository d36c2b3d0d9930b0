// Do kernels launched with hipExtAnyOrderLaunch into ONE stream overlap on gfx950?  (hip_ext.h says the flag is not supported on gfx9
// for the module-launch entry point.)  Four one-workgroup kernels that spin ~50 us each: in order = ~200 us, overlapped = ~50 us.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/anyorder tools/microbench/anyorder.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void spin_kernel(unsigned long long cycles, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < cycles) x = x * 1664525u + 1013904223u;
    if (x == 0x12345u) *sink = x;
}

int main() {
    unsigned* sink;
    hipMalloc(&sink, 4);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned long long cycles = 5000;  // wall_clock64 ticks at 100 MHz: 50 us
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            for (int k = 0; k < 4; ++k) {
                if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cycles, sink);
                else hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cycles, sink);
            }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s: four 50-us kernels in one stream: %.1f us\n", mode ? "hipExtAnyOrderLaunch" : "in order", ms * 1000.0f);
        }
    }
    return 0;
}
