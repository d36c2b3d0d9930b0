// Which SIMD does wavefront w of a workgroup land on?  (HW_ID.SIMD_ID, gfx9 layout: bits 5:4; CU_ID bits 11:8)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main() {
    for (int threads : {256, 320, 384, 512, 640}) {
        const int blocks = 6, waves = threads / 64;
        unsigned* d;
        (void)hipMalloc(&d, blocks * 16 * sizeof(unsigned));
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(blocks * 16);
        (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
        printf("%d threads:\n", threads);
        for (int b = 0; b < blocks; ++b) {
            printf("  wg %d (cu %2u): simd of wave 0..%d =", b, (h[b * 16] >> 8) & 15u, waves - 1);
            for (int w = 0; w < waves; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3u);
            printf("\n");
        }
        (void)hipFree(d);
    }
    return 0;
}
