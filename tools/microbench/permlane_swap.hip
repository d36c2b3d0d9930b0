// Lane mapping of v_permlane32_swap / v_permlane16_swap on gfx950 (the butterfly reduction of scope_fast_kernels.hip relies on it).
// Build: hipcc --offload-arch=gfx950 -O3 -o permlane_swap permlane_swap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    v2u r = __builtin_amdgcn_permlane32_swap(1000 + l, 2000 + l, false, false);
    out[l] = r.x;
    out[64 + l] = r.y;
    v2u q = __builtin_amdgcn_permlane16_swap(1000 + l, 2000 + l, false, false);
    out[128 + l] = q.x;
    out[192 + l] = q.y;
}
int main() {
    unsigned* d;
    unsigned h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane32_swap .x", "permlane32_swap .y", "permlane16_swap .x", "permlane16_swap .y"};
    for (int r = 0; r < 4; ++r) {
        printf("%s:", names[r]);
        for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * r + l]);
        printf("\n");
    }
    return 0;
}
