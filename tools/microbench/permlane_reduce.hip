// Check of the butterfly reduction used by scope_fast_kernels.hip (copy of reduce_m) against plain sums.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
// (inline asm: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane32_swap into one register when they are added —
// tools/microbench/permlane_reduce.hip; the s_nop covers the VALU-write -> permlane-read wait states the compiler would insert)
__device__ __forceinline__ float swap_add32(float a, float b) {  // lanes 0-31: a[l] + a[l + 32]; lanes 32-63: b[l - 32] + b[l]
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float swap_add16(float a, float b) {  // rows (a0 + a1, b0 + b1, a2 + a3, b2 + b3)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float row_sum(float x) {
    x += dpp_f<0x111, 0xf>(0.0f, x);
    x += dpp_f<0x112, 0xf>(0.0f, x);
    x += dpp_f<0x114, 0xf>(0.0f, x);
    x += dpp_f<0x118, 0xf>(0.0f, x);
    return x;
}
__global__ void k(float* out) {
    const int l = threadIdx.x;
    float v[4] = {1.0f * l, 100.0f + l, 1000.0f + 2 * l, 5.0f};
    const float z1 = swap_add32(v[0], v[1]);
    const float z2 = swap_add32(v[2], v[3]);
    out[l] = row_sum(swap_add16(z1, z2));
    float z = row_sum(swap_add32(v[0], v[1]));
    z += dpp_f<0x142, 0xa>(0.0f, z);
    out[64 + l] = z;
    out[128 + l] = z1;
    out[192 + l] = swap_add16(z1, z2);
}
int main() {
    float* d;
    float h[256];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("expect v0 %g v1 %g v2 %g v3 %g\n", 63.0 * 64 / 2, 6400 + 63.0 * 32, 64000 + 63.0 * 64, 320.0);
    printf("M=4: lane15 %g lane31 %g lane47 %g lane63 %g\n", h[15], h[31], h[47], h[63]);
    printf("M=2: lane31 %g lane63 %g\n", h[64 + 31], h[64 + 63]);
    printf("z1: [0]=%g [31]=%g [32]=%g [63]=%g (expect 32, 94, 232, 294)\n", h[128], h[128 + 31], h[128 + 32], h[128 + 63]);
    printf("after swap16 rows: [0]=%g [16]=%g [32]=%g [48]=%g\n", h[192], h[192 + 16], h[192 + 32], h[192 + 48]);
    return 0;
}
