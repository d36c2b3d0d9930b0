// Latency of one step of the sequential window fold (window_sum_kernels.hip) on gfx950: a dependent v_add_f32 whose OTHER operand comes
// through a quad_perm DPP read.  hipcc inserts `s_nop 1` between consecutive steps (its DPP hazard check covers every VGPR the
// instruction reads, the accumulator included); the hardware hazard concerns the DPP-read operand.  Wall time per step, one wavefront
// per SIMD (1024 blocks) and two.   hipcc --offload-arch=gfx950 -O3 dpp_chain.hip -o dpp_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITERS = 2048, UNROLL = 16;

// KIND 0: plain dependent v_add_f32; 1: builtin DPP (hipcc's nops); 2: asm DPP, no nops; 3: asm DPP, two chains interleaved
template <int KIND>
__global__ __launch_bounds__(64) void chain_kernel(const float* in, float* out) {
    float x[UNROLL];
    for (int u = 0; u < UNROLL; ++u) x[u] = in[threadIdx.x * UNROLL + u];
    float sum = -0.0f, sum2 = -0.0f;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if constexpr (KIND == 0) {
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(sum) : "v"(x[u]));
            } else if constexpr (KIND == 1) {
                const float b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[u]), 0x55, 0xf, 0xf, true));
                sum = b + sum;
            } else if constexpr (KIND == 2) {
                asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(sum) : "v"(x[u]));
            } else {
                asm volatile("v_add_f32_dpp %0, %2, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                             "v_add_f32_dpp %1, %2, %1 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:0"
                             : "+v"(sum), "+v"(sum2)
                             : "v"(x[u]));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = sum + sum2;
}

template <int KIND>
void run(const char* name, int blocks) {
    float *din, *dout;
    hipMalloc(&din, 64 * UNROLL * sizeof(float));
    hipMalloc(&dout, blocks * 64 * sizeof(float));
    std::vector<float> h(64 * UNROLL);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f + 0.001f * (float)(i % 37);
    hipMemcpy(din, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, din, dout);
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(chain_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, din, dout);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> o(blocks * 64);
    hipMemcpy(o.data(), dout, o.size() * sizeof(float), hipMemcpyDeviceToHost);
    const double steps = (double)ITERS * UNROLL;
    printf("%-44s blocks=%5d : %.2f ns / step (%.1f cycles at 2.4 GHz), %.1f us per launch; lane 5 -> %.6g\n", name, blocks, ms / 5 * 1e6 / steps,
           ms / 5 * 1e6 / steps * 2.4, ms / 5 * 1e3, (double)o[5]);
    hipFree(din);
    hipFree(dout);
}

int main() {
    for (int blocks : {1024, 2048, 4096}) {
        run<0>("plain dependent v_add_f32", blocks);
        run<1>("DPP operand, hipcc's hazard nops", blocks);
        run<2>("DPP operand, no nops (asm)", blocks);
        run<3>("DPP operand, two chains per wave (asm)", blocks);
    }
    return 0;
}
