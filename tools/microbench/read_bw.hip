// What a read-only pass reaches on this part: every lane sums 16-byte loads, nothing is written but one float per wavefront.
// Variants: bytes per workgroup-contiguous piece (the loudness / waveform pass-A shape: 8 KiB pieces 512 KiB apart, against one
// dense stream), loads in flight per lane, buffer size.   hipcc --offload-arch=gfx950 -O3 read_bw.hip -o read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// grid-stride dense read: workgroup w reads float4s w*256+tid, then + grid*256 ...
template <int INFLIGHT>
__global__ __launch_bounds__(256) void dense_kernel(const float4* __restrict__ in, size_t n4, float* out) {
    float acc = 0.0f;
    const size_t stride = (size_t)gridDim.x * 256u;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += stride * INFLIGHT) {
        float4 v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) v[k] = i + k * stride < n4 ? in[i + k * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
// the meter banks' shape: wavefront (stream s, block c) reads the 8 KiB piece at s * stream_bytes + c * 8 KiB, 8 loads in flight
__global__ __launch_bounds__(64) void pieces_kernel(const float4* __restrict__ in, size_t stream_f4, uint32_t n_streams, float* out, int block_major) {
    const uint32_t s = block_major ? blockIdx.y : blockIdx.x, c = block_major ? blockIdx.x : blockIdx.y;
    const float4* p = in + (size_t)s * stream_f4 + (size_t)c * 512u + threadIdx.x;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[k * 64];
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

// the loudness chunk kernels' tile shape: wavefront = (group of 8 streams, block of 256 frames x 8 channels); a tile = 16 frames of the
// eight streams = eight 512-byte pieces 512 KiB apart, four 16-byte loads per lane; INFLIGHT tiles requested ahead
template <int INFLIGHT>
__global__ __launch_bounds__(64) void tiles_kernel(const float4* __restrict__ in, size_t stream_f4, float* out) {
    const uint32_t group = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
    // lane + 64 n -> byte (lane + 64 n) * 16 of the tile: row = byte / 512 (stream of the group), inrow = byte % 512
    const float4* src[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const uint32_t byte = (lane + 64u * n) * 16u, row = byte / 512u, inrow = byte % 512u;
        src[n] = in + ((size_t)(group * 8u + row) * stream_f4 + (size_t)c * 512u) + inrow / 16u;
    }
    float acc = 0.0f;
    float4 v[INFLIGHT][4];
#pragma unroll
    for (int t = 0; t < INFLIGHT; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) v[t][n] = src[n][(size_t)t * 32u];
    for (uint32_t t0 = 0; t0 < 16u; t0 += INFLIGHT) {
#pragma unroll
        for (int t = 0; t < INFLIGHT; ++t) {
#pragma unroll
            for (int n = 0; n < 4; ++n) acc += v[t][n].x + v[t][n].y + v[t][n].z + v[t][n].w;
            const uint32_t nxt = t0 + (uint32_t)t + INFLIGHT;
            if (nxt < 16u)
#pragma unroll
                for (int n = 0; n < 4; ++n) v[t][n] = src[n][(size_t)nxt * 32u];
        }
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

template <class F>
static double time_ms(F&& launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    for (size_t mb : {512ul, 2048ul, 8192ul}) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        float4* in;
        float* out;
        hipMalloc(&in, bytes);
        hipMalloc(&out, 1 << 20);
        hipMemset(in, 0, bytes);
        for (int grid : {2048, 8192, 32768}) {
            double t1 = time_ms([&] { hipLaunchKernelGGL(dense_kernel<1>, dim3(grid), dim3(256), 0, 0, in, n4, out); }, 20);
            double t4 = time_ms([&] { hipLaunchKernelGGL(dense_kernel<4>, dim3(grid), dim3(256), 0, 0, in, n4, out); }, 20);
            double t8 = time_ms([&] { hipLaunchKernelGGL(dense_kernel<8>, dim3(grid), dim3(256), 0, 0, in, n4, out); }, 20);
            printf("dense  %5zu MiB grid %6d: 1 in flight %.2f TB/s, 4 in flight %.2f TB/s, 8 in flight %.2f TB/s\n", mb, grid, bytes / t1 * 1e-9, bytes / t4 * 1e-9,
                   bytes / t8 * 1e-9);
        }
        const uint32_t n_streams = 1024;
        const size_t stream_f4 = n4 / n_streams;
        const uint32_t blocks = (uint32_t)(stream_f4 / 512u);
        for (int bm : {0, 1}) {
            double t = time_ms([&] { hipLaunchKernelGGL(pieces_kernel, bm ? dim3(blocks, n_streams) : dim3(n_streams, blocks), dim3(64), 0, 0, in, stream_f4, n_streams, out, bm); }, 20);
            printf("pieces %5zu MiB (%u streams x %u blocks of 8 KiB, %s): %.2f TB/s\n", mb, n_streams, blocks, bm ? "block-major" : "stream-major", bytes / t * 1e-9);
        }
        if (blocks >= 1) {
            double t2 = time_ms([&] { hipLaunchKernelGGL(tiles_kernel<2>, dim3(n_streams / 8, blocks), dim3(64), 0, 0, in, stream_f4, out); }, 20);
            double t4 = time_ms([&] { hipLaunchKernelGGL(tiles_kernel<4>, dim3(n_streams / 8, blocks), dim3(64), 0, 0, in, stream_f4, out); }, 20);
            double t8 = time_ms([&] { hipLaunchKernelGGL(tiles_kernel<8>, dim3(n_streams / 8, blocks), dim3(64), 0, 0, in, stream_f4, out); }, 20);
            printf("tiles  %5zu MiB (wavefront = 8 streams x 8 KiB as 16 tiles of eight 512-byte pieces): 2 tiles in flight %.2f TB/s, 4: %.2f, 8: %.2f\n", mb,
                   bytes / t2 * 1e-9, bytes / t4 * 1e-9, bytes / t8 * 1e-9);
        }
        hipFree(in);
        hipFree(out);
    }
    return 0;
}
