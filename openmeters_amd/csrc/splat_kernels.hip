// K11: reassigned-splat accumulation + dB resolve, the consumer right after the spectrogram path
// (reference src/visuals/render/shaders/spectrogram.wgsl:66-76, 126-147, 215-237).
//   splat_accumulate_kernel  one lane per point slot of a column: per-point map to the pixel grid, f32 atomic adds
//                            (the raster pipeline's additive blend).  HBM/L2-atomic bound: 12 B read + >= 1 atomic per point.
//   splat_resolve_kernel     elementwise: x reassigned_power_scale, ln -> dB, -140 dB floor, -inf where nothing landed.
#include "splat.hpp"

namespace omx {

namespace {
__device__ __forceinline__ float freq_scale_value(uint32_t scale, float hz) {
    switch (scale) {
        case OMX_FREQ_SCALE_LOGARITHMIC: return asinhf(hz / 20.0f);
        case OMX_FREQ_SCALE_ERB: return 21.4f * logf(1.0f + hz / 228.8f) * 0.4342944819f;
        default: return hz;
    }
}
}  // namespace

// One atomic per run of neighbouring lanes that target the same pixel: neighbouring lanes hold neighbouring bins of one
// column, and in the upper half of a log / ERB axis dozens of bins share a pixel — un-merged, those atomics serialise on
// one address.  Segmented inclusive scan over the wave (runs = maximal sequences of equal keys), the tail lane of each
// run adds the run's sum.  Non-adjacent duplicates simply issue separate atomics.
__device__ __forceinline__ void wave_merged_add(float* acc, bool active, uint32_t pixel, float power) {
    const int lane = threadIdx.x & 63;
    const uint32_t key = active ? pixel : 0xffffffffu;
    float v = active ? power : 0.0f;
    const uint32_t prev = __shfl_up(key, 1), next = __shfl_down(key, 1);
    int flag = (lane == 0 || prev != key) ? 1 : 0;  // head of a run
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float ov = __shfl_up(v, d);
        const int of = __shfl_up(flag, d);
        if (lane >= d) {
            if (!flag) v += ov;
            flag |= of;
        }
    }
    if (active && (lane == 63 || next != key)) unsafeAtomicAdd(acc + key, v);
}

constexpr uint32_t SPLAT_COLS_PER_WG = 8;  // a workgroup walks 8 columns x ceil(points / 256) chunks: few, fat workgroups

__device__ __forceinline__ void splat_chunk(const SplatArgs& a, uint32_t s, uint32_t col, uint32_t n, uint32_t i) {
    bool live = i < n;
    omx_spectrogram_point p{0.0f, 0.0f, 0.0f};
    if (live) p = a.points[((uint64_t)s * a.n_columns + col) * a.column_stride + i];
    const float zoomed = ((freq_scale_value(a.freq_scale, p.freq_hz) - a.axis_lo) * a.axis_inv - a.uv_lo) * a.inv_uv;
    live = live && p.power > 0.0f && !(zoomed < -0.01f) && !(zoomed > 1.01f);
    float power = p.power;
    if (a.tilt_db != 0.0f && !(power > 1.0023052e-14f)) live = false;
    if (a.tilt_db != 0.0f && p.freq_hz > 0.0f) power *= exp2f(a.tilt_db * log2f(p.freq_hz / 1000.0f) * 0.3321928095f);
    const float sf = a.scale_factor;
    const float age = (float)(a.n_columns - 1u - col);
    const float x = a.extent_x - (age - p.time_offset) * sf, y = (1.0f - zoomed) * a.extent_y;
    const float x0 = x - 0.5f * sf, x1 = x + 0.5f * sf, y0 = y - 0.5f * sf, y1 = y + 0.5f * sf;
    // pixel (i, j) is covered when its centre lies in [x0, x1) x [y0, y1)
    const float fi0 = ceilf(x0 - 0.5f), fi1 = ceilf(x1 - 0.5f), fj0 = ceilf(y0 - 0.5f), fj1 = ceilf(y1 - 0.5f);
    live = live && fi1 > 0.0f && fj1 > 0.0f && fi0 < (float)a.width && fj0 < (float)a.height;
    uint32_t i0 = 0, i1 = 0, j0 = 0, j1 = 0;
    if (live) {
        i0 = (uint32_t)fmaxf(fi0, 0.0f);
        i1 = (uint32_t)fminf(fi1, (float)a.width);
        j0 = (uint32_t)fmaxf(fj0, 0.0f);
        j1 = (uint32_t)fminf(fj1, (float)a.height);
    }
    float* acc = a.accum + (uint64_t)s * a.width * a.height;
    // footprint offsets are walked in lock-step by the whole wave (a footprint is at most ceil(scale_factor) + 1 wide)
    const uint32_t reach = (uint32_t)ceilf(sf) + 1u;
    for (uint32_t dx = 0; dx < reach; ++dx) {
        for (uint32_t dy = 0; dy < reach; ++dy) {
            const bool on = live && i0 + dx < i1 && j0 + dy < j1;
            if (__ballot(on) == 0ull) continue;
            // [width][height]: neighbouring bins of one column are neighbouring addresses
            wave_merged_add(acc, on, (i0 + dx) * a.height + (j0 + dy), power);
        }
    }
}

__global__ __launch_bounds__(256) void splat_accumulate_kernel(SplatArgs a) {
    // grid: x = ceil(n_columns / SPLAT_COLS_PER_WG), y = stream
    const uint32_t s = blockIdx.y;
    const uint32_t col_end = min(a.n_columns, (blockIdx.x + 1u) * SPLAT_COLS_PER_WG);
    for (uint32_t col = blockIdx.x * SPLAT_COLS_PER_WG; col < col_end; ++col) {
        const uint32_t n = min(a.counts[(uint64_t)s * a.n_columns + col], a.column_stride);
        for (uint32_t base = 0; base < n; base += 256u) {
            if (base + (threadIdx.x & ~63u) >= n) continue;  // whole wave past the end of the column
            splat_chunk(a, s, col, n, base + threadIdx.x);
        }
    }
}

__global__ __launch_bounds__(256) void splat_resolve_kernel(const float* __restrict__ accum, float* __restrict__ db, uint64_t n,
                                                           float power_scale) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float power = accum[i] * power_scale;
    float out = -__builtin_inff();
    if (power > 0.0f) out = fmaxf(logf(fmaxf(power, 1e-20f)) * 4.342944819f, -140.0f);
    db[i] = out;
}

void launch_splat(const SplatArgs& a, float* db, float power_scale, hipStream_t stream) {
    const uint64_t px = (uint64_t)a.n_streams * a.width * a.height;
    if (px == 0) return;
    OMX_HIP(hipMemsetAsync(a.accum, 0, px * sizeof(float), stream));
    if (a.n_columns && a.column_stride)
        hipLaunchKernelGGL(splat_accumulate_kernel, dim3((a.n_columns + SPLAT_COLS_PER_WG - 1) / SPLAT_COLS_PER_WG, a.n_streams), dim3(256),
                           0, stream, a);
    if (db) hipLaunchKernelGGL(splat_resolve_kernel, dim3((uint32_t)((px + 255) / 256)), dim3(256), 0, stream, a.accum, db, px, power_scale);
}

}  // namespace omx
