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
        // asinh(x) = ln(x + sqrt(x^2 + 1)); x = hz / 20 >= 0 here and the display axis starts at 1 Hz, so the direct form
        // loses nothing a pixel could see (WGSL leaves asinh's precision to the implementation anyway)
        case OMX_FREQ_SCALE_LOGARITHMIC: {
            const float x = hz / 20.0f;
            return x >= 0.0f ? __logf(x + sqrtf(x * x + 1.0f)) : asinhf(x);
        }
        case OMX_FREQ_SCALE_ERB: return 21.4f * __logf(1.0f + hz / 228.8f) * 0.4342944819f;
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

// column c (time order, 0 = oldest) -> storage slot; age = n_columns - 1 - c is the reference's (newest + hl - slot) % hl
// (spectrogram.wgsl:141) for the visible slots of the ring
__device__ __forceinline__ uint32_t splat_slots(const SplatArgs& a) { return a.ring_slots ? a.ring_slots : a.n_columns; }
__device__ __forceinline__ uint32_t splat_slot(const SplatArgs& a, uint32_t col) {
    if (!a.ring_slots) return col;
    const uint32_t v = a.slot0 + col;
    return v >= a.ring_slots ? v - a.ring_slots : v;
}

constexpr uint32_t SPLAT_COLS_PER_WG = 8;  // global-atomic form: a workgroup walks 8 columns x ceil(points / 256) chunks

struct SplatFootprint {
    bool live;
    float power;
    uint32_t i0, i1, j0, j1;  // covered pixels [i0, i1) x [j0, j1)
};
// vs_accum_splat + fs_accum for point `i` of column `col` (spectrogram.wgsl:126-147, 215-226)
__device__ __forceinline__ SplatFootprint splat_footprint(const SplatArgs& a, uint32_t s, uint32_t col, uint32_t n, uint32_t i) {
    SplatFootprint f{};
    bool live = i < n;
    omx_spectrogram_point p{0.0f, 0.0f, 0.0f};
    if (live) p = a.points[((uint64_t)s * splat_slots(a) + splat_slot(a, col)) * a.column_stride + i];
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
    f.live = live;
    f.power = power;
    if (live) {
        f.i0 = (uint32_t)fmaxf(fi0, 0.0f);
        f.i1 = (uint32_t)fminf(fi1, (float)a.width);
        f.j0 = (uint32_t)fmaxf(fj0, 0.0f);
        f.j1 = (uint32_t)fminf(fj1, (float)a.height);
    }
    return f;
}

// ---- form 1: straight to the image with global atomics (any image size) ---------------------------------------------
__global__ __launch_bounds__(256) void splat_accumulate_kernel(SplatArgs a) {
    // grid: x = ceil(n_columns / SPLAT_COLS_PER_WG), y = stream
    const uint32_t s = blockIdx.y;
    const uint32_t col_end = min(a.n_columns, (blockIdx.x + 1u) * SPLAT_COLS_PER_WG);
    float* acc = a.accum + (uint64_t)s * a.width * a.height;
    const uint32_t reach = (uint32_t)ceilf(a.scale_factor) + 1u;  // a footprint is at most ceil(scale_factor) + 1 wide
    for (uint32_t col = blockIdx.x * SPLAT_COLS_PER_WG; col < col_end; ++col) {
        const uint32_t n = min(a.counts[(uint64_t)s * splat_slots(a) + splat_slot(a, col)], a.column_stride);
        for (uint32_t base = 0; base < n; base += 256u) {
            if (base + (threadIdx.x & ~63u) >= n) continue;  // whole wave past the end of the column
            const SplatFootprint f = splat_footprint(a, s, col, n, base + threadIdx.x);
            // footprint offsets are walked in lock-step by the whole wave
            for (uint32_t dx = 0; dx < reach; ++dx) {
                for (uint32_t dy = 0; dy < reach; ++dy) {
                    const bool on = f.live && f.i0 + dx < f.i1 && f.j0 + dy < f.j1;
                    if (__ballot(on) == 0ull) continue;
                    // [width][height]: neighbouring bins of one column are neighbouring addresses
                    wave_merged_add(acc, on, (f.i0 + dx) * a.height + (f.j0 + dy), f.power);
                }
            }
        }
    }
}

// ---- form 2: LDS-tiled.  The global form is bound by the L2's atomic transaction rate (one transaction per scattered
// pixel).  Here a workgroup owns a tile of consecutive columns and a band of rows: it accumulates every footprint pixel that
// falls into its [x window] x [row band] with LDS atomics, then flushes the window to the image with coalesced atomics (the
// windows of neighbouring tiles overlap by the time-reassignment margin).  Pixels are partitioned by ROW BAND, so every
// (point, pixel) pair is added exactly once; pixels of the band outside the x window go straight to the image.
__global__ __launch_bounds__(1024) void splat_tiled_kernel(SplatArgs a, SplatTiling t) {
    extern __shared__ float tile[];  // [window_width][band_rows + 1]: the odd stride spreads one row of many columns over the banks
    const uint32_t s = blockIdx.z, band = blockIdx.y;
    const uint32_t c0 = blockIdx.x * t.tile_cols, c1 = min(a.n_columns, c0 + t.tile_cols);
    const uint32_t j_lo = band * t.band_rows, j_hi = min(a.height, j_lo + t.band_rows);
    // x window: from margin_cols behind the tile's oldest column to 2 columns ahead of its newest one
    const float sf = a.scale_factor;
    const float x_old = a.extent_x - ((float)(a.n_columns - 1u - c0) + (float)t.margin_cols) * sf;
    const float x_new = a.extent_x - ((float)(a.n_columns - c1) - 2.5f) * sf;
    const int xl = max(0, (int)floorf(x_old) - 1);
    const int xh = min((int)a.width, min(xl + (int)t.window_width, (int)ceilf(x_new) + 1));
    const uint32_t x_lo = (uint32_t)xl, x_hi = (uint32_t)max(xh, xl);
    const uint32_t rows = j_hi - j_lo, stride = t.band_rows + 1u, cells = (x_hi - x_lo) * stride;
    for (uint32_t k = threadIdx.x; k < cells; k += 1024u) tile[k] = 0.0f;
    __syncthreads();
    float* acc = a.accum + (uint64_t)s * a.width * a.height;
    const uint32_t reach = (uint32_t)ceilf(sf) + 1u;
    for (uint32_t col = c0; col < c1; ++col) {
        const uint32_t n = min(a.counts[(uint64_t)s * splat_slots(a) + splat_slot(a, col)], a.column_stride);
        for (uint32_t base = 0; base < n; base += 1024u) {
            const SplatFootprint f = splat_footprint(a, s, col, n, base + threadIdx.x);
            if (!f.live || f.j1 <= j_lo || f.j0 >= j_hi) continue;
            for (uint32_t dx = 0; dx < reach && f.i0 + dx < f.i1; ++dx) {
                const uint32_t ix = f.i0 + dx;
                for (uint32_t dy = 0; dy < reach && f.j0 + dy < f.j1; ++dy) {
                    const uint32_t jy = f.j0 + dy;
                    if (jy < j_lo || jy >= j_hi) continue;  // another band's pixel
                    if (ix >= x_lo && ix < x_hi)
                        atomicAdd(&tile[(ix - x_lo) * stride + (jy - j_lo)], f.power);
                    else
                        unsafeAtomicAdd(acc + (uint64_t)ix * a.height + jy, f.power);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < cells; k += 1024u) {
        const uint32_t r = k % stride;
        const float v = tile[k];
        if (r < rows && v != 0.0f) unsafeAtomicAdd(acc + (uint64_t)(x_lo + k / stride) * a.height + j_lo + r, v);
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

void launch_splat(const SplatArgs& a, float* db, float power_scale, hipStream_t stream, int force_form) {
    const uint64_t px = (uint64_t)a.n_streams * a.width * a.height;
    if (px == 0) return;
    OMX_HIP(hipMemsetAsync(a.accum, 0, px * sizeof(float), stream));
    if (a.n_columns && a.column_stride) {
        // tiling: 32 columns per workgroup, window = tile + 17 columns behind (time reassignment reaches back by up to
        // window/hop columns; anything further falls back to global atomics) + 3 ahead, rows banded to fit 128 KiB of LDS
        SplatTiling t{};
        t.tile_cols = 32;
        t.margin_cols = 17;
        t.window_width = (uint32_t)std::ceil((float)(t.tile_cols + t.margin_cols + 3) * a.scale_factor) + 3u;
        constexpr uint32_t lds_floats = 32768;
        t.band_rows = std::min<uint32_t>(lds_floats / t.window_width - 1u, a.height);
        const uint32_t bands = t.band_rows ? (a.height + t.band_rows - 1) / t.band_rows : 0;
        const bool tiled = force_form == 2 || (force_form != 1 && t.band_rows >= 64 && bands <= 8);
        if (tiled && t.band_rows) {
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(splat_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(lds_floats * sizeof(float)));
                attr_set = true;
            }
            hipLaunchKernelGGL(splat_tiled_kernel, dim3((a.n_columns + t.tile_cols - 1) / t.tile_cols, bands, a.n_streams), dim3(1024),
                               (size_t)t.window_width * (t.band_rows + 1u) * sizeof(float), stream, a, t);
        } else {
            hipLaunchKernelGGL(splat_accumulate_kernel, dim3((a.n_columns + SPLAT_COLS_PER_WG - 1) / SPLAT_COLS_PER_WG, a.n_streams),
                               dim3(256), 0, stream, a);
        }
    }
    if (db) hipLaunchKernelGGL(splat_resolve_kernel, dim3((uint32_t)((px + 255) / 256)), dim3(256), 0, stream, a.accum, db, px, power_scale);
}

// ---- column history ring ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void history_scatter_kernel(HistoryScatterArgs a) {
    const uint32_t c = a.first_col + blockIdx.x, s = blockIdx.y;
    uint32_t slot = (a.slot0 + c) % a.ring_slots;
    const uint32_t n = a.src_counts ? min(a.src_counts[(uint64_t)s * a.n_cols + c], a.ring_stride) : min(a.src_stride, a.ring_stride);
    const unsigned char* src_b = a.src + ((uint64_t)s * a.n_cols + c) * a.src_stride * a.elem;
    unsigned char* dst_b = a.ring + ((uint64_t)s * a.ring_slots + slot) * a.ring_stride * a.elem;
    if (a.elem == 2u) {  // classic u16 codes: bank rows of an odd bin count are only 2-byte aligned
        const uint16_t* src = reinterpret_cast<const uint16_t*>(src_b);
        uint16_t* dst = reinterpret_cast<uint16_t*>(dst_b);
        for (uint32_t i = threadIdx.x; i < a.ring_stride; i += 256u) dst[i] = i < n ? src[i] : (uint16_t)0;  // zero-filled to the stride
    } else {             // 12-byte points: whole 4-byte words
        const uint32_t* src = reinterpret_cast<const uint32_t*>(src_b);
        uint32_t* dst = reinterpret_cast<uint32_t*>(dst_b);
        const uint32_t words = n * (a.elem / 4u);
        for (uint32_t i = threadIdx.x; i < words; i += 256u) dst[i] = src[i];
    }
    if (a.slot_counts && threadIdx.x == 0) a.slot_counts[(uint64_t)s * a.ring_slots + slot] = a.src_counts[(uint64_t)s * a.n_cols + c];
}
void launch_history_scatter(const HistoryScatterArgs& a, hipStream_t stream) {
    if (a.n_cols <= a.first_col || a.n_streams == 0) return;
    hipLaunchKernelGGL(history_scatter_kernel, dim3(a.n_cols - a.first_col, a.n_streams), dim3(256), 0, stream, a);
}

__global__ __launch_bounds__(256) void history_remap_kernel(HistoryRemapArgs a) {
    const uint32_t src = blockIdx.x, s = blockIdx.y;
    const uint32_t dst = (src + a.old_slots - a.start) % a.old_slots;
    if (dst >= a.keep || dst >= a.new_slots) return;
    const uint32_t* from = reinterpret_cast<const uint32_t*>(a.old_ring + ((uint64_t)s * a.old_slots + src) * a.stride_bytes);
    uint32_t* to = reinterpret_cast<uint32_t*>(a.new_ring + ((uint64_t)s * a.new_slots + dst) * a.stride_bytes);
    for (uint32_t i = threadIdx.x; i < a.stride_bytes / 4u; i += 256u) to[i] = from[i];
    if (a.old_counts && threadIdx.x == 0) a.new_counts[(uint64_t)s * a.new_slots + dst] = a.old_counts[(uint64_t)s * a.old_slots + src];
}
void launch_history_remap(const HistoryRemapArgs& a, hipStream_t stream) {
    if (a.old_slots == 0 || a.n_streams == 0) return;
    hipLaunchKernelGGL(history_remap_kernel, dim3(a.old_slots, a.n_streams), dim3(256), 0, stream, a);
}

__global__ __launch_bounds__(64) void history_fit_kernel(const uint32_t* slot_counts, uint32_t ring_slots, uint32_t* state) {
    uint32_t needed = 1;
    for (uint32_t i = threadIdx.x; i < ring_slots; i += 64u) needed = max(needed, slot_counts[i]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) needed = max(needed, (uint32_t)__shfl_xor((int)needed, o));
    if (threadIdx.x == 0) {
        const uint32_t current = state[0];
        const unsigned long long quad = max((unsigned long long)needed * 4ull, 1ull);
        if (needed > current || (unsigned long long)current > quad) state[0] = needed;
    }
}
void launch_history_fit(const uint32_t* slot_counts, uint32_t ring_slots, uint32_t* state, hipStream_t stream) {
    hipLaunchKernelGGL(history_fit_kernel, dim3(1), dim3(64), 0, stream, slot_counts, ring_slots, state);
}

}  // namespace omx
