// HIP kernels of the spectrum path (gfx950 only).
//   K3a spectrum_power_*   DC-removed windowed real FFT -> |X|^2 * norm per (stream, trace, hop)
//                          (reference src/util/audio/window.rs:66-88, src/visuals/spectrum/processor.rs:215-244)
//   K3b spectrum_levels    per-bin None / Exponential / PeakHold recurrence over hops + raw and
//                          A-weighted dB (reference spectrum/processor.rs:349-402)
#include "fft_device.hpp"
#include "stft_kernels.hpp"

namespace omx {

// ---- K3a fast: N = 4096, one (stream, trace, hop) per 256-thread workgroup, FFT in LDS ------------
__global__ __launch_bounds__(256) void spectrum_power_4096_kernel(SpectrumPowerArgs a) {
    __shared__ v2f A[FFT4096_LDS];
    __shared__ float wave_sum[4];
    const uint32_t item = blockIdx.x;  // ((s * n_traces) + tr) * n_hops + h, hop fastest
    const uint32_t h = item % a.n_hops, st = item / a.n_hops;
    const uint32_t tr = st % a.n_traces, s = st / a.n_traces;
    const int j = threadIdx.x;
    const float* ring = a.ring[tr] + (uint64_t)s * a.cap;
    const uint64_t mask = a.cap - 1;
    const uint64_t p0 = a.tail + (uint64_t)(a.first_hop + h) * a.hop;
    float x[16];
    float partial = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        x[t] = ring[(p0 + (uint32_t)(j + 256 * t)) & mask];
        partial += x[t];
    }
    // window.rs:80-84 mean (tree order here; the generic kernel keeps the sequential order)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) partial += __shfl_xor(partial, off);
    if ((j & 63) == 0) wave_sum[j >> 6] = partial;
    __syncthreads();
    const float mean = (wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3]) / 4096.0f;
    v2f v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = v2f{(x[t] - mean) * a.window[j + 256 * t], 0.0f};
    const Fft4096Tables tb{a.tw256, a.tw4096};
    fft4096<false>(v, A, j, tb);
    float* out = a.power + (uint64_t)item * a.bins;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int k = j + 256 * t;
        out[k] = (v[t].x * v[t].x + v[t].y * v[t].y) * a.bin_norm[k];
    }
    if (j == 0) out[2048] = (v[8].x * v[8].x + v[8].y * v[8].y) * a.bin_norm[2048];
}

// ---- K3a generic: any power-of-two N, radix-2 in a global workspace, reference operation order ------
__global__ __launch_bounds__(256) void spectrum_power_generic_kernel(SpectrumPowerArgs a) {
    __shared__ float mean_sh;
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    const uint64_t total = (uint64_t)a.n_streams * a.n_traces * a.n_hops;
    v2f* ws = a.workspace + (uint64_t)blockIdx.x * a.fft_size;
    for (uint64_t item = blockIdx.x; item < total; item += gridDim.x) {
        const uint32_t h = (uint32_t)(item % a.n_hops), st = (uint32_t)(item / a.n_hops);
        const uint32_t tr = st % a.n_traces, s = st / a.n_traces;
        const float* ring = a.ring[tr] + (uint64_t)s * a.cap;
        const uint64_t mask = a.cap - 1;
        const uint64_t p0 = a.tail + (uint64_t)(a.first_hop + h) * a.hop;
        __syncthreads();
        if (tid == 0) {
            float sum = -0.0f;
            for (uint32_t i = 0; i < a.fft_size; ++i) sum = sum + ring[(p0 + i) & mask];
            mean_sh = sum / (float)a.fft_size;
        }
        __syncthreads();
        const float mean = mean_sh;
        for (uint32_t i = tid; i < a.fft_size; i += nt) ws[i] = v2f{(ring[(p0 + i) & mask] - mean) * a.window[i], 0.0f};
        fft_radix2(ws, a.fft_size, a.log_fft, a.tw_fft, false, tid, nt);
        float* out = a.power + item * a.bins;
        for (uint32_t i = tid; i < a.bins; i += nt) {
            const v2f c = ws[i];
            out[i] = (c.x * c.x + c.y * c.y) * a.bin_norm[i];
        }
    }
}

void launch_spectrum_power(const SpectrumPowerArgs& a, bool fast4096, uint32_t generic_wgs, hipStream_t stream) {
    const uint64_t total = (uint64_t)a.n_streams * a.n_traces * a.n_hops;
    if (total == 0) return;
    if (fast4096) hipLaunchKernelGGL(spectrum_power_4096_kernel, dim3((uint32_t)total), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(spectrum_power_generic_kernel, dim3(generic_wgs), dim3(256), 0, stream, a);
}

// ---- K3b: per-bin recurrences + dB ------------------------------------------------------------------
__global__ __launch_bounds__(256) void spectrum_levels_kernel(SpectrumLevelsArgs a) {
    const uint32_t bin = blockIdx.x * 256 + threadIdx.x;
    const uint32_t tr = blockIdx.y, s = blockIdx.z;
    if (bin >= a.bins) return;
    const uint32_t slot = a.trace_slot[tr];  // 0/1: which output trace this active trace fills
    float* state = a.smoothed ? a.smoothed + ((uint64_t)s * 2 + slot) * a.bins + bin : nullptr;
    float st = state ? *state : 0.0f;
    const float aw = a.a_weighting_db[bin];
    const float* pw = a.power + ((uint64_t)s * a.n_traces + tr) * a.n_hops * a.bins + bin;
    for (uint32_t h = 0; h < a.n_hops; ++h) {
        const float power = pw[(uint64_t)h * a.bins];
        float p = power;
        if (a.mode == OMX_AVERAGING_EXPONENTIAL) {  // :366-379
            st = (st <= 0.0f) ? power : st * a.alpha + power * (1.0f - a.alpha);
            if (st < a.state_floor) st = 0.0f;
            p = st;
        } else if (a.mode == OMX_AVERAGING_PEAK_HOLD) {  // :380-389
            st = fmaxf(st * a.decay, power);
            if (st < a.state_floor) st = 0.0f;
            p = st;
        }
        const bool last = h + 1 == a.n_hops;
        if (a.emit_all || last) {
            const uint32_t ho = a.emit_all ? h : 0;
            float* out = a.traces + (((uint64_t)s * a.n_hops_out + ho) * 2 + slot) * 2 * a.bins + bin;
            float raw = a.floor_db, weighted = a.floor_db;  // :392-401
            if (!(p < a.state_floor)) {
                const float db = logf(p) * 4.3429448f;
                raw = fmaxf(db, a.floor_db);
                weighted = fmaxf(db + aw, a.floor_db);
            }
            out[0] = weighted;
            out[a.bins] = raw;
        }
    }
    if (state) *state = st;
}

void launch_spectrum_levels(const SpectrumLevelsArgs& a, hipStream_t stream) {
    if (a.n_hops == 0 || a.n_traces == 0) return;
    dim3 grid((a.bins + 255) / 256, a.n_traces, a.n_streams);
    hipLaunchKernelGGL(spectrum_levels_kernel, grid, dim3(256), 0, stream, a);
}

__global__ void fill_kernel(float* p, uint64_t n, float v) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill(float* p, uint64_t n, float v, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, n, v);
}

}  // namespace omx
