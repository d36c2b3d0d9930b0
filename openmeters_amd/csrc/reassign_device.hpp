// Per-bin reassignment and the ordered compaction's offsets, shared by the fused reassigned-STFT kernels
// (reference src/visuals/spectrogram/processor.rs:459-485).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/omx.h"
#include "fft_device.hpp"

namespace omx {

struct ReassignConsts {
    float bin_hz, max_hz, inv_2pi, inv_hop, latency_hops;
};

// processor.rs:459-485 for one bin, the same statements without the early returns: every value is computed and what the returns
// decided goes into the result (NaN / inf from a zero power fail the comparisons).  Written with the early returns, a column's
// bins become a serial chain of `LDS read - wait - arithmetic - branch`, every wait exposed.
// 1 / pow by v_rcp_f32 (1 ulp) and one Newton step: ten dependent VALU instructions fewer per bin than the IEEE division
// sequence, <= 1 ulp from it; a kept bin has pow >= 1e-14 / norm, far from the denormal range the long sequence exists for.
__device__ __forceinline__ bool reassign_flat(uint32_t i, v2f b, v2f d, v2f t, float norm, const ReassignConsts& c, omx_spectrogram_point& p) {
    const float pow = b.x * b.x + b.y * b.y;
    const float scaled_power = pow * norm;
    const float r0 = __builtin_amdgcn_rcpf(pow);
    const float inv_pow = __builtin_fmaf(__builtin_fmaf(-pow, r0, 1.0f), r0, r0);
    const float d_omega = -(d.y * b.x - d.x * b.y) * inv_pow;
    const float freq_hz = (float)i * c.bin_hz + d_omega * c.inv_2pi;
    p.time_offset = (t.x * b.x + t.y * b.y) * inv_pow * c.inv_hop - c.latency_hops;
    p.freq_hz = freq_hz;
    p.power = scaled_power;
    return !(scaled_power < 1e-14f) && freq_hz > 0.0f && c.max_hz - freq_hz > 0.0f;  // ANALYSIS_FLOOR_POWER (:69), the band (:472-475)
}

// inclusive prefix sum over the 64 lanes of a wavefront (row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, then row_bcast 15 / 31)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t x) {
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    x = dpp_add<0x111, 0xf>(x);
    x = dpp_add<0x112, 0xf>(x);
    x = dpp_add<0x114, 0xf>(x);
    x = dpp_add<0x118, 0xf>(x);
    x = dpp_add<0x142, 0xa>(x);  // lane 15 of rows 0 / 2 -> rows 1 / 3
    x = dpp_add<0x143, 0xc>(x);  // lane 31 -> rows 2 and 3
    return x;
}
// number of set bits of a ballot mask below this lane
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}  // namespace omx
