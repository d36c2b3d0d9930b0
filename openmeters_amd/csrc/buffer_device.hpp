// Buffer-addressed global loads for gfx950: `buffer_load_* vdst, voffset, s[rsrc:4], soffset offen`.  A run of loads that walk a
// table or a ring window in fixed steps shares ONE per-lane byte offset (a VGPR) and takes its step from a scalar register — no
// 64-bit vector address per load, which is what hipcc builds for `table[lane + 256 t]` (two VGPRs and a 64-bit add each, and with
// 30-50 such loads in a kernel the addresses alone spill).  Reads past `bytes` return 0 (raw-buffer range check).
#pragma once
#include <hip/hip_runtime.h>

#include "fft_device.hpp"

namespace omx {

struct GlobalBuffer {
    __amdgpu_buffer_rsrc_t rsrc;
};
// `base` and `bytes` must be wave-uniform.  They are pinned to SGPRs here: a base that is uniform in fact but derived from loaded data
// (a ring position read from the stream's state) counts as divergent for the compiler, which then wraps EVERY load of the buffer
// in a waterfall loop — four v_readfirstlane, two 64-bit compares, an exec save / restore and a branch per load (the spectrum
// kernels' 32 ring loads carried ~220 VALU and ~200 SALU instructions of that per wavefront, round 5).
__device__ __forceinline__ GlobalBuffer global_buffer(const void* base, uint32_t bytes) {
#ifdef OMX_BUFFER_NO_PIN  // A/B builds only (tools/build_ab.sh): the descriptor as the compiler sees it
    return GlobalBuffer{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000)};
#endif
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p >> 32));
    const void* ub = reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
    // word 3: DATA_FORMAT = 32 (bits 15-18 unused for raw untyped loads on gfx9), the value every gfx90a / gfx94x / gfx950 raw-buffer user sets
    return GlobalBuffer{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ub), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000)};
}
__device__ __forceinline__ float load_f32(const GlobalBuffer& b, uint32_t lane_bytes, uint32_t uniform_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.rsrc, (int)lane_bytes, (int)uniform_bytes, 0));
}
__device__ __forceinline__ v2f load_v2f(const GlobalBuffer& b, uint32_t lane_bytes, uint32_t uniform_bytes) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 u = __builtin_amdgcn_raw_buffer_load_b64(b.rsrc, (int)lane_bytes, (int)uniform_bytes, 0);
    return __builtin_bit_cast(v2f, u);
}

}  // namespace omx
