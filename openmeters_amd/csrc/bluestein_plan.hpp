// Kernel-argument view of a Bluestein (chirp-z) plan: an n-point DFT through a cyclic convolution of m = next_pow2(2n - 1) points
// (device code: fft_bluestein in fft_device.hpp; host builder: BluesteinTables in common.hpp).
#pragma once
#include <cstdint>

namespace omx {

typedef float v2f __attribute__((ext_vector_type(2)));  // (re, im)

struct BluesteinPlan {
    uint32_t m;         // 0: the length is a power of two (plain radix-2)
    uint32_t log_m;
    const v2f* chirp;   // [n]  exp(-i pi k^2 / n)
    const v2f* bf;      // [m]  FFT_m of the chirp filter b[k] = b[m - k] = exp(+i pi k^2 / n), k < n
    const v2f* tw_m;    // exp(-2 pi i k / m), k < m/2
};

}  // namespace omx
