// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is shipped or measured as the
// product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// Plain unnormalised DFT/IDFT, the published contract of the reference's un-vendored FFT
// dependencies (rustfft 6.4.1 `Fft::process_with_scratch`, realfft 3.5.0 `RealToComplex` /
// `ComplexToReal`; reference Cargo.lock:2448-2449, :2344-2345).  rustfft's own f32 rounding
// pattern (AVX/SSE mixed-radix butterflies) cannot be reproduced here: PARITY UNPINNED at the
// 1e-5 level through the FFT boundary — the f64 path below is the definitional check.
//
// Call sites restated: reference src/visuals/spectrogram/processor.rs:235-243, :342, :553-556,
// :574-595; src/visuals/spectrum/processor.rs:128, :221-233; src/visuals/oscilloscope/
// processor.rs:60-62, :154-164.
#pragma once
#include <cmath>
#include <complex>
#include <cstddef>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace omxo {

template <class T>
struct FftPlan {
    size_t n = 0;
    bool pow2 = false;
    std::vector<std::complex<T>> tw;   // tw[k] = exp(-2*pi*i*k/n), k < n/2 (pow2) or k < n (naive)
    std::vector<uint32_t> rev;         // bit reversal (pow2)
};

template <class T>
inline std::shared_ptr<const FftPlan<T>> fft_plan(size_t n) {
    // per-thread front cache so the timed multi-thread baseline does not serialise on the mutex
    static thread_local std::map<size_t, std::shared_ptr<const FftPlan<T>>> local;
    auto lit = local.find(n);
    if (lit != local.end()) return lit->second;
    static std::mutex mu;
    static std::map<size_t, std::shared_ptr<const FftPlan<T>>> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(n);
    if (it != cache.end()) {
        local[n] = it->second;
        return it->second;
    }
    auto p = std::make_shared<FftPlan<T>>();
    p->n = n;
    p->pow2 = n >= 1 && (n & (n - 1)) == 0;
    const double step = -2.0 * M_PI / (double)(n ? n : 1);
    if (p->pow2) {
        p->tw.resize(n / 2 ? n / 2 : 1);
        for (size_t k = 0; k < n / 2; ++k)
            p->tw[k] = std::complex<T>((T)std::cos(step * (double)k), (T)std::sin(step * (double)k));
        p->rev.resize(n);
        unsigned bits = 0;
        while ((size_t(1) << bits) < n) ++bits;
        for (size_t i = 0; i < n; ++i) {
            uint32_t r = 0;
            for (unsigned b = 0; b < bits; ++b)
                if (i & (size_t(1) << b)) r |= 1u << (bits - 1 - b);
            p->rev[i] = r;
        }
    } else {
        p->tw.resize(n);
        for (size_t k = 0; k < n; ++k)
            p->tw[k] = std::complex<T>((T)std::cos(step * (double)k), (T)std::sin(step * (double)k));
    }
    cache[n] = p;
    local[n] = p;
    return p;
}

// Unnormalised in-place complex DFT (inverse = conjugate kernel, still unnormalised).
template <class T>
inline void fft_inplace(std::complex<T>* a, size_t n, bool inverse) {
    if (n <= 1) return;
    auto plan = fft_plan<T>(n);
    if (plan->pow2) {
        for (size_t i = 0; i < n; ++i) {
            size_t j = plan->rev[i];
            if (i < j) std::swap(a[i], a[j]);
        }
        for (size_t len = 2; len <= n; len <<= 1) {
            const size_t half = len >> 1, stride = n / len;
            for (size_t base = 0; base < n; base += len) {
                for (size_t k = 0; k < half; ++k) {
                    std::complex<T> w = plan->tw[k * stride];
                    if (inverse) w = std::conj(w);
                    const std::complex<T> u = a[base + k];
                    const std::complex<T> x = a[base + k + half];
                    // complex multiply spelled out (no library NaN-recovery path)
                    const std::complex<T> v(x.real() * w.real() - x.imag() * w.imag(),
                                            x.real() * w.imag() + x.imag() * w.real());
                    a[base + k] = std::complex<T>(u.real() + v.real(), u.imag() + v.imag());
                    a[base + k + half] = std::complex<T>(u.real() - v.real(), u.imag() - v.imag());
                }
            }
        }
    } else {
        std::vector<std::complex<T>> out(n);
        for (size_t k = 0; k < n; ++k) {
            double re = 0.0, im = 0.0;
            for (size_t j = 0; j < n; ++j) {
                std::complex<T> w = plan->tw[(j * k) % n];
                if (inverse) w = std::conj(w);
                re += (double)a[j].real() * (double)w.real() - (double)a[j].imag() * (double)w.imag();
                im += (double)a[j].real() * (double)w.imag() + (double)a[j].imag() * (double)w.real();
            }
            out[k] = std::complex<T>((T)re, (T)im);
        }
        for (size_t k = 0; k < n; ++k) a[k] = out[k];
    }
}

// rustfft processes a buffer of k*len as k independent chunks (used at spectrogram :342).
template <class T>
inline void fft_chunks(std::complex<T>* a, size_t total, size_t len, bool inverse) {
    if (len == 0) return;
    for (size_t off = 0; off + len <= total; off += len) fft_inplace(a + off, len, inverse);
}

// realfft RealToComplex: N real -> N/2+1 complex, unnormalised.
template <class T>
inline void rfft(const T* in, size_t n, std::complex<T>* out) {
    std::vector<std::complex<T>> buf(n);
    for (size_t i = 0; i < n; ++i) buf[i] = std::complex<T>(in[i], (T)0);
    fft_inplace(buf.data(), n, false);
    for (size_t k = 0; k <= n / 2; ++k) out[k] = buf[k];
}

// realfft ComplexToReal: N/2+1 complex -> N real, unnormalised.  realfft errors when the
// DC / Nyquist imaginary parts are non-zero; returns false in that case like `.is_err()`.
template <class T>
inline bool irfft(const std::complex<T>* in, size_t n, T* out) {
    if (n == 0) return true;
    bool ok = in[0].imag() == (T)0;
    if (n % 2 == 0 && in[n / 2].imag() != (T)0) ok = false;
    std::vector<std::complex<T>> buf(n);
    buf[0] = std::complex<T>(in[0].real(), (T)0);
    for (size_t k = 1; k <= n / 2; ++k) {
        buf[k] = in[k];
        if (k != n - k) buf[n - k] = std::conj(in[k]);
    }
    if (n % 2 == 0) buf[n / 2] = std::complex<T>(in[n / 2].real(), (T)0);
    fft_inplace(buf.data(), n, true);
    for (size_t i = 0; i < n; ++i) out[i] = buf[i].real();
    return ok;
}

}  // namespace omxo
