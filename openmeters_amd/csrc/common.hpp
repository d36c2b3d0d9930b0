// openmeters_amd — shared host-side plumbing for the HIP product library (libomx_hip.so).
// No DSP lives here: error reporting, device buffers, the per-call audio format, and the
// closed-form coefficient tables that the reference also computes once per config on the host.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/omx.h"

namespace omx {

// ---------------------------------------------------------------- errors
void set_last_error(const std::string& msg);
int device_ready();  // OMX_NONE when a gfx950 device is usable, else OMX_ERR_NO_DEVICE / OMX_ERR_BACKEND
int device_count();
int select_device(int index);

struct BackendError {
    int status;
};

#define OMX_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            ::omx::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                                  ":" + std::to_string(__LINE__) + ")");                                \
            throw ::omx::BackendError{OMX_ERR_BACKEND};                                                 \
        }                                                                                               \
    } while (0)

[[noreturn]] inline void unsupported(const std::string& what) {
    set_last_error("unsupported configuration for the HIP path: " + what);
    throw BackendError{OMX_ERR_UNSUPPORTED};
}

// Runs `body` and converts BackendError into a negative omx_status: nothing unwinds across the C ABI.
// Environment-variable A/B hooks exist in the tuning build only (make TUNING=1 -> libomx_hip_tuning.so): the product library reads no
// tuning variable.  (OMX_WAVEFORM_SINGLE and OMX_SPLAT_FORM stay: the test suite cross-checks equivalent kernel forms through them.)
inline const char* tuning_env(const char* name) {
#ifdef OMX_TUNING
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
void bind_thread_device();  // the calling host thread onto the device omx_set_device selected (hipSetDevice is per thread)
template <class F>
inline int guarded(F&& body) {
    try {
        bind_thread_device();
        return body();
    } catch (const BackendError& e) {
        return e.status;
    } catch (const std::exception& e) {
        set_last_error(std::string("exception: ") + e.what());
        return OMX_ERR_BACKEND;
    } catch (...) {
        set_last_error("unknown exception");
        return OMX_ERR_BACKEND;
    }
}

// ---------------------------------------------------------------- device memory
template <class T>
struct DeviceBuffer {
    T* ptr = nullptr;
    size_t count = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { release(); }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    // Grow-only; contents are NOT preserved.
    void reserve(size_t n) {
        if (n <= count) return;
        release();
        OMX_HIP(hipMalloc(reinterpret_cast<void**>(&ptr), std::max<size_t>(n, 1) * sizeof(T)));
        count = n;
    }
    // Blocking upload (setup paths only): the host vector may be a temporary.
    void upload(const std::vector<T>& host, hipStream_t stream) {
        reserve(host.size());
        if (!host.empty()) {
            OMX_HIP(hipMemcpyAsync(ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, stream));
            OMX_HIP(hipStreamSynchronize(stream));
        }
    }
};

// Pinned (page-locked) host memory: the device reads / writes it through PCIe without a copy command; what the device wrote is
// visible to the host after a stream synchronisation.  The single-stream handles stage their blocks and take their snapshots
// through it: one synchronisation per call instead of a blocking copy in each direction.
template <class T>
struct PinnedBuffer {
    T* ptr = nullptr;
    size_t count = 0;
    PinnedBuffer() = default;
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    ~PinnedBuffer() { release(); }
    void release() {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    void reserve(size_t n) {  // grow-only; contents are NOT preserved
        if (n <= count) return;
        release();
        OMX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ptr), std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault));
        count = n;
    }
};

// Output of a bank: device memory, or — for the single-stream handles, while it stays small — pinned host memory
template <class T>
struct OutBuffer {
    T* ptr = nullptr;
    size_t count = 0;
    bool pinned = false;
    OutBuffer() = default;
    OutBuffer(const OutBuffer&) = delete;
    OutBuffer& operator=(const OutBuffer&) = delete;
    ~OutBuffer() { release(); }
    void release() {
        if (ptr) (void)(pinned ? hipHostFree(ptr) : hipFree(ptr));
        ptr = nullptr;
        count = 0;
    }
    void reserve(size_t n, bool want_pinned = false) {
        if (n <= count && want_pinned == pinned) return;
        release();
        pinned = want_pinned;
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        if (pinned) OMX_HIP(hipHostMalloc(reinterpret_cast<void**>(&ptr), bytes, hipHostMallocDefault));
        else OMX_HIP(hipMalloc(reinterpret_cast<void**>(&ptr), bytes));
        count = n;
    }
};

// Host PCM handed to a bank ("pcm_on_device = 0"): a small block goes through pinned memory that the kernels read directly over
// PCIe (no copy command, no second synchronisation); a large one by one bulk copy into device memory.
struct HostStage {
    PinnedBuffer<float> pinned;
    DeviceBuffer<float> device;
    const float* stage(const float* host, size_t n, hipStream_t stream) {
        if (n * sizeof(float) <= (size_t(256) << 10)) {
            OMX_HIP(hipStreamSynchronize(stream));  // the previous call's kernels may still be reading the buffer
            pinned.reserve(n);
            std::memcpy(pinned.ptr, host, n * sizeof(float));
            return pinned.ptr;
        }
        device.reserve(n);
        OMX_HIP(hipMemcpyAsync(device.ptr, host, n * sizeof(float), hipMemcpyHostToDevice, stream));
        OMX_HIP(hipStreamSynchronize(stream));  // the caller's buffer is borrowed for the call only (include/omx.h)
        return device.ptr;
    }
};

// A bank output handed to the caller: straight from pinned memory once the stream is idle, else one device-to-host copy.
inline void copy_out(void* dst, const void* src, size_t bytes, bool pinned, hipStream_t stream) {
    if (pinned) {
        OMX_HIP(hipStreamSynchronize(stream));  // ~1 us when the kernels that wrote `src` have already finished
        std::memcpy(dst, src, bytes);
        return;
    }
    OMX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
    OMX_HIP(hipStreamSynchronize(stream));
}

// The per-call host arrays of a ragged bank call (per-stream counts, reset flags) on their way to the device: two pinned staging sets
// used alternately, each guarded by an event recorded behind its copies — a call waits only if the copies of the call BEFORE the
// previous one are still in flight (practically never), instead of synchronising the stream on every call.
// a device array somebody else owns (RaggedStaging's blob)
template <class T>
struct DeviceView {
    T* ptr = nullptr;
};
struct RaggedStaging {
    PinnedBuffer<uint8_t> host[2];
    DeviceBuffer<uint8_t> dev;  // [counts: n u32][mask: n u8, padded to 4][lengths: n u32]
    hipEvent_t done[2] = {nullptr, nullptr};
    int next = 0;
    RaggedStaging() = default;
    RaggedStaging(const RaggedStaging&) = delete;
    RaggedStaging& operator=(const RaggedStaging&) = delete;
    ~RaggedStaging() {
        for (auto& e : done)
            if (e) (void)hipEventDestroy(e);
    }
    // counts_in[n], mask_in[n] (nullptr = all zero) and — chunk calls: one block of lengths_in[s] frames per stream — lengths_in[n] go
    // to the device as ONE copy on `stream` (round 6: three copies and their stream latency per bank and call before); the views
    // point into the staging's own device blob afterwards.  The caller's arrays are free again when this returns.
    void upload(const uint32_t* counts_in, const uint8_t* mask_in, uint32_t n, DeviceView<uint32_t>& d_counts, DeviceView<uint8_t>& d_mask,
                hipStream_t stream, const uint32_t* lengths_in = nullptr, DeviceView<uint32_t>* d_lengths = nullptr) {
        const int b = next;
        next ^= 1;
        if (done[b]) OMX_HIP(hipEventSynchronize(done[b]));
        else OMX_HIP(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
        const size_t mask_at = (size_t)n * 4, lengths_at = mask_at + ((size_t)n + 3) / 4 * 4, all = lengths_at + (size_t)n * 4;
        const bool with_lengths = lengths_in && d_lengths;
        host[b].reserve(all);
        dev.reserve(all);
        std::memcpy(host[b].ptr, counts_in, (size_t)n * 4);
        if (mask_in) std::memcpy(host[b].ptr + mask_at, mask_in, n);
        else std::memset(host[b].ptr + mask_at, 0, n);
        if (with_lengths) std::memcpy(host[b].ptr + lengths_at, lengths_in, (size_t)n * 4);
        OMX_HIP(hipMemcpyAsync(dev.ptr, host[b].ptr, with_lengths ? all : lengths_at, hipMemcpyHostToDevice, stream));
        OMX_HIP(hipEventRecord(done[b], stream));
        d_counts.ptr = reinterpret_cast<uint32_t*>(dev.ptr);
        d_mask.ptr = dev.ptr + mask_at;
        if (d_lengths) d_lengths->ptr = with_lengths ? reinterpret_cast<uint32_t*>(dev.ptr + lengths_at) : nullptr;
    }
};

// A small per-call host structure (a plan: cut lists, column tables) on its way to the device, same double-buffering as above
struct BlobStaging {
    PinnedBuffer<uint8_t> buf[2];
    hipEvent_t done[2] = {nullptr, nullptr};
    int next = 0;
    BlobStaging() = default;
    BlobStaging(const BlobStaging&) = delete;
    BlobStaging& operator=(const BlobStaging&) = delete;
    ~BlobStaging() {
        for (auto& e : done)
            if (e) (void)hipEventDestroy(e);
    }
    void upload(const void* src, size_t bytes, void* d_dst, hipStream_t stream) {
        const int b = next;
        next ^= 1;
        if (done[b]) OMX_HIP(hipEventSynchronize(done[b]));
        else OMX_HIP(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
        buf[b].reserve(bytes);
        std::memcpy(buf[b].ptr, src, bytes);
        OMX_HIP(hipMemcpyAsync(d_dst, buf[b].ptr, bytes, hipMemcpyHostToDevice, stream));
        OMX_HIP(hipEventRecord(done[b], stream));
    }
};

struct EventTimer {  // HIP-event timing of one kernel family on its launch stream
    hipEvent_t start = nullptr, stop = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    bool enabled = false;
    ~EventTimer() {
        if (start) (void)hipEventDestroy(start);
        if (stop) (void)hipEventDestroy(stop);
        for (auto& p : pending) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    }
    // A begin() whose end() never comes (the launch between them threw) leaves one pair behind: it is destroyed by the next begin().
    // The tally is bounded: beyond kMaxPending launches without a collect() the oldest pair is dropped.
    static constexpr size_t kMaxPending = 4096;
    void begin(hipStream_t s) {
        if (!enabled) return;
        if (start) (void)hipEventDestroy(start);
        if (stop) (void)hipEventDestroy(stop);
        start = stop = nullptr;
        OMX_HIP(hipEventCreate(&start));
        OMX_HIP(hipEventCreate(&stop));
        OMX_HIP(hipEventRecord(start, s));
    }
    void end(hipStream_t s) {
        if (!enabled || !start) return;
        OMX_HIP(hipEventRecord(stop, s));
        if (pending.size() >= kMaxPending) {
            (void)hipEventDestroy(pending.front().first);
            (void)hipEventDestroy(pending.front().second);
            pending.erase(pending.begin());
        }
        pending.emplace_back(start, stop);
        start = stop = nullptr;
    }
    // average ms over the recorded launches; clears the tally
    double collect(uint64_t* launches) {
        double total = 0.0;
        uint64_t n = 0;
        for (auto& p : pending) {
            OMX_HIP(hipEventSynchronize(p.second));
            float ms = 0.0f;
            OMX_HIP(hipEventElapsedTime(&ms, p.first, p.second));
            total += ms;
            ++n;
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        pending.clear();
        if (launches) *launches = n;
        return n ? total / (double)n : 0.0;
    }
};

// ---------------------------------------------------------------- small numeric helpers (host)
constexpr float kDefaultSampleRate = 48000.0f;  // reference src/util/audio/rate.rs:6
constexpr float kMaxSampleRate = 768000.0f;     // rate.rs:7
constexpr float kDbFloor = -140.0f;             // level.rs:4
constexpr float kLnToDb = 4.3429448f;           // level.rs:5
constexpr float kTau = 6.28318530717958647692f;
constexpr float kFrac1Sqrt2 = 0.707106781186547524400844362104849039f;

inline float sanitize_sample_rate(float r) {  // rate.rs:9-13
    float v = (std::isfinite(r) && r > 0.0f) ? r : kDefaultSampleRate;
    return v < 1.0f ? 1.0f : (v > kMaxSampleRate ? kMaxSampleRate : v);
}
inline size_t f2usize(double x) {  // Rust `as usize`
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551615.0) return SIZE_MAX;
    return (size_t)x;
}
inline bool is_pow2(size_t n) { return n >= 1 && (n & (n - 1)) == 0; }
inline unsigned log2_exact(size_t n) {
    unsigned l = 0;
    while ((size_t(1) << l) < n) ++l;
    return l;
}
inline size_t next_pow2(size_t v) {
    size_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
inline float db_to_power_host(float db) {  // level.rs:36-39
    const float DB_TO_LOG2 = 0.1f * 3.32192809488736234787f;
    return std::exp2(db * DB_TO_LOG2);
}

// ---------------------------------------------------------------- audio format of one call
// Host mirror of AudioBlock::with_positions' derived data (reference src/dsp.rs:117-213): the 8x2
// stereo down-mix matrix.  `stereo_channels` trimming (dsp.rs:197-204) is a CPU-side shortcut with no
// observable effect (trimmed channels are all-zero bits and contribute +0.0), so the device folds
// over every channel.
struct AudioFormatArgs {
    uint32_t channels;       // 1..=8
    float m[OMX_MAX_CHANNELS][2];
};
AudioFormatArgs make_format(uint32_t channels, const uint8_t positions[OMX_MAX_CHANNELS]);
void positions_fallback(uint32_t channels, uint8_t out[OMX_MAX_CHANNELS]);
void positions_normalize(uint32_t channels, const uint8_t in[OMX_MAX_CHANNELS], uint8_t out[OMX_MAX_CHANNELS]);

// window.rs:20-43 (periodic cosine-sum windows, f32) and :90-109 (bin normalisation)
std::vector<float> window_coefficients(uint32_t kind, size_t len);
std::vector<float> fft_bin_normalization(const std::vector<float>& window, size_t fft_size);
// exp(-2*pi*i*k/n) for k < count, computed in f64 and rounded to f32 (interleaved re,im)
std::vector<float> twiddle_table(size_t n, size_t count);
// Bluestein tables of an n-point DFT (n not a power of two), as interleaved (re, im) floats; built in double precision
struct BluesteinHostTables {
    size_t m = 0;
    std::vector<float> chirp, bf, tw_m;
};
BluesteinHostTables bluestein_tables(size_t n);
// the reference's spectral derivative window (spectrogram/processor.rs:569-599) for any window length, evaluated in double precision
std::vector<float> derivative_window_host(const std::vector<float>& window);

}  // namespace omx
