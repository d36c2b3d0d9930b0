// Host-side plumbing of libomx_hip.so (see common.hpp).  The channel-layout and window helpers
// mirror what the reference computes on the host once per format / config:
//   reference src/dsp.rs:36-76 (ChannelPosition::fallback / normalize), :117-176 (stereo matrix),
//   src/util/audio/window.rs:20-43, :90-109.
#include "common.hpp"

#include <atomic>
#include <complex>

namespace omx {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const std::string& last_error() { return g_last_error; }

static std::atomic<int> g_device_ready{2};    // 2 = unknown
static std::atomic<int> g_selected_device{-1};  // omx_set_device's choice, process-wide (-1: the HIP default, device 0)
static thread_local int tl_bound_device = -1;

// hipSetDevice is per host thread: a capture thread and a UI thread of one host must both land on the device omx_set_device chose.
// Every C-ABI entry passes through here (guarded(), device_ready()); one relaxed load when nothing changed.
void bind_thread_device() {
    const int want = g_selected_device.load(std::memory_order_relaxed);
    if (want >= 0 && tl_bound_device != want) {
        if (hipSetDevice(want) == hipSuccess) tl_bound_device = want;
        else (void)hipGetLastError();
    }
}

// omx_device_count / omx_set_device: a host without the HIP headers (the Rust service, one process per GPU) picks its device here,
// before it creates handles; handles live on the device that was current when they were created
int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
int select_device(int index) {
    const int n = device_count();
    if (index < 0 || index >= n) {
        set_last_error("omx_set_device: index out of range (" + std::to_string(n) + " HIP devices visible)");
        return n == 0 ? OMX_ERR_NO_DEVICE : OMX_ERR_INVALID;
    }
    if (hipSetDevice(index) != hipSuccess) {
        set_last_error("hipSetDevice failed");
        return OMX_ERR_BACKEND;
    }
    tl_bound_device = index;
    g_selected_device.store(index);
    g_device_ready.store(2);  // the architecture check runs again for the new device
    return device_ready();
}

int device_ready() {
    bind_thread_device();
    struct Cached {  // (`return cached = x` below stores and returns)
        int operator=(int v) { g_device_ready.store(v); return v; }
    } cached;
    const int known = g_device_ready.load();
    if (known != 2) return known;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_last_error("no HIP device visible (libomx_hip.so has no CPU fallback)");
        (void)hipGetLastError();
        return cached = OMX_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        set_last_error("hipGetDeviceProperties failed");
        return cached = OMX_ERR_BACKEND;
    }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_last_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        return cached = OMX_ERR_NO_DEVICE;
    }
    return cached = OMX_NONE;
}

void positions_fallback(uint32_t channels, uint8_t out[OMX_MAX_CHANNELS]) {
    static const uint8_t surround[8] = {OMX_POS_FRONT_LEFT, OMX_POS_FRONT_RIGHT, OMX_POS_FRONT_CENTER,
                                        OMX_POS_LOW_FREQUENCY, OMX_POS_REAR_LEFT, OMX_POS_REAR_RIGHT,
                                        OMX_POS_SIDE_LEFT, OMX_POS_SIDE_RIGHT};
    const uint32_t c = std::min<uint32_t>(channels, OMX_MAX_CHANNELS);
    for (int i = 0; i < OMX_MAX_CHANNELS; ++i) out[i] = (uint32_t)i < c ? surround[i] : (uint8_t)OMX_POS_UNKNOWN;
    if (c == 1) out[0] = OMX_POS_MONO;
    if (c == 4) { out[2] = OMX_POS_REAR_LEFT; out[3] = OMX_POS_REAR_RIGHT; }
    if (c == 5) { out[3] = OMX_POS_REAR_LEFT; out[4] = OMX_POS_REAR_RIGHT; }
}

void positions_normalize(uint32_t channels, const uint8_t in[OMX_MAX_CHANNELS], uint8_t out[OMX_MAX_CHANNELS]) {
    const uint32_t c = std::min<uint32_t>(channels, OMX_MAX_CHANNELS);
    uint8_t p[OMX_MAX_CHANNELS];
    for (uint32_t i = 0; i < OMX_MAX_CHANNELS; ++i) p[i] = i < c ? in[i] : (uint8_t)OMX_POS_UNKNOWN;
    for (uint32_t i = 0; i < c; ++i) {
        bool dup = false;
        for (uint32_t j = 0; j < i; ++j) dup = dup || p[j] == p[i];
        if (dup) p[i] = OMX_POS_UNKNOWN;
    }
    uint8_t fb[OMX_MAX_CHANNELS], sur[OMX_MAX_CHANNELS];
    positions_fallback(c, fb);
    positions_fallback(OMX_MAX_CHANNELS, sur);
    auto unused = [&](uint8_t cand) {
        if (cand == OMX_POS_UNKNOWN) return false;
        for (uint32_t j = 0; j < c; ++j)
            if (p[j] == cand) return false;
        return true;
    };
    for (uint32_t i = 0; i < c; ++i) {
        if (p[i] != OMX_POS_UNKNOWN) continue;
        bool placed = false;
        if (unused(fb[i])) { p[i] = fb[i]; placed = true; }
        for (int k = 0; !placed && k < OMX_MAX_CHANNELS; ++k)
            if (unused(fb[k])) { p[i] = fb[k]; placed = true; }
        for (int k = 0; !placed && k < OMX_MAX_CHANNELS; ++k)
            if (unused(sur[k])) { p[i] = sur[k]; placed = true; }
        for (int k = 0; !placed && k < OMX_MAX_CHANNELS; ++k)
            if (unused((uint8_t)(OMX_POS_AUX0 + k))) { p[i] = (uint8_t)(OMX_POS_AUX0 + k); placed = true; }
    }
    for (int i = 0; i < OMX_MAX_CHANNELS; ++i) out[i] = p[i];
}

AudioFormatArgs make_format(uint32_t channels_in, const uint8_t positions[OMX_MAX_CHANNELS]) {
    AudioFormatArgs f;
    const uint32_t channels = std::min<uint32_t>(std::max<uint32_t>(channels_in, 1), OMX_MAX_CHANNELS);
    f.channels = channels;
    const float s = kFrac1Sqrt2;
    for (int i = 0; i < OMX_MAX_CHANNELS; ++i) f.m[i][0] = f.m[i][1] = 0.0f;
    for (uint32_t i = 0; i < channels; ++i) {
        switch (positions[i]) {
            case OMX_POS_FRONT_LEFT: f.m[i][0] = 1.0f; break;
            case OMX_POS_FRONT_RIGHT: f.m[i][1] = 1.0f; break;
            case OMX_POS_FRONT_CENTER: f.m[i][0] = s; f.m[i][1] = s; break;
            case OMX_POS_REAR_LEFT:
            case OMX_POS_SIDE_LEFT: f.m[i][0] = s; break;
            case OMX_POS_REAR_RIGHT:
            case OMX_POS_SIDE_RIGHT: f.m[i][1] = s; break;
            case OMX_POS_MONO: f.m[i][0] = 1.0f; f.m[i][1] = 1.0f; break;
            default: break;
        }
    }
    bool l = false, r = false;
    for (uint32_t i = 0; i < channels; ++i) {
        l = l || f.m[i][0] != 0.0f;
        r = r || f.m[i][1] != 0.0f;
    }
    if (!l && !r) {  // dsp.rs:117-133 stereo_indices
        auto find = [&](uint8_t want) {
            for (uint32_t i = 0; i < channels; ++i)
                if (positions[i] == want) return (int)i;
            return -1;
        };
        const int explicit_right = find(OMX_POS_FRONT_RIGHT);
        int left = find(OMX_POS_FRONT_LEFT);
        if (left < 0) left = find(OMX_POS_MONO);
        for (uint32_t i = 0; left < 0 && i < channels; ++i)
            if ((int)i != explicit_right) left = (int)i;
        if (left < 0) left = 0;
        int right = (explicit_right >= 0 && explicit_right != left) ? explicit_right : -1;
        for (uint32_t i = 0; right < 0 && i < channels; ++i)
            if ((int)i != left) right = (int)i;
        if (right < 0) right = left;
        f.m[left][0] = 1.0f;
        f.m[right][1] = 1.0f;
    } else if (!l) {
        for (int i = 0; i < OMX_MAX_CHANNELS; ++i) f.m[i][0] = f.m[i][1];
    } else if (!r) {
        for (int i = 0; i < OMX_MAX_CHANNELS; ++i) f.m[i][1] = f.m[i][0];
    }
    return f;
}

std::vector<float> window_coefficients(uint32_t kind, size_t len) {
    if (len <= 1) return std::vector<float>(len, 1.0f);
    const float hann[] = {0.5f, -0.5f};
    const float hamming[] = {25.0f / 46.0f, -21.0f / 46.0f};
    const float blackman[] = {0.42f, -0.5f, 0.08f};
    const float bh[] = {0.35875f, -0.48829f, 0.14128f, -0.01168f};
    const float* c = nullptr;
    size_t nc = 0;
    switch (kind) {
        case OMX_WINDOW_HANN: c = hann; nc = 2; break;
        case OMX_WINDOW_HAMMING: c = hamming; nc = 2; break;
        case OMX_WINDOW_BLACKMAN: c = blackman; nc = 3; break;
        case OMX_WINDOW_BLACKMAN_HARRIS: c = bh; nc = 4; break;
        default: return std::vector<float>(len, 1.0f);
    }
    const float step = kTau / (float)len;
    std::vector<float> w(len);
    for (size_t n = 0; n < len; ++n) {
        const float phi = (float)n * step;
        float sum = 0.0f;
        for (size_t k = 0; k < nc; ++k) sum = sum + c[k] * std::cos(phi * (float)k);
        w[n] = sum;
    }
    return w;
}

std::vector<float> fft_bin_normalization(const std::vector<float>& window, size_t fft_size) {
    const size_t bins = fft_size / 2 + 1;
    float window_sum = -0.0f;  // Rust's float Sum identity
    for (float v : window) window_sum = window_sum + v;
    float inv_sum;
    if (std::fabs(window_sum) > std::numeric_limits<float>::epsilon()) inv_sum = 1.0f / window_sum;
    else if (fft_size > 0) inv_sum = 1.0f / (float)fft_size;
    else inv_sum = 0.0f;
    const float dc = inv_sum * inv_sum, ac = 4.0f * dc;
    std::vector<float> norms(bins, ac);
    norms[0] = dc;
    if (fft_size % 2 == 0 && bins > 1) norms[bins - 1] = dc;
    return norms;
}

std::vector<float> twiddle_table(size_t n, size_t count) {
    std::vector<float> t(2 * std::max<size_t>(count, 1), 0.0f);
    t[0] = 1.0f;
    const double step = -2.0 * M_PI / (double)(n ? n : 1);
    for (size_t k = 0; k < count; ++k) {
        t[2 * k] = (float)std::cos(step * (double)k);
        t[2 * k + 1] = (float)std::sin(step * (double)k);
    }
    return t;
}

namespace {
void fft_radix2_host(std::vector<std::complex<double>>& a) {  // forward, in place, power-of-two length
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const std::complex<double> w(std::cos(ang * (double)k), std::sin(ang * (double)k));
                const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}
}  // namespace

BluesteinHostTables bluestein_tables(size_t n) {
    BluesteinHostTables t;
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    t.m = m;
    t.chirp.resize(2 * n);
    std::vector<std::complex<double>> b(m, std::complex<double>(0.0, 0.0));
    for (size_t k = 0; k < n; ++k) {
        const uint64_t k2 = ((uint64_t)k * (uint64_t)k) % (2 * (uint64_t)n);  // the phase pi k^2 / n is periodic in k^2 mod 2n
        const double ang = M_PI * (double)k2 / (double)n;
        t.chirp[2 * k] = (float)std::cos(ang);
        t.chirp[2 * k + 1] = (float)-std::sin(ang);
        b[k] = std::complex<double>(std::cos(ang), std::sin(ang));
        if (k) b[m - k] = b[k];
    }
    fft_radix2_host(b);
    t.bf.resize(2 * m);
    for (size_t k = 0; k < m; ++k) {
        t.bf[2 * k] = (float)b[k].real();
        t.bf[2 * k + 1] = (float)b[k].imag();
    }
    t.tw_m = twiddle_table(m, m / 2);
    return t;
}

std::vector<float> derivative_window_host(const std::vector<float>& window) {
    const size_t n = window.size();
    std::vector<float> out(n, 0.0f);
    if (n <= 1) return out;
    // X[k] = sum_j w[j] e^{-2 pi i j k / n};  D[k] = i omega_k X[k] (0 at DC and, for even n, at Nyquist);  w'[j] = Re IDFT(D)[j] / n
    // (one n-entry table of e^{-2 pi i m / n}, indexed by (j k) mod n: the two O(n^2) sums are multiply-adds only — at the 65 536
    // samples this path accepts, per-term cos / sin would be minutes of libm calls)
    std::vector<std::complex<double>> tw(n);
    for (size_t m = 0; m < n; ++m) {
        const double ang = -2.0 * M_PI * (double)m / (double)n;
        tw[m] = std::complex<double>(std::cos(ang), std::sin(ang));
    }
    std::vector<std::complex<double>> X(n);
    for (size_t k = 0; k < n; ++k) {
        std::complex<double> acc(0.0, 0.0);
        size_t m = 0;  // (j k) mod n, advanced by k per term
        for (size_t j = 0; j < n; ++j) {
            acc += (double)window[j] * tw[m];
            m += k;
            if (m >= n) m -= n;
        }
        X[k] = acc;
    }
    const size_t half = n / 2;
    const double scale = 2.0 * M_PI / (double)n;
    for (size_t k = 0; k < n; ++k) {
        const double omega = scale * ((double)k - (k > half ? (double)n : 0.0));
        X[k] = (k == 0 || (n % 2 == 0 && k == half)) ? std::complex<double>(0.0, 0.0) : std::complex<double>(-omega * X[k].imag(), omega * X[k].real());
    }
    for (size_t j = 0; j < n; ++j) {
        double acc = 0.0;
        size_t m = 0;
        for (size_t k = 0; k < n; ++k) {
            const std::complex<double> w = std::conj(tw[m]);  // e^{+2 pi i j k / n}
            acc += X[k].real() * w.real() - X[k].imag() * w.imag();
            m += j;
            if (m >= n) m -= n;
        }
        out[j] = (float)(acc / (double)n);
    }
    return out;
}

}  // namespace omx
