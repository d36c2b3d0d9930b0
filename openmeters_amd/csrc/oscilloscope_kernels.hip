// K6 nsdf_period + K7 template_trigger (+ zero-crossing trigger and trace resampling) — one 256-thread
// workgroup per stream, blocks processed in timeline order.  Single-pass form: the path of the sample rates whose
// autocorrelation is not an 8192-point transform (below 27.3 kHz, above 54.6 kHz); 44.1 / 48 kHz run scope_fast_kernels.hip.
// reference src/visuals/oscilloscope/processor.rs:85-182 (PeriodEstimator), :184-263 (helpers),
// :272-528 (StableTrigger), :530-551, :769-803 (zero crossing, downsample), :611-750 (process_block).
//
// Scalar control decisions (stabilise, lock bookkeeping, coarse-to-fine argmax replay) run on thread 0
// and are broadcast; every map / reduction / FFT is spread over the workgroup.  Reductions use a tree
// order, so sums differ from the reference's sequential f32 sums at the 1e-7 level (see tests for the
// decision-level parity this implies).
#include "scope_device.hpp"

#include "fft_device.hpp"

namespace omx {

namespace {

struct Shared {
    float redn[4][6];  // block_sum_n: [wave][component]
    float redf[8];
    unsigned long long redu[8];
    float f[8];
    uint32_t u[8];
    int i[4];
};

__device__ float block_sum(float v, Shared& sh) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.redf[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh.redf[0] + sh.redf[1]) + sh.redf[2]) + sh.redf[3];
}
__device__ float block_max(float v, Shared& sh) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.redf[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh.redf[0], sh.redf[1]), fmaxf(sh.redf[2], sh.redf[3]));
}
// K sums in one barrier pair; every component is reduced exactly like block_sum (same shuffle tree, same wave order), so
// fusing several block_sum calls into one changes no bit.  MAX_LAST: the last component is a maximum instead of a sum.
template <int K, bool MAX_LAST = false>
__device__ void block_sum_n(float (&v)[K], Shared& sh) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float o = __shfl_xor(v[k], off);
            v[k] = (MAX_LAST && k == K - 1) ? fmaxf(v[k], o) : v[k] + o;
        }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sh.redn[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (MAX_LAST && k == K - 1) v[k] = fmaxf(fmaxf(sh.redn[0][k], sh.redn[1][k]), fmaxf(sh.redn[2][k], sh.redn[3][k]));
        else v[k] = ((sh.redn[0][k] + sh.redn[1][k]) + sh.redn[2][k]) + sh.redn[3][k];
    }
}
__device__ unsigned long long block_max_u64(unsigned long long v, Shared& sh) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.redu[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long m = sh.redu[0];
    for (int w = 1; w < 4; ++w) m = sh.redu[w] > m ? sh.redu[w] : m;
    return m;
}
__device__ uint32_t block_min_u32(uint32_t v, Shared& sh) { return ~(uint32_t)block_max_u64((unsigned long long)(~v), sh); }
struct Scratch {
    float* work;        // [max_kernel + max_search]
    float* candidate;   // [max_kernel]
    float* retuned;     // [max_kernel]
    float* nsdf;        // [max_period + 2]
    float* scores;      // [max_search + 2]
    float* energy;      // [probe_frames + 1]
    float* partials;    // [256] block-scan partials of estimate_period
};

// tuning aid (ScopeArgs::phase_timing): cycles thread 0 spends between marks, summed over workgroups and blocks
__device__ unsigned long long g_scope_phase_cycles[SCOPE_PHASES];
struct PhaseClock {
    long long t;
    bool on;
    __device__ __forceinline__ void start(bool enabled) {
        on = enabled && threadIdx.x == 0;
        if (on) t = clock64();
    }
    __device__ __forceinline__ void mark(int i) {
        if (on) {
            const long long now = clock64();
            atomicAdd(&g_scope_phase_cycles[i], (unsigned long long)(now - t));
            t = now;
        }
    }
};

// ---------------------------------------------------------------- PeriodEstimator::estimate_period (:93-181)
using ScopeTwiddles = TwiddleSource<true, true>;  // pass-2 table in LDS, pass-3 twiddles in VGPRs, loaded once per kernel

__device__ Estimate estimate_period(const View& x, float rate, float& last_peak, const ScopeArgs& a, v2f* fft, Scratch& sc,
                                    Shared& sh, PhaseClock& pc, const ScopeTwiddles& tw) {
    const unsigned tid = threadIdx.x;
    Estimate none{0, 0.0f, 0.0f};
    last_peak = 0.0f;
    const uint32_t n = x.n;
    if (n < 3) return none;
    // Steady state (8192-point autocorrelation, transform buffer in LDS): the probe is read from the trace ring ONCE, into the
    // still unused transform buffer; mean, peak, prefix energy and the transform's input all come from that LDS copy (four more
    // passes over the global ring before, each exposing an L2 round trip to the one workgroup a CU holds).
    const bool staged = a.tw4096 != nullptr && !a.fft_global && n <= 2u * FFT4096_LDS;
    float* stage = reinterpret_cast<float*>(fft);
    float part = 0.0f;
    if (staged) {
        for (uint32_t i = tid; i < n; i += 256) {
            const float v = x.at(i);
            stage[i] = v;
            part += v;
        }
    } else {
        for (uint32_t i = tid; i < n; i += 256) part += x.at(i);
    }
    const float mean = block_sum(part, sh) / (float)n;  // its barriers also publish `stage`
    auto xs = [&](uint32_t i) { return staged ? stage[i] : x.at(i); };
    float pk = 0.0f;
    for (uint32_t i = tid; i < n; i += 256) pk = fmaxf(pk, fabsf(xs(i) - mean));
    last_peak = block_max(pk, sh);
    if (last_peak < MIN_SIGNAL_PEAK) return none;
    const uint32_t min_period = f2u(fmaxf(roundf(rate / MAX_HZ), 2.0f));
    const uint32_t max_period = min(f2u(roundf(rate / MIN_HZ)), n / 2);
    if (max_period <= min_period + 1) return none;

    // compute_periodicity (:133-181)
    const uint32_t max_lag = max_period;
    uint32_t fft_size = 1, logn = 0;
    while (fft_size < n + max_lag) {
        fft_size <<= 1;
        ++logn;
    }
    const uint32_t tw_step = a.fft_size / fft_size;  // a.fft_size is the largest size this config can need
    const uint32_t chunk = (n + 255) / 256;
    const bool fast = fft_size == 8192 && a.tw4096 != nullptr;  // steady state at 44.1 / 48 kHz
    {  // centred copy + prefix energy: per-thread chunk sums, then a wave-level scan of the 256 chunk sums
        const uint32_t lo = min(tid * chunk, n), hi = min(lo + chunk, n);
        float local = 0.0f;
        if (staged && !fast) __syncthreads();  // (never in practice) the radix-2 path overwrites the staging area below
        float cached[24];                      // chunk <= 24 whenever the probe is staged (n <= 6144)
        const bool keep = staged && chunk <= 24;
        if (keep) {
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                const uint32_t i = lo + (uint32_t)q;
                cached[q] = i < hi ? stage[i] - mean : 0.0f;
                local = i < hi ? cached[q] * cached[q] + local : local;
            }
        } else {
            for (uint32_t i = lo; i < hi; ++i) {
                const float c = x.at(i) - mean;
                if (!fast) fft[i] = v2f{c, 0.0f};
                local = c * c + local;
            }
        }
        if (!fast) {
            if (keep) {
                __syncthreads();
                for (uint32_t i = lo; i < hi; ++i) fft[i] = v2f{x.at(i) - mean, 0.0f};
            }
            for (uint32_t i = n + tid; i < fft_size; i += 256) fft[i] = v2f{0.0f, 0.0f};
        }
        // exclusive scan of the chunk sums (the first form scanned the 256 partials serially on thread 0: 256 dependent
        // LDS round trips, a tenth of the whole block time)
        const unsigned lane = tid & 63u, wave = tid >> 6;
        float incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float up = __shfl_up(incl, off);
            if ((int)lane >= off) incl += up;
        }
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 0.0f;
        __syncthreads();
        if (lane == 63) sh.redf[wave] = incl;
        if (tid == 0) sc.energy[0] = 0.0f;
        __syncthreads();
        float run = 0.0f;
        for (unsigned w = 0; w < wave; ++w) run += sh.redf[w];
        run += excl;
        if (keep) {
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                const uint32_t i = lo + (uint32_t)q;
                if (i < hi) {
                    run = cached[q] * cached[q] + run;
                    sc.energy[i + 1] = run;
                }
            }
        } else {
            for (uint32_t i = lo; i < hi; ++i) {
                const float c = x.at(i) - mean;
                run = c * c + run;
                sc.energy[i + 1] = run;
            }
        }
    }
    pc.mark(1);  // mean / peak / centred copy / prefix energy
    const float norm = 1.0f / (float)fft_size;
    if (fast) {
        // Autocorrelation of the zero-padded real probe through two 4096-point register/LDS transforms instead of two
        // 8192-point complex radix-2 transforms in LDS (:147-160 computes FFT_8192(x + 0i), |.|^2, IFFT_8192, real part):
        //   z[m] = x[2m] + i x[2m+1];  Z = FFT_4096(z);  E, O = even / odd sample spectra;  X[k] = E + w^k O, X[k+N] = E - w^k O
        //   P = |X|^2 (real, P[2N-k] = P[k]);  acf[2m] + i acf[2m+1] = IFFT_4096( (P[k] + P[k+N]) + i (P[k] - P[k+N]) conj(w^k) )
        constexpr uint32_t N = 4096;
        const int j = (int)tid;
        v2f v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t m = (uint32_t)(j + 256 * t);
            float re, im;
            if (staged) {  // one 8-byte LDS read (the staging area covers 2m + 1 <= 8191; what lies beyond n is selected away)
                const v2f pair = *reinterpret_cast<const v2f*>(stage + 2u * m);
                re = 2u * m < n ? pair.x - mean : 0.0f;
                im = 2u * m + 1u < n ? pair.y - mean : 0.0f;
            } else {
                re = 2u * m < n ? x.at(2u * m) - mean : 0.0f;
                im = 2u * m + 1u < n ? x.at(2u * m + 1u) - mean : 0.0f;
            }
            v[t] = v2f{re, im};
        }
        if (staged) __syncthreads();  // every thread has taken its samples out of the staging area the first pass overwrites
        fft4096t<false, false>(v, fft, fft, j, tw);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) fft[pad16(j + 256 * t)] = v[t];
        __syncthreads();
        v2f y[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t k = (uint32_t)(j + 256 * t);
            const v2f z = v[t];
            const v2f zr = fft[pad16((int)((N - k) & (N - 1)))];
            const v2f e{(z.x + zr.x) * 0.5f, (z.y - zr.y) * 0.5f};  // (Z + conj Zr) / 2
            const v2f o{(z.y + zr.y) * 0.5f, (zr.x - z.x) * 0.5f};  // (Z - conj Zr) / (2i)
            const v2f w = a.tw_fft[k];                               // exp(-2 pi i k / 8192)
            const v2f wo{o.x * w.x - o.y * w.y, o.x * w.y + o.y * w.x};
            const v2f xp{e.x + wo.x, e.y + wo.y}, xm{e.x - wo.x, e.y - wo.y};
            const float p0 = xp.x * xp.x + xp.y * xp.y, p1 = xm.x * xm.x + xm.y * xm.y;  // P[k], P[k + N]
            const float sum = p0 + p1, dif = p0 - p1;
            y[t] = v2f{sum + dif * w.y, dif * w.x};  // (P[k] + P[k+N]) + i (P[k] - P[k+N]) conj(w^k)
        }
        __syncthreads();
        fft4096t<true, false>(y, fft, fft, j, tw);  // y[t] = (acf[2m], acf[2m + 1]), m = j + 256 t
        const float total_energy_f = sc.energy[n];
        if (total_energy_f > F32_EPS) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint32_t m = (uint32_t)(j + 256 * t);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t tau = 2u * m + (uint32_t)h;
                    if (tau > max_lag) continue;
                    const float acf = h ? y[t].y : y[t].x;
                    const float left = sc.energy[n - tau], right = total_energy_f - sc.energy[tau];
                    const float denom = left + right;
                    sc.nsdf[tau] = denom > F32_EPS ? 2.0f * acf * norm / denom : 0.0f;
                }
            }
        }
        pc.mark(2);  // two FFTs (+ NSDF fill)
        if (total_energy_f <= F32_EPS) return none;
    } else {
        fft_radix2(fft, fft_size, logn, a.tw_fft, false, tid, 256, tw_step);
        for (uint32_t k = tid; k < fft_size; k += 256) {
            const v2f b = fft[k];
            fft[k] = v2f{b.x * b.x + b.y * b.y, 0.0f};
        }
        fft_radix2(fft, fft_size, logn, a.tw_fft, true, tid, 256, tw_step);
        pc.mark(2);  // two FFTs
        const float total_energy = sc.energy[n];
        if (total_energy <= F32_EPS) return none;
        for (uint32_t tau = tid; tau <= max_lag; tau += 256) {
            const float left = sc.energy[n - tau], right = total_energy - sc.energy[tau];
            const float denom = left + right;
            sc.nsdf[tau] = denom > F32_EPS ? 2.0f * fft[tau].x * norm / denom : 0.0f;
        }
    }
    __syncthreads();
    const float* nsdf = sc.nsdf;
    // first tau >= 1 with nsdf <= 0 (:110)
    uint32_t zc = 0xFFFFFFFFu;
    for (uint32_t tau = 1 + tid; tau <= max_period; tau += 256)
        if (nsdf[tau] <= 0.0f) { zc = tau; break; }
    zc = block_min_u32(zc, sh);
    if (zc == 0xFFFFFFFFu) return none;
    const uint32_t first_tau = max(min_period, zc);
    if (first_tau >= max_period) return none;
    auto is_candidate = [&](uint32_t tau) {
        return nsdf[tau] >= MIN_PERIODICITY && nsdf[tau] >= nsdf[tau - 1] && nsdf[tau] >= nsdf[tau + 1];
    };
    // max_by(total_cmp) keeps the LAST maximum (:119-121): order by (value, index)
    unsigned long long bestk = 0ull;
    for (uint32_t tau = first_tau + tid; tau < max_period; tau += 256)
        if (is_candidate(tau)) {
            const unsigned long long k = ((unsigned long long)total_order_key(nsdf[tau]) << 32) | tau;
            bestk = k > bestk ? k : bestk;
        }
    bestk = block_max_u64(bestk, sh);
    if (bestk == 0ull) return none;
    const uint32_t best = (uint32_t)(bestk & 0xFFFFFFFFull);
    const float cutoff = nsdf[best] * PEAK_CUTOFF;
    uint32_t peak = 0xFFFFFFFFu;
    for (uint32_t tau = first_tau + tid; tau <= best; tau += 256)
        if (is_candidate(tau) && nsdf[tau] >= cutoff) { peak = tau; break; }
    peak = block_min_u32(peak, sh);
    if (peak == 0xFFFFFFFFu) peak = best;
    Estimate e;
    e.some = 1;
    e.period = parabolic_refine(nsdf[peak - 1], nsdf[peak], nsdf[peak + 1], peak);
    e.confidence = rclamp(nsdf[peak], 0.0f, 1.0f);
    return e;
}

// ---------------------------------------------------------------- StableTrigger pieces
__device__ void correlation_stats(const float* y, uint32_t n, float& sum, float& squares, Shared& sh) {  // :206-208
    float s = 0.0f, q = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        s += y[i];
        q += y[i] * y[i];
    }
    float v[2] = {s, q};
    block_sum_n<2>(v, sh);
    sum = v[0];
    squares = v[1];
}

// prepare_template (:422-439)
__device__ void prepare_template(float* candidate, const float* reference, uint32_t len, float period, bool use_reference) {
    const uint32_t midpoint = len / 2;
    const float max_width = fmaxf((float)max(midpoint, 1u) / 3.0f, 1.0f);
    const float width = rclamp(SLOPE_WIDTH_PERIODS * period, 1.0f, max_width);
    // the reference writes -g(i) at i and +g(i) at the mirror index for i < (len + 1) / 2 (the middle element of an odd length
    // ends up +g), then adds the learnt reference: element-wise here, one pass
    for (uint32_t e = threadIdx.x; e < len; e += 256) {
        const uint32_t mirror = len - 1 - e;
        const bool lower = e < (len + 1) / 2 && mirror != e;
        const float weight = gaussian(len, lower ? e : mirror, width);
        float v = lower ? -0.5f * EDGE_STRENGTH * 2.0f * weight : 0.5f * EDGE_STRENGTH * 2.0f * weight;
        if (use_reference) v += reference[e];
        candidate[e] = v;
    }
    __syncthreads();
}

// Evaluate scores[offset] for offsets lo + k*step (k = 0..count-1); one wave per offset.
// Up to M offsets per wavefront in ONE sweep over the template: y[i] is read once for all of them, the M accumulation chains
// are independent (a single correlation per sweep was a dependent ds_read -> fma chain, 30 exposed LDS round trips each), and
// every offset keeps the element order of a one-offset-per-wavefront sweep (lane l sums elements l, l + 64, ...; xor-shuffle
// tree) — normalized_correlation (:210-236) with the template statistics precomputed; same sums, bit for bit.
template <int M>
__device__ void wave_correlations(float* scores, const float* work, const float* tmpl, uint32_t len, float sum_y, float sum_yy,
                                  const uint32_t (&off)[M], int valid) {
    const unsigned lane = threadIdx.x & 63;
    float sx[M], sxx[M], sxy[M];
#pragma unroll
    for (int m = 0; m < M; ++m) sx[m] = sxx[m] = sxy[m] = 0.0f;
    for (uint32_t i = lane; i < len; i += 64) {
        const float yv = tmpl[i];
        float xv[M];
#pragma unroll
        for (int m = 0; m < M; ++m) xv[m] = work[off[m] + i];  // invalid slots repeat a valid offset: always in bounds
#pragma unroll
        for (int m = 0; m < M; ++m) {
            sx[m] += xv[m];
            sxx[m] += xv[m] * xv[m];
            sxy[m] += xv[m] * yv;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            sx[m] += __shfl_xor(sx[m], o);
            sxx[m] += __shfl_xor(sxx[m], o);
            sxy[m] += __shfl_xor(sxy[m], o);
        }
    }
    if (lane != 0) return;
    const float nf = (float)len;
    const float ey = fmaxf(sum_yy - sum_y * sum_y / nf, 0.0f);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        if (m >= valid) break;
        float v = 0.0f;
        if (len != 0) {
            const float dot = sxy[m] - sx[m] * sum_y / nf;
            const float ex = fmaxf(sxx[m] - sx[m] * sx[m] / nf, 0.0f);
            const float denom = sqrtf(ex * ey);
            v = denom > F32_EPS ? rclamp(dot / denom, -1.0f, 1.0f) : 0.0f;
        }
        scores[off[m]] = v;
    }
}

// Evaluate scores[offset] for offsets lo + k*step (k = 0..count-1); wavefront w takes k = w, w + 4, ...
__device__ void eval_scores(float* scores, const float* work, const float* tmpl, uint32_t len, float sum_y, float sum_yy,
                            uint32_t lo, uint32_t step, uint32_t count_in, bool extra_zero) {
    const unsigned wave = threadIdx.x >> 6;
    constexpr int M = 8;
    const uint32_t count = count_in + (extra_zero ? 1u : 0u);  // entry `count_in` is offset 0
    for (uint32_t k0 = wave; k0 < count; k0 += 4 * M) {
        uint32_t off[M];
        int valid = 0;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const uint32_t k = k0 + 4u * (uint32_t)m;
            const uint32_t kc = k < count ? k : k0;
            off[m] = kc < count_in ? lo + kc * step : 0u;
            valid += k < count ? 1 : 0;
        }
        if (valid <= 2) {
            const uint32_t o2[2] = {off[0], off[1]};
            wave_correlations<2>(scores, work, tmpl, len, sum_y, sum_yy, o2, valid);
        } else if (valid <= 4) {
            const uint32_t o4[4] = {off[0], off[1], off[2], off[3]};
            wave_correlations<4>(scores, work, tmpl, len, sum_y, sum_yy, o4, valid);
        } else {
            wave_correlations<M>(scores, work, tmpl, len, sum_y, sum_yy, off, valid);
        }
    }
    __syncthreads();
}

// The strict-> scan of one search round (:455-470): offsets top, top - step, ... (count of them), then 0 when `extra_zero`;
// the incumbent (bo, bs) only loses to a strictly larger score, the earliest offset in scan order wins among equals.  One
// wavefront reads the scores in parallel and reduces (score, scan index) — the serial scan on thread 0 was 10-27 dependent LDS
// round trips per round.
__device__ void select_best(const float* scores, uint32_t top, uint32_t step, uint32_t count, bool extra_zero, uint32_t& bo, float& bs,
                            Shared& sh) {
    const uint32_t total = count + (extra_zero ? 1u : 0u);
    if (threadIdx.x < 64) {
        if (total <= 64) {
            const uint32_t k = threadIdx.x;
            const uint32_t off = k < count ? top - k * step : 0u;
            float v = k < total ? scores[off] : NEG_INF;
            uint32_t kk = k < total ? k : 0xFFFFFFFFu;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const float ov = __shfl_xor(v, o);
                const uint32_t ok = __shfl_xor(kk, o);
                if (ov > v || (ov == v && ok < kk)) {
                    v = ov;
                    kk = ok;
                }
            }
            if (threadIdx.x == 0) {
                if (kk != 0xFFFFFFFFu && v > bs) {
                    bo = kk < count ? top - kk * step : 0u;
                    bs = v;
                }
                sh.u[0] = bo;
                sh.f[0] = bs;
            }
        } else if (threadIdx.x == 0) {
            for (uint32_t k = 0; k < total; ++k) {
                const uint32_t off = k < count ? top - k * step : 0u;
                if (scores[off] > bs) { bo = off; bs = scores[off]; }
            }
            sh.u[0] = bo;
            sh.f[0] = bs;
        }
    }
    __syncthreads();
    bo = sh.u[0];
    bs = sh.f[0];
    __syncthreads();
}

// find_best (:441-484): coarse-to-fine search; the lazily filled score cache of the reference becomes
// "evaluate the offsets of this round in parallel, then replay the strict-> scan".
__device__ void find_best(float* scores, const float* work, const float* tmpl, uint32_t len, uint32_t search, float period,
                          uint32_t& best_off_out, float& frac_out, Shared& sh) {
    float sum_y, sum_yy;
    correlation_stats(tmpl, len, sum_y, sum_yy, sh);
    uint32_t stride = f2u(roundf(period / 16.0f));
    stride = min(max(stride, 1u), 128u);
    stride = min(stride, max(search, 1u));
    // coarse: (0..=search).rev().step_by(stride).chain([0]) — offset 0 rides in the same sweep
    const uint32_t n_coarse = search / stride + 1;
    const uint32_t lowest = search - (n_coarse - 1) * stride;
    eval_scores(scores, work, tmpl, len, sum_y, sum_yy, lowest, stride, n_coarse, lowest != 0);
    uint32_t best_off = search / 2;
    float best_score = NEG_INF;
    select_best(scores, search, stride, n_coarse, true, best_off, best_score, sh);
    // offsets [dense_lo, dense_hi] all carry a score of THIS call (needed for the parabolic refinement below)
    uint32_t dense_lo = stride == 1 ? 0u : 1u, dense_hi = stride == 1 ? search : 0u;
    uint32_t step = stride;
    while (step > 1) {
        const uint32_t next = max(step / 4, 1u);
        const uint32_t lo = best_off > step ? best_off - step : 0;
        const uint32_t hi = min(best_off + step, search);
        const uint32_t cnt = (hi - lo) / next + 1;
        const uint32_t first = hi - (cnt - 1) * next;
        eval_scores(scores, work, tmpl, len, sum_y, sum_yy, first, next, cnt, false);
        select_best(scores, hi, next, cnt, false, best_off, best_score, sh);  // (lo..=hi).rev().step_by(next)
        if (next == 1) {
            dense_lo = first;
            dense_hi = hi;
        }
        step = next;
    }
    float frac = 0.0f;
    if (best_off > 0 && best_off < search) {
        if (!(best_off - 1 >= dense_lo && best_off + 1 <= dense_hi))  // the reference's cache would compute them now
            eval_scores(scores, work, tmpl, len, sum_y, sum_yy, best_off - 1, 2, 2, false);
        const float prev = scores[best_off - 1], nxt = scores[best_off + 1];
        frac = rclamp(parabolic_refine(prev, best_score, nxt, best_off) - (float)best_off, -0.5f, 0.5f);
    }
    best_off_out = best_off;
    frac_out = frac;
}

// write_candidate (:509-527): candidate = windowed, peak-normalised, mean-removed segment; returns the
// correlation with the reference.
__device__ float write_candidate(float* candidate, const float* reference, const View& seg, float period, Shared& sh) {
    // same arithmetic per element as mean-remove / normalize_peak / window / correlation_stats / normalized_correlation run one
    // after the other (seven workgroup reductions and as many passes): fused into three passes and three reductions
    const uint32_t n = seg.n;
    float part = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 256) part += seg.at(i);
    const float mean = block_sum(part, sh) / (float)max(n, 1u);
    float pk = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const float c = seg.at(i) - mean;
        candidate[i] = c;
        pk = fmaxf(pk, fabsf(c));
    }
    pk = block_max(pk, sh);
    const float scale = 1.0f / fmaxf(pk, NORMALIZE_FLOOR);  // normalize_peak (:191-197)
    const float std_ = fmaxf(period * BUFFER_FALLOFF_PERIODS, 1.0f);
    float acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // sum y, sum y^2, sum x, sum x^2, sum x y
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t mirror = n - 1 - i;
        const float weight = gaussian(n, i < (n + 1) / 2 ? i : mirror, std_);
        float yv = candidate[i] * scale;
        yv *= weight;
        candidate[i] = yv;
        const float xv = reference[i];
        acc[0] += yv;
        acc[1] += yv * yv;
        acc[2] += xv;
        acc[3] += xv * xv;
        acc[4] += xv * yv;
    }
    block_sum_n<5>(acc, sh);  // its barriers also publish candidate[]
    if (n == 0) return 0.0f;
    const float sum_y = acc[0], sum_yy = acc[1], sx = acc[2], sxx = acc[3], sxy = acc[4];
    const float nf = (float)n;
    const float dot = sxy - sx * sum_y / nf;
    const float ex = fmaxf(sxx - sx * sx / nf, 0.0f);
    const float ey = fmaxf(sum_yy - sum_y * sum_y / nf, 0.0f);
    const float denom = sqrtf(ex * ey);
    return denom > F32_EPS ? rclamp(dot / denom, -1.0f, 1.0f) : 0.0f;
}

// locate (:358-411)
__device__ Capture locate(ScopeTriggerState& t, float* reference, const View& trace, Estimate est, uint32_t cycles, float rate,
                          Scratch& sc, Shared& sh, PhaseClock& pc) {
    Capture none{0, 0.0f, 0, 0.0f};
    const uint32_t n = trace.n;
    const float period = fmaxf(est.period, 1.0f);
    const float span = period * (float)max(cycles, 1u);
    const uint32_t frames = f2u(ceilf(span)) + 1;
    const uint32_t len = trigger_kernel_len(period, rate);
    const uint32_t before = len / 2, after = len - before;
    const uint32_t tail = max(frames, after);
    if (n < tail) return none;
    const uint32_t right = n - tail;
    if (right < before) return none;
    uint32_t search = max(f2u(roundf(period * SEARCH_PERIODS)), 1u);
    search = min(min(search, len / 2), right - before);
    const uint32_t left = right - search;
    const View data = trace.sub(left - before, (right + after) - (left - before));

    // prepare (:413-420): retune_reference (:486-498) + EMA-tracked mean + work = data - mean
    if (t.ref_len == 0) {
        for (uint32_t i = threadIdx.x; i < len; i += 256) reference[i] = 0.0f;
        t.ref_len = len;
        t.reference_period = period;
    } else {
        const float semitones = log2f(period / t.reference_period) * 12.0f;
        if (t.ref_len != len || fabsf(semitones) >= BUFFER_RETUNE_SEMITONES) {  // retune_reference fn (:249-263)
            const float ratio = period / t.reference_period;
            const bool bad = !isfinite(ratio) || ratio <= F32_EPS;
            const float old_center = (float)(t.ref_len ? t.ref_len - 1 : 0) * 0.5f;
            const float new_center = (float)(len ? len - 1 : 0) * 0.5f;
            for (uint32_t i = threadIdx.x; i < len; i += 256)
                sc.retuned[i] = bad ? 0.0f : sample_linear_zero(reference, t.ref_len, old_center + ((float)i - new_center) / ratio);
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < len; i += 256) reference[i] = sc.retuned[i];
            t.ref_len = len;
            t.reference_period = period;
        }
    }
    __syncthreads();
    float red[2] = {0.0f, 0.0f};  // sum of the search span, peak of the reference: one reduction
    for (uint32_t i = threadIdx.x; i < data.n; i += 256) red[0] += data.at(i);
    for (uint32_t i = threadIdx.x; i < len; i += 256) red[1] = fmaxf(red[1], fabsf(reference[i]));
    block_sum_n<2, true>(red, sh);
    const float mean = red[0] / (float)max(data.n, 1u);
    t.mean += MEAN_RESPONSIVENESS * (mean - t.mean);
    for (uint32_t i = threadIdx.x; i < data.n; i += 256) sc.work[i] = data.at(i) - t.mean;
    const bool use_reference = red[1] > 1.0e-3f;  // any(|sample| > 1e-3) (:381)
    prepare_template(sc.candidate, reference, len, period, use_reference);  // ends with the barrier that also publishes work[]
    pc.mark(6);  // (sub-phase of locate) retune, span mean, work[], template
    uint32_t offset;
    float frac_offset;
    find_best(sc.scores, sc.work, sc.candidate, len, search, period, offset, frac_offset, sh);
    pc.mark(7);  // (sub-phase) coarse-to-fine search
    const bool confident = est.confidence >= MIN_PERIODICITY;
    bool reset = false;
    bool candidate_written = false;
    if (confident && use_reference) {
        reset = write_candidate(sc.candidate, reference, trace.sub(left + offset - before, len), period, sh) < RESET_BELOW_MATCH;
        candidate_written = true;
    }
    if (reset) {
        for (uint32_t i = threadIdx.x; i < len; i += 256) reference[i] = 0.0f;
        __syncthreads();
        prepare_template(sc.candidate, reference, len, period, false);
        find_best(sc.scores, sc.work, sc.candidate, len, search, period, offset, frac_offset, sh);
        candidate_written = false;
    }
    pc.mark(8);  // (sub-phase) candidate vs reference (+ reset search)
    if (confident) {
        if (!use_reference || reset || !candidate_written)
            write_candidate(sc.candidate, reference, trace.sub(left + offset - before, len), period, sh);
        // update_reference (:500-507)
        float rpk = 0.0f;
        for (uint32_t i = threadIdx.x; i < len; i += 256) rpk = fmaxf(rpk, fabsf(reference[i]));
        rpk = block_max(rpk, sh);
        const float rscale = 1.0f / fmaxf(rpk, NORMALIZE_FLOOR);  // normalize_peak (:191-197), then the EMA, in one pass
        for (uint32_t i = threadIdx.x; i < len; i += 256) {
            float rv = reference[i] * rscale;
            rv += BUFFER_RESPONSIVENESS * (sc.candidate[i] - rv);
            reference[i] = rv;
        }
        t.reference_period += BUFFER_RESPONSIVENESS * (period - t.reference_period);
        __syncthreads();
    }
    pc.mark(9);  // (sub-phase) reference update
    uint32_t start = left + offset;
    if (frac_offset < 0.0f && start > 0) {
        start -= 1;
        frac_offset += 1.0f;
    }
    return Capture{1, span, start, frac_offset};
}

// StableTrigger::capture (:306-334)
__device__ Capture stable_capture(ScopeTriggerState& t, float* reference, const View& trace, const ScopeArgs& a, v2f* fft,
                                  Scratch& sc, Shared& sh, PhaseClock& pc, const ScopeTwiddles& tw, const ScopeEstimate* pre) {
    const uint32_t n = trace.n;
    const uint32_t probe_len = min(a.probe_frames, n);
    float last_peak = 0.0f;
    Estimate detected{0, 0.0f, 0.0f};
    if (pre) {  // two-pass form: computed by scope_estimate_kernel from the same view
        detected = Estimate{pre->some, pre->period, pre->confidence};
        last_peak = pre->last_peak;
    } else if (probe_len >= 3) {
        detected = estimate_period(trace.sub(n - probe_len, probe_len), a.sample_rate, last_peak, a, fft, sc, sh, pc, tw);
    }
    pc.mark(3);  // NSDF + peak picking
    if (probe_len > 0 && last_peak < MIN_SIGNAL_PEAK) trigger_unlock(t);
    const Estimate est = stabilize(t, detected);
    if (est.some) {
        const Capture c = locate(t, reference, trace, est, a.num_cycles, a.sample_rate, sc, sh, pc);
        pc.mark(4);  // locate (template correlation search)
        if (c.some) return c;
    }
    Capture c;
    c.some = 1;
    c.span = (float)max(a.base_frames > 0 ? a.base_frames - 1 : 0u, 1u);
    c.start = n > a.base_frames ? n - a.base_frames : 0;
    c.frac_offset = 0.0f;
    return c;
}

// find_rising_zero_crossing over indices lo..=hi (:530-551); reversed = iterate from hi down
__device__ uint32_t find_rising_zero_crossing(const View& v, uint32_t lo, uint32_t hi, bool reversed, Shared& sh) {
    if (lo > hi) return 0xFFFFFFFFu;  // callers only pass in-range spans (hi < v.n)
    // a crossing between adjacent indices (i-1, i): v[i] > 0 && v[i-1] <= 0, reported as i
    if (!reversed) {
        uint32_t best = 0xFFFFFFFFu;
        for (uint32_t i = lo + 1 + threadIdx.x; i <= hi; i += 256)
            if (v.at(i) > 0.0f && v.at(i - 1) <= 0.0f) { best = i; break; }
        return block_min_u32(best, sh);
    }
    unsigned long long best = 0ull;
    for (uint32_t i = lo + 1 + threadIdx.x; i <= hi; i += 256)
        if (v.at(i) > 0.0f && v.at(i - 1) <= 0.0f) best = (unsigned long long)i + 1ull;  // keep the largest
    best = block_max_u64(best, sh);
    return best == 0ull ? 0xFFFFFFFFu : (uint32_t)(best - 1ull);
}

// zero_crossing_capture (:769-786)
__device__ Capture zero_crossing_capture(const View& v, uint32_t frames_in, uint32_t search_range, Shared& sh) {
    const uint32_t frames = min(frames_in, v.n);
    if (frames == 0) return Capture{0, 0.0f, 0, 0.0f};
    const uint32_t end = v.n > 0 ? v.n - 1 : 0;
    const uint32_t right_lo = end > search_range ? end - search_range : 0;
    uint32_t right = find_rising_zero_crossing(v, right_lo, end, true, sh);
    if (right == 0xFFFFFFFFu) right = end;
    const uint32_t left_lo = right > frames ? right - frames : 0;
    const uint32_t left_hi = min(left_lo + search_range, right > 2 ? right - 2 : 0u);
    uint32_t left = find_rising_zero_crossing(v, left_lo, left_hi, false, sh);
    if (left == 0xFFFFFFFFu) left = left_lo;
    return Capture{1, (float)max(right > left ? right - left : 0u, 1u), left, 0.0f};
}

}  // namespace

__global__ __launch_bounds__(256) void oscilloscope_kernel(ScopeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ Shared sh;
    __shared__ ScopeTriggerState trig[kScopeTraces];
    const unsigned tid = threadIdx.x;
    const uint32_t s = blockIdx.x;
    v2f* fft = a.fft_global ? a.fft_global + (uint64_t)s * a.fft_size : reinterpret_cast<v2f*>(smem_raw);
    float* scratch = a.scratch + (uint64_t)s * a.scratch_stride;
    __shared__ float partials[256];
    Scratch sc;
    {
        // layout mirrors scope_scratch_floats()
        const uint32_t ms = f2u(ceilf((float)a.max_period * SEARCH_PERIODS)) + 2;
        uint64_t off = 0;
        sc.work = scratch + off;       off += (uint64_t)a.max_kernel + ms + 8;
        sc.candidate = scratch + off;  off += (uint64_t)a.max_kernel + 8;
        sc.retuned = scratch + off;    off += (uint64_t)a.max_kernel + 8;
        sc.nsdf = scratch + off;       off += (uint64_t)a.max_period + 8;
        sc.scores = scratch + off;     off += (uint64_t)max(ms, 256u) + 8;
        sc.energy = scratch + off;
        sc.partials = partials;
        if (a.lds_scratch == 2) {
            // pre-pushed form: the estimate arrays are not used here, the whole dynamic region is the locate layout
            float* lds_f = reinterpret_cast<float*>(smem_raw);
            sc.work = lds_f;
            sc.candidate = sc.work + a.max_kernel + ms + 8;
            sc.scores = sc.candidate + a.max_kernel + 8;
        } else if (a.lds_scratch) {
            // The hot arrays live in LDS (the global copies above stay allocated as the fallback layout).  Two phases alias the
            // same dynamic region:  estimate_period  [ 4096-point FFT buffer | nsdf | energy ]
            //                       locate           [ work | candidate | scores ]
            float* lds_f = reinterpret_cast<float*>(smem_raw);
            sc.nsdf = lds_f + 2 * FFT4096_LDS;
            sc.energy = sc.nsdf + a.max_period + 8;
            sc.work = lds_f;
            sc.candidate = sc.work + a.max_kernel + ms + 8;
            sc.scores = sc.candidate + a.max_kernel + 8;
        }
    }
    __shared__ v2f tw2_lds[256];
    ScopeTwiddles tw;
    tw.j = tid;
    tw.tw3_global = a.tw4096;
    tw.tw2 = tw2_lds;
    if (a.tw4096) {  // fast autocorrelation configuration: twiddles resident for every block of the call
#pragma unroll
        for (int t = 1; t < 16; ++t) tw.tw3[t - 1] = a.tw4096[tid * (unsigned)t];
        tw2_lds[tid] = a.tw256[tid];
    }
    // ragged banks: the stream's own block count, ring positions and reset flag (workgroup-uniform: one workgroup per stream)
    const bool ragged = a.blocks_v != nullptr;
    // resume_mode: the capped wide trigger pass (scope_fast_kernels.hip) ran the stream's blocks up to resume_blk[s]; this kernel takes
    // the rest from the positions it left (its reset, if any, is already in the state)
    const bool resume = a.resume_mode != 0;
    const bool reset_stream = !resume && ragged && a.reset_v != nullptr && a.reset_v[s] != 0;  // clear_history (:714-723) of this stream
    const uint32_t n_blocks_s = ragged ? a.blocks_v[s] : a.n_blocks;
    const uint32_t block_frames_s = a.frames_v != nullptr ? a.frames_v[s] : a.block_frames;  // chunk calls: the stream's own block length
    const uint32_t first_blk = resume ? a.resume_blk[s] : 0u;
    if (first_blk >= n_blocks_s && resume) return;  // (workgroup-uniform; nothing left for this stream)
    if (tid < kScopeTraces) {
        ScopeTriggerState t0;
        memset(&t0, 0, sizeof(t0));
        trig[tid] = reset_stream ? t0 : a.trig[(uint64_t)s * kScopeTraces + tid];
    }
    __syncthreads();
    const uint64_t mask = a.cap - 1;
    float* rings = a.rings + (uint64_t)s * kScopeTraces * a.cap;
    const float* pcm = a.pcm + (uint64_t)s * a.frames_total * a.fmt.channels;
    uint64_t head[kScopeTraces], len[kScopeTraces];
    for (int t = 0; t < kScopeTraces; ++t) {
        head[t] = ragged ? a.pos_v[((uint64_t)s * kScopeTraces + t) * 2] : a.head[t];
        len[t] = ragged ? (reset_stream ? 0ull : a.pos_v[((uint64_t)s * kScopeTraces + t) * 2 + 1]) : a.len[t];
        if (resume) {
            head[t] = a.resume_pos[((uint64_t)s * kScopeTraces + t) * 2];
            len[t] = a.resume_pos[((uint64_t)s * kScopeTraces + t) * 2 + 1];
        }
    }
    const bool active[2] = {a.trace_channel[0] != OMX_CHANNEL_NONE, a.trace_channel[1] != OMX_CHANNEL_NONE};

    PhaseClock pc;
    pc.start(a.phase_timing != 0);
    for (uint32_t blk = first_blk; blk < n_blocks_s; ++blk) {
        // ---- push projected frames (:657-681) — unless scope_push2_kernel has put the whole call into the rings already
        for (uint32_t f = a.pre_pushed ? block_frames_s : tid; f < block_frames_s; f += 256) {
            const float* frame = pcm + ((uint64_t)blk * block_frames_s + f) * a.fmt.channels;
            float left = 0.0f, right = 0.0f;
            for (uint32_t c = 0; c < a.fmt.channels; ++c) {
                const float v = frame[c];
                left = left + v * a.fmt.m[c][0];
                right = right + v * a.fmt.m[c][1];
            }
            for (int t = 0; t < kScopeTraces; ++t) {
                const uint32_t ch = t < 2 ? a.trace_channel[t] : a.trigger_source;
                const bool on = t < 2 ? active[t] : a.separate_source != 0;
                if (!on) continue;
                float v;
                switch (ch) {
                    case OMX_CHANNEL_LEFT: v = left; break;
                    case OMX_CHANNEL_RIGHT: v = right; break;
                    case OMX_CHANNEL_MID: v = (left + right) * 0.5f; break;
                    case OMX_CHANNEL_SIDE: v = (left - right) * 0.5f; break;
                    default: v = 0.0f; break;
                }
                rings[(uint64_t)t * a.cap + ((head[t] + f) & mask)] = v;
            }
        }
        for (int t = 0; t < kScopeTraces; ++t) {
            const bool on = t < 2 ? active[t] : a.separate_source != 0;
            if (on) {
                head[t] += block_frames_s;
                len[t] = min(len[t] + (uint64_t)block_frames_s, (uint64_t)a.history_frames);
            } else {
                len[t] = 0;
            }
        }
        __syncthreads();
        pc.mark(0);  // ring push
        View views[kScopeTraces];
        for (int t = 0; t < kScopeTraces; ++t)
            views[t] = View{rings + (uint64_t)t * a.cap, (uint32_t)((head[t] - len[t]) & mask), (uint32_t)mask, (uint32_t)len[t]};

        // ---- captures (:683-700)
        auto capture = [&](int view_index, int trig_index) -> Capture {
            const View& trace = views[view_index];
            // estimates ahead of the trigger pass (scope_estimate_big_kernel: fft_size 16384 / 32768), else computed here
            const ScopeEstimate* pre = a.pre_pushed && a.estimates ? a.estimates + ((uint64_t)s * a.n_blocks + blk) * kScopeTraces + view_index : nullptr;
            if (a.trigger_mode == OMX_TRIGGER_ZERO_CROSSING) return zero_crossing_capture(trace, a.base_frames, a.max_period, sh);
            if (trace.n >= a.base_frames) {
                float* reference = a.reference + ((uint64_t)s * kScopeTraces + trig_index) * a.max_kernel;
                // the trigger state lives in LDS; thread-uniform updates are done redundantly by every thread
                ScopeTriggerState local = trig[trig_index];
                const Capture c = stable_capture(local, reference, trace, a, fft, sc, sh, pc, tw, pre);
                __syncthreads();
                if (tid == 0) trig[trig_index] = local;
                __syncthreads();
                return c;
            }
            return Capture{0, 0.0f, 0, 0.0f};
        };
        Capture linked{0, 0.0f, 0, 0.0f};
        if (a.matching_trace >= 0) linked = capture(a.matching_trace, 2);
        else if (a.separate_source) linked = capture(2, 2);
        Capture caps[2];
        for (int slot = 0; slot < 2; ++slot) {
            caps[slot] = Capture{0, 0.0f, 0, 0.0f};
            if (!active[slot]) continue;
            caps[slot] = linked.some ? linked : capture(slot, slot);
        }

        // ---- write_snapshot (:725-750) + downsample_trace (:788-803)
        ScopeBlockHeader hdr;
        memset(&hdr, 0, sizeof(hdr));
        if (caps[0].some || caps[1].some) {
            uint32_t target = 0;
            bool any = false;
            for (int slot = 0; slot < 2; ++slot)
                if (caps[slot].some) {
                    const uint32_t tt = f2u(fmaxf(roundf(caps[slot].span), 1.0f)) + 1;
                    target = any ? max(target, tt) : tt;
                    any = true;
                }
            target = min(max(target, 2u), (uint32_t)kScopeTarget);
            hdr.produced = 1;
            hdr.capture_start = caps[0].some ? caps[0].start : caps[1].start;
            hdr.capture_frac = caps[0].some ? caps[0].frac_offset : caps[1].frac_offset;
            const bool newest = blk + 1 == n_blocks_s;
            for (int slot = 0; slot < 2; ++slot) {
                if (!caps[slot].some) continue;
                const View& tr = views[slot];
                const uint32_t start = min(caps[slot].start, tr.n);
                const View data = tr.sub(start, tr.n - start);
                if (data.n < 2) continue;
                const float last = (float)(data.n - 1);
                const float start_offset = rclamp(caps[slot].frac_offset, 0.0f, last);
                const float span = fminf(caps[slot].span, last - start_offset);
                if (!(isfinite(span) && span > 0.0f)) continue;
                const float step = span / (float)(target - 1);
                if (newest) {
                    float* out = a.samples + ((uint64_t)s * 2 + hdr.channels) * kScopeTarget;
                    for (uint32_t i = tid; i < target; i += 256) out[i] = sample_linear_zero_view(data, start_offset + (float)i * step);
                }
                hdr.slots[hdr.channels] = (uint32_t)slot;
                hdr.channels += 1;
            }
            hdr.samples_per_channel = hdr.channels == 0 ? 0 : target;
        }
        // last_cycle_rate (:602-609): source trigger first, then the traces
        {
            int which = -1;
            if (trig[2].has_period) which = 2;
            else if (trig[0].has_period) which = 0;
            else if (trig[1].has_period) which = 1;
            hdr.locked = which >= 0 ? 1 : 0;
            hdr.period = which >= 0 ? trig[which].period : 0.0f;
        }
        if (tid == 0) a.headers[(uint64_t)s * a.n_blocks + blk] = hdr;
        __syncthreads();
        pc.mark(5);  // snapshot
    }
    if (tid < kScopeTraces) a.trig[(uint64_t)s * kScopeTraces + tid] = trig[tid];
    if (ragged && tid == 0) {
        for (int t = 0; t < kScopeTraces; ++t) {
            a.pos_v[((uint64_t)s * kScopeTraces + t) * 2] = head[t];
            a.pos_v[((uint64_t)s * kScopeTraces + t) * 2 + 1] = len[t];
        }
        if (reset_stream) a.epoch_v[s] += 1;
    }
}

uint64_t scope_scratch_floats(uint32_t max_kernel, uint32_t max_search_unused, uint32_t probe_frames, uint32_t max_period) {
    (void)max_search_unused;
    const uint32_t ms = (uint32_t)std::ceil((float)max_period * 1.5f) + 2;
    uint64_t off = 0;
    off += (uint64_t)max_kernel + ms + 8;
    off += (uint64_t)max_kernel + 8;
    off += (uint64_t)max_kernel + 8;
    off += (uint64_t)max_period + 8;
    off += (uint64_t)std::max(ms, 256u) + 8;
    off += (uint64_t)probe_frames + 8;
    return off;
}

// bytes of the two aliased LDS layouts of the fast configuration (see oscilloscope_kernel)
uint64_t scope_lds_scratch_bytes(uint32_t max_kernel, uint32_t max_period, uint32_t probe_frames) {
    const uint32_t ms = (uint32_t)std::ceil((float)max_period * 1.5f) + 2;
    const uint64_t estimate = 2ull * FFT4096_LDS + (max_period + 8ull) + (probe_frames + 8ull);
    const uint64_t locate = (max_kernel + ms + 8ull) + (max_kernel + 8ull) + (std::max(ms, 256u) + 8ull);
    return std::max(estimate, locate) * sizeof(float);
}

// bytes of the locate layout alone (work | candidate | scores): the pre-pushed form, whose estimates come from another kernel
uint64_t scope_locate_lds_bytes(uint32_t max_kernel, uint32_t max_period) {
    const uint32_t ms = (uint32_t)std::ceil((float)max_period * 1.5f) + 2;
    return ((max_kernel + ms + 8ull) + (max_kernel + 8ull) + (std::max(ms, 256u) + 8ull)) * sizeof(float);
}

void scope_phase_cycles(unsigned long long out[SCOPE_PHASES], bool reset) {
    OMX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scope_phase_cycles), SCOPE_PHASES * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long zero[SCOPE_PHASES] = {};
        OMX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_scope_phase_cycles), zero, sizeof(zero)));
    }
}

void launch_oscilloscope(const ScopeArgs& a, hipStream_t stream) {
    if (a.n_streams == 0 || a.n_blocks == 0) return;
    size_t lds = (a.fft_global || a.pre_pushed) ? 0 : (size_t)a.fft_size * sizeof(v2f);
    if (a.lds_scratch == 2) lds = (size_t)scope_locate_lds_bytes(a.max_kernel, a.max_period);
    else if (a.lds_scratch) lds = std::max(lds, (size_t)scope_lds_scratch_bytes(a.max_kernel, a.max_period, a.probe_frames));
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(oscilloscope_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL(oscilloscope_kernel, dim3(a.n_streams), dim3(256), lds, stream, a);
}

}  // namespace omx
