// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Reassigned-splat accumulation + dB resolve, restated from
//   src/visuals/render/shaders/spectrogram.wgsl:66-76    freq_to_norm
//   src/visuals/render/shaders/spectrogram.wgsl:126-147  vs_accum_splat (position, cull)
//   src/visuals/render/shaders/spectrogram.wgsl:215-237  fs_accum (tilt), fs_resolve (scale, dB)
//   src/util/audio/frequency.rs:25-31                    FrequencyScale::scale
//   src/visuals/spectrogram/render.rs:214-224            freq_axis = (scale(min), 1 / (scale(max) - scale(min)))
//   src/visuals/spectrogram/state.rs:53-175              SpectrogramHistory: slot ring bookkeeping (KATs: state.rs:803-861)
//   src/visuals/spectrogram/render.rs:106-160, 221, 457-597  visible slots, newest_col, ring resize copy plan, slot uploads
// PARITY UNPINNED for the pixel values: the reference has no test for these shaders and its numbers come out of a GPU raster
// pipeline with an Rg16Float blend target; this restatement keeps the per-point arithmetic and the pixel-centre coverage rule,
// sums in f32.  The ring bookkeeping IS pinned: the reference's own state tests are ported (tests/test_kat_splat.py).
// Deliberately NOT modelled (properties of the raster target, not of the data path):
//   * Rg16Float storage: every blend result is rounded to f16 in the reference, and the red channel saturates at 65504;
//   * the green side channel `power * LOW_POWER_SCALE` (x 2^24) that fs_resolve prefers while it is < F16_MAX — it exists
//     only to keep small powers above f16's denormal range, an f32 accumulator has no such range problem;
//   * blend order of the rasteriser (f32 atomics here, order-free up to f32 rounding);
//   * `reassigned_points_per_slot` only sizes the GPU vertex buffer (state.rs:131-148): tracked for the KAT, the ring here
//     always has fft_size / 2 + 1 point slots per column;
//   * palette mapping / rotation / classic-mode texture sampling (presentation).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/omx.h"

namespace omxo {

inline float freq_scale_value(uint32_t scale, float hz) {
    switch (scale) {
        case OMX_FREQ_SCALE_LOGARITHMIC: return std::asinh(hz / 20.0f);
        case OMX_FREQ_SCALE_ERB: return 21.4f * std::log(1.0f + hz / 228.8f) * 0.4342944819f;  // wgsl: log * LOG10_E
        default: return hz;
    }
}

// FrequencyScale::scale (frequency.rs:25-31) — the host side of the uniforms uses log10, the shader ln * LOG10_E
inline float freq_scale_host(uint32_t scale, float hz) {
    switch (scale) {
        case OMX_FREQ_SCALE_LOGARITHMIC: return std::asinh(hz / 20.0f);
        case OMX_FREQ_SCALE_ERB: return 21.4f * std::log10(1.0f + hz / 228.8f);
        default: return hz;
    }
}

inline void splat_view_size(omx_splat_view* v) {
    v->width = (uint32_t)std::ceil(std::fmax(v->extent_x, 1.0f));
    v->height = (uint32_t)std::ceil(std::fmax(v->extent_y, 1.0f));
}

struct SplatConsts {
    float axis_lo, axis_inv, inv_uv;
};
inline SplatConsts splat_consts(const omx_splat_view& v) {
    const float lo = freq_scale_host(v.freq_scale, v.freq_min), hi = freq_scale_host(v.freq_scale, v.freq_max);
    return SplatConsts{lo, 1.0f / std::fmax(hi - lo, 1e-12f), 1.0f / std::fmax(v.uv_hi - v.uv_lo, 1e-12f)};
}

// one point of the column with age `age`; adds its (tilted) power to every covered pixel of `accum` [width][height]
inline void splat_point(const omx_spectrogram_point& p, uint32_t age, const omx_splat_view& v, const SplatConsts& c, float* accum) {
    const float zoomed = ((freq_scale_value(v.freq_scale, p.freq_hz) - c.axis_lo) * c.axis_inv - v.uv_lo) * c.inv_uv;
    if (!(p.power > 0.0f) || zoomed < -0.01f || zoomed > 1.01f) return;
    float power = p.power;
    if (v.tilt_db != 0.0f && !(power > 1.0023052e-14f)) return;  // fs_accum: floor bins are not lifted
    if (v.tilt_db != 0.0f && p.freq_hz > 0.0f) power *= std::exp2(v.tilt_db * std::log2(p.freq_hz / 1000.0f) * 0.3321928095f);
    const float sf = v.scale_factor;
    const float x = v.extent_x - ((float)age - p.time_offset) * sf, y = (1.0f - zoomed) * v.extent_y;
    const float x0 = x - 0.5f * sf, x1 = x + 0.5f * sf, y0 = y - 0.5f * sf, y1 = y + 0.5f * sf;
    // pixel (i, j) is covered when its centre (i + 0.5, j + 0.5) lies in [x0, x1) x [y0, y1)
    const float fi0 = std::ceil(x0 - 0.5f), fi1 = std::ceil(x1 - 0.5f), fj0 = std::ceil(y0 - 0.5f), fj1 = std::ceil(y1 - 0.5f);
    if (!(fi1 > 0.0f && fj1 > 0.0f && fi0 < (float)v.width && fj0 < (float)v.height)) return;
    const int64_t i0 = (int64_t)std::fmax(fi0, 0.0f), i1 = (int64_t)std::fmin(fi1, (float)v.width);
    const int64_t j0 = (int64_t)std::fmax(fj0, 0.0f), j1 = (int64_t)std::fmin(fj1, (float)v.height);
    for (int64_t j = j0; j < j1; ++j)
        for (int64_t i = i0; i < i1; ++i) accum[(size_t)i * v.height + (size_t)j] += power;  // [width][height]: one time column is contiguous
}

inline float splat_resolve(float accumulated, float reassigned_power_scale) {
    const float power = accumulated * reassigned_power_scale;
    if (power <= 0.0f || power != power) return -std::numeric_limits<float>::infinity();
    return std::fmax(std::log(std::fmax(power, 1e-20f)) * 4.342944819f, -140.0f);
}

inline void spectrogram_splat(const omx_spectrogram_point* points, const uint32_t* counts, uint64_t n_streams, uint64_t n_columns,
                              uint64_t column_stride, float power_scale, const omx_splat_view& view, float* accum, float* db) {
    const SplatConsts c = splat_consts(view);
    const size_t px = (size_t)view.width * view.height;
    for (uint64_t s = 0; s < n_streams; ++s) {
        float* acc = accum + s * px;
        for (size_t i = 0; i < px; ++i) acc[i] = 0.0f;
        for (uint64_t col = 0; col < n_columns; ++col) {
            const omx_spectrogram_point* col_points = points + (s * n_columns + col) * column_stride;
            const uint32_t n = counts[s * n_columns + col];
            for (uint32_t i = 0; i < n && i < column_stride; ++i)
                splat_point(col_points[i], (uint32_t)(n_columns - 1 - col), view, c, acc);
        }
        if (db)
            for (size_t i = 0; i < px; ++i) db[s * px + i] = splat_resolve(acc[i], power_scale);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// SpectrogramHistory (state.rs:53-175) together with the renderer's column ring it drives (render.rs:457-597): what the GPU
// buffer holds after `prepare` ran on every update.  One stream.
struct SpectrogramHistoryRing {
    uint32_t col_kind = OMX_COLUMN_REASSIGNED, reassigned_points_per_slot = 1, ring_capacity = 0, write_slot = 0, col_count = 0;
    uint32_t points_per_column = 0;
    std::vector<uint32_t> slot_counts;
    std::vector<omx_spectrogram_point> points;  // [ring_capacity][points_per_column]
    std::vector<uint16_t> codes;                // [ring_capacity][points_per_column] (classic)

    static uint64_t byte_stride(uint32_t kind, uint32_t pts) {  // processor.rs:144-151
        return kind == OMX_COLUMN_REASSIGNED ? (uint64_t)pts * sizeof(omx_spectrogram_point) : (((uint64_t)pts + 1) / 2) * 4;
    }
    static uint64_t capacity_for(uint32_t kind, uint32_t pts, uint64_t requested) {  // processor.rs:153-158
        const uint64_t clamped = std::min<uint64_t>(std::max<uint64_t>(requested, 1), 8192);
        const uint64_t budget = (uint64_t)(128u * 1024u * 1024u) * (1 + (kind == OMX_COLUMN_REASSIGNED ? 1 : 0)) /
                                std::max<uint64_t>(byte_stride(kind, pts), 1);
        return std::min(clamped, budget);
    }

    // remap_retained (state.rs:150-174) applied to the stored data too (render.rs:457-504 copies slot src -> dst)
    void remap_retained(uint32_t start, uint32_t keep, uint32_t new_capacity) {
        const uint32_t old_cap = std::max(ring_capacity, 1u), ppc = points_per_column;
        std::vector<uint32_t> counts(col_kind == OMX_COLUMN_REASSIGNED ? keep : 0, 0);
        std::vector<omx_spectrogram_point> np((size_t)new_capacity * ppc);
        std::vector<uint16_t> nc((size_t)new_capacity * ppc);
        for (uint32_t src = 0; src < old_cap; ++src) {
            const uint32_t dst = (src + old_cap - start) % old_cap;
            if (dst >= keep || dst >= new_capacity) continue;
            if (col_kind == OMX_COLUMN_REASSIGNED) {
                if (src < slot_counts.size()) counts[dst] = slot_counts[src];
                if ((size_t)(src + 1) * ppc <= points.size())
                    std::copy(points.begin() + (size_t)src * ppc, points.begin() + (size_t)(src + 1) * ppc, np.begin() + (size_t)dst * ppc);
            } else if ((size_t)(src + 1) * ppc <= codes.size()) {
                std::copy(codes.begin() + (size_t)src * ppc, codes.begin() + (size_t)(src + 1) * ppc, nc.begin() + (size_t)dst * ppc);
            }
        }
        if (col_kind == OMX_COLUMN_REASSIGNED) slot_counts = counts;
        points.swap(np);
        codes.swap(nc);
    }

    void apply_update(const omx_spectrogram_update& snap) {  // state.rs:66-123
        const uint64_t ppc64 = snap.fft_size / 2 + 1;
        if (ppc64 == 0) return;
        const uint32_t ppc = (uint32_t)ppc64;
        const uint32_t new_kind = snap.n_columns ? snap.kind : col_kind;
        const uint32_t capacity = (uint32_t)capacity_for(new_kind, ppc, snap.history_length);
        if (capacity == 0) return;
        if (snap.reset || ppc != points_per_column || new_kind != col_kind) {
            // the reference only rebuilds on `reset`; the processor raises it whenever the shape or the mode changes
            // (processor.rs:518-543), so the two extra conditions never fire on its updates — they keep a hand-made update
            // sequence from indexing a ring of the wrong shape
            *this = SpectrogramHistoryRing();
            col_kind = new_kind;
            ring_capacity = capacity;
            points_per_column = ppc;
            if (new_kind == OMX_COLUMN_REASSIGNED) slot_counts.assign(capacity, 0);
            points.assign(new_kind == OMX_COLUMN_REASSIGNED ? (size_t)capacity * ppc : 0, omx_spectrogram_point{0, 0, 0});
            codes.assign(new_kind == OMX_COLUMN_CLASSIC ? (size_t)capacity * ppc : 0, 0);
        } else if (capacity != ring_capacity) {
            if (capacity > ring_capacity && col_count >= ring_capacity) {
                remap_retained(write_slot, col_count, capacity);
                write_slot = col_count % capacity;
            } else if (capacity < ring_capacity && col_count >= capacity) {
                const uint32_t oldest_kept = (write_slot + ring_capacity - capacity) % ring_capacity;
                remap_retained(oldest_kept, capacity, capacity);
                col_count = capacity;
                write_slot = 0;
            } else {
                remap_retained(0, ring_capacity, capacity);  // identity plan: slots keep their index in the resized buffer
            }
            ring_capacity = capacity;
            if (col_kind == OMX_COLUMN_REASSIGNED) slot_counts.resize(capacity, 0);
            points.resize(col_kind == OMX_COLUMN_REASSIGNED ? (size_t)capacity * ppc : 0, omx_spectrogram_point{0, 0, 0});
            codes.resize(col_kind == OMX_COLUMN_CLASSIC ? (size_t)capacity * ppc : 0, 0);
        }
        for (uint64_t c = 0; c < snap.n_columns; ++c) {
            const uint32_t slot = write_slot;
            const uint64_t lo = snap.column_offsets[c], n = snap.column_offsets[c + 1] - lo;
            if (col_kind == OMX_COLUMN_REASSIGNED) {
                slot_counts[slot] = (uint32_t)n;
                const uint64_t written = std::min<uint64_t>(n, ppc);  // render.rs:573
                std::copy(snap.points + lo, snap.points + lo + written, points.begin() + (size_t)slot * ppc);
            } else if (n) {  // render.rs:583-594: copied, zero-filled to the stride; empty columns are not uploaded
                const uint64_t written = std::min<uint64_t>(n, ppc);
                std::copy(snap.codes + lo, snap.codes + lo + written, codes.begin() + (size_t)slot * ppc);
                std::fill(codes.begin() + (size_t)slot * ppc + written, codes.begin() + (size_t)(slot + 1) * ppc, (uint16_t)0);
            }
            write_slot = (write_slot + 1) % ring_capacity;
            if (col_count < ring_capacity) col_count += 1;
        }
        // fit_reassigned_slot_capacity (state.rs:131-148)
        if (col_kind != OMX_COLUMN_REASSIGNED) {
            reassigned_points_per_slot = 1;
        } else {
            uint32_t needed = 1;
            for (uint32_t i = 0; i < ring_capacity && i < slot_counts.size(); ++i) needed = std::max(needed, slot_counts[i]);
            const uint32_t current = reassigned_points_per_slot;
            const uint64_t quad = std::max<uint64_t>((uint64_t)needed * 4, 1);
            if (needed > current || (uint64_t)current > quad) reassigned_points_per_slot = needed;
        }
    }

    uint32_t newest_slot() const { return ring_capacity ? (write_slot + ring_capacity - 1) % ring_capacity : 0; }  // render.rs:221
    uint32_t visible_slots() const { return std::min(col_count, ring_capacity); }                                 // render.rs:106

    // the accumulation pass over the ring (render.rs:139-160) + resolve: slot -> age = (newest + hl - slot) % hl (wgsl:141)
    void splat(float power_scale, const omx_splat_view& view, float* accum, float* db) const {
        const SplatConsts c = splat_consts(view);
        const size_t px = (size_t)view.width * view.height;
        for (size_t i = 0; i < px; ++i) accum[i] = 0.0f;
        if (col_kind == OMX_COLUMN_REASSIGNED && ring_capacity) {
            const uint32_t hl = ring_capacity, newest = newest_slot();
            for (uint32_t slot = 0; slot < visible_slots(); ++slot) {
                const uint32_t n = std::min(slot_counts[slot], points_per_column), age = (newest + hl - slot) % hl;
                for (uint32_t i = 0; i < n; ++i) splat_point(points[(size_t)slot * points_per_column + i], age, view, c, accum);
            }
        }
        if (db)
            for (size_t i = 0; i < px; ++i) db[i] = splat_resolve(accum[i], power_scale);
    }
};

}  // namespace omxo
