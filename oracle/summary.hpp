// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// State-side summary reductions, restated from
//   src/visuals/spectrum/state.rs:320-356   peak_bin, interpolated_peak
//   src/visuals/loudness/state.rs:36-60     PeakHold
//   src/visuals/loudness/state.rs:121-184   get_value, aggregate_channels, visible_values
//   src/visuals/loudness/state.rs:211-246   update_peak_holds, channel_side
// Pinned by the reference's own state tests (loudness/state.rs:370-427), ported in tests/test_kat_summary.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../include/omx.h"
#include "primitives.hpp"

namespace omxo {

constexpr float kPeakEpsilon = 1e-6f;          // spectrum/state.rs:20
constexpr float kMeterDbLo = -60.0f, kMeterDbHi = 4.0f;  // loudness/render.rs:11 DB_RANGE
constexpr double kPeakHoldSeconds = 2.0;       // loudness/state.rs:20
constexpr float kPeakDecayDbPerSec = 60.0f;    // :21

// f32::total_cmp key: monotone map of the bit pattern to a signed integer
inline int32_t total_order_key(float v) {
    int32_t b;
    std::memcpy(&b, &v, 4);
    return b ^ (int32_t)((uint32_t)(b >> 31) >> 1);
}

// spectrum/state.rs:320-324: last maximum (Iterator::max_by keeps the later of equal elements)
inline bool peak_bin(const float* bins, const float* db, size_t n, float min_f, float max_f, size_t* out) {
    bool found = false;
    size_t best = 0;
    for (size_t i = 1; i + 1 < n; ++i) {
        if (!(bins[i] >= min_f && bins[i] <= max_f) || !std::isfinite(db[i])) continue;
        if (!found || total_order_key(db[i]) >= total_order_key(db[best])) {
            best = i;
            found = true;
        }
    }
    *out = best;
    return found;
}

// spectrum/state.rs:326-356
inline bool interpolated_peak(const float* bins, const float* db, size_t n, size_t bin, float* freq, float* level) {
    const size_t next = bin + 1;
    if (bin == 0 || next >= n) return false;
    const float bin_hz = bins[1] - bins[0];
    const float center_freq = bins[bin], center = db[bin];
    if (!(std::isfinite(bin_hz) && bin_hz > 0.0f) || !std::isfinite(center_freq) || !std::isfinite(center)) return false;
    const float left = db[bin - 1], right = db[next];
    float offset = 0.0f;
    if (std::isfinite(left) && std::isfinite(right)) {
        const float denom = left - 2.0f * center + right;
        if (denom < -kPeakEpsilon) {
            offset = 0.5f * (left - right) / denom;
            offset = offset < -0.5f ? -0.5f : (offset > 0.5f ? 0.5f : offset);
        }
    }
    float lv = center;
    if (offset != 0.0f) {
        lv = center - 0.25f * (left - right) * offset;
        lv = lv > center ? lv : center;  // f32::max
    }
    const float f = center_freq + offset * bin_hz;
    *freq = f > 0.0f ? f : 0.0f;
    *level = lv;
    return true;
}

inline omx_spectrum_peak spectrum_peak(const float* bins, const float* db, size_t n, float min_f, float max_f) {
    omx_spectrum_peak p{0, 0, 0.0f, 0.0f};
    size_t bin;
    if (n < 3 || !peak_bin(bins, db, n, min_f, max_f, &bin)) return p;
    float f, l;
    if (!interpolated_peak(bins, db, n, bin, &f, &l)) return p;
    p.found = 1;
    p.bin = (uint32_t)bin;
    p.freq_hz = f;
    p.level_db = l;
    return p;
}

// loudness/state.rs:36-60
inline void peak_hold_update(omx_peak_hold& h, float value, double now) {
    if (value > h.db) {
        h.db = value;
        h.decay_from = now + kPeakHoldSeconds;
    } else if (now > h.decay_from) {
        const float decay_dt = (float)(now - h.decay_from);
        const float d = h.db - kPeakDecayDbPerSec * decay_dt;
        h.db = d > value ? d : value;
        h.decay_from = now;
    }
}

enum MeterSide { SIDE_LEFT, SIDE_RIGHT, SIDE_BOTH, SIDE_NEITHER };
// loudness/state.rs:222-246
inline MeterSide channel_side(uint8_t position, size_t index, size_t total) {
    if (position >= OMX_POS_AUX0 || position == OMX_POS_UNKNOWN) position = positions_fallback((uint32_t)total)[index];
    switch (position) {
        case OMX_POS_FRONT_LEFT: case OMX_POS_REAR_LEFT: case OMX_POS_SIDE_LEFT: return SIDE_LEFT;
        case OMX_POS_FRONT_RIGHT: case OMX_POS_REAR_RIGHT: case OMX_POS_SIDE_RIGHT: return SIDE_RIGHT;
        case OMX_POS_FRONT_CENTER: case OMX_POS_MONO: return SIDE_BOTH;
        default: return SIDE_NEITHER;
    }
}
// :121-131
inline float meter_value(const omx_loudness_snapshot& s, uint32_t mode, size_t channel) {
    switch (mode) {
        case OMX_METER_LUFS_SHORT_TERM: return s.short_term_loudness;
        case OMX_METER_LUFS_MOMENTARY: return s.momentary_loudness;
        case OMX_METER_RMS_FAST: return channel < OMX_MAX_CHANNELS ? s.rms_fast_db[channel] : kMeterDbLo;
        case OMX_METER_RMS_SLOW: return channel < OMX_MAX_CHANNELS ? s.rms_slow_db[channel] : kMeterDbLo;
        default: return channel < OMX_MAX_CHANNELS ? s.true_peak_db[channel] : kMeterDbLo;
    }
}
// :153-169
inline float aggregate_channels(const omx_loudness_snapshot& s, uint32_t mode, MeterSide wanted) {
    if (mode == OMX_METER_LUFS_SHORT_TERM || mode == OMX_METER_LUFS_MOMENTARY) return meter_value(s, mode, 0);
    float acc = kMeterDbLo;
    for (size_t ch = 0; ch < s.channel_count; ++ch) {
        const MeterSide side = channel_side(s.positions[ch], ch, s.channel_count);
        if (side != SIDE_BOTH && side != wanted) continue;
        const float v = meter_value(s, mode, ch);
        acc = std::fmax(acc, v);  // f32::max: NaN loses
    }
    return acc;
}
// :178-184 + :211-217
inline omx_meter_row apply_meter_snapshot(const omx_loudness_snapshot& s, uint32_t left_mode, uint32_t right_mode, double now,
                                          omx_peak_hold holds[3]) {
    omx_meter_row r;
    r.values[0] = aggregate_channels(s, left_mode, SIDE_LEFT);
    r.values[1] = aggregate_channels(s, left_mode, SIDE_RIGHT);
    r.values[2] = meter_value(s, right_mode, 0);
    for (int i = 0; i < 3; ++i) {
        float v = r.values[i];
        v = v < kMeterDbLo ? kMeterDbLo : (v > kMeterDbHi ? kMeterDbHi : v);  // f32::clamp (NaN stays NaN)
        peak_hold_update(holds[i], v, now);
        r.peaks[i] = holds[i].db;
    }
    return r;
}

}  // namespace omxo
