// Column history ring between the spectrogram processor and the splat passes:
//   SpectrogramHistory::apply_update   reference src/visuals/spectrogram/state.rs:53-175
//   ring buffer upload / resize copy    reference src/visuals/spectrogram/render.rs:457-597
//   accumulation pass over the ring     reference src/visuals/spectrogram/render.rs:106-160, :221, spectrogram.wgsl:141-142
// Host side: the ring's integers (capacity, write_slot, col_count — common to the streams of a lock-step bank).  Device side:
// the slots themselves ([n_streams][ring_capacity][fft_size / 2 + 1] points or u16 codes) and slot_counts, moved by the
// kernels at the end of splat_kernels.hip; columns never visit the host on the bank path.
#include "splat.hpp"
#include "spectrogram.hpp"

using namespace omx;

namespace {
float freq_scale_host(uint32_t scale, float hz) {  // util/audio/frequency.rs:25-31
    switch (scale) {
        case OMX_FREQ_SCALE_LOGARITHMIC: return std::asinh(hz / 20.0f);
        case OMX_FREQ_SCALE_ERB: return 21.4f * std::log10(1.0f + hz / 228.8f);
        default: return hz;
    }
}
}  // namespace

struct omx_spectrogram_history {
    uint32_t n_streams;
    uint32_t kind = OMX_COLUMN_REASSIGNED, ring_capacity = 0, write_slot = 0, col_count = 0, ppc = 0;
    DeviceBuffer<unsigned char> ring;     // [n_streams][ring_capacity][ppc] x elem bytes
    DeviceBuffer<uint32_t> slot_counts;   // [n_streams][ring_capacity] (reassigned)
    DeviceBuffer<uint32_t> fit_state;     // [1] reassigned_points_per_slot of stream 0
    DeviceBuffer<unsigned char> staging;  // host-update path
    DeviceBuffer<uint32_t> staging_counts;
    hipStream_t last_stream = nullptr;

    explicit omx_spectrogram_history(uint32_t n) : n_streams(n) {}

    uint32_t elem() const { return kind == OMX_COLUMN_REASSIGNED ? (uint32_t)sizeof(omx_spectrogram_point) : 2u; }
    // classic rows are padded to an even number of codes (col_byte_stride, processor.rs:144-151)
    uint32_t stride_elems() const { return kind == OMX_COLUMN_REASSIGNED ? ppc : ((ppc + 1u) / 2u) * 2u; }
    size_t ring_bytes(uint32_t slots) const { return (size_t)n_streams * slots * stride_elems() * elem(); }
    uint32_t newest_slot() const { return ring_capacity ? (write_slot + ring_capacity - 1) % ring_capacity : 0; }
    uint32_t visible_slots() const { return std::min(col_count, ring_capacity); }

    void set_fit(uint32_t v, hipStream_t st) {
        fit_state.reserve(1);
        OMX_HIP(hipMemcpyAsync(fit_state.ptr, &v, sizeof(v), hipMemcpyHostToDevice, st));
        OMX_HIP(hipStreamSynchronize(st));
    }

    void rebuild(uint32_t new_kind, uint32_t capacity, uint32_t new_ppc, hipStream_t st) {  // the `reset` arm (state.rs:78-88)
        kind = new_kind;
        ring_capacity = capacity;
        ppc = new_ppc;
        write_slot = col_count = 0;
        ring.release();
        slot_counts.release();
        ring.reserve(ring_bytes(capacity));
        OMX_HIP(hipMemsetAsync(ring.ptr, 0, ring_bytes(capacity), st));
        if (kind == OMX_COLUMN_REASSIGNED) {
            slot_counts.reserve((size_t)n_streams * capacity);
            OMX_HIP(hipMemsetAsync(slot_counts.ptr, 0, (size_t)n_streams * capacity * sizeof(uint32_t), st));
        }
        set_fit(1, st);
    }

    void remap(uint32_t start, uint32_t keep, uint32_t capacity, hipStream_t st) {  // remap_retained + the resize copy plan
        DeviceBuffer<unsigned char> bigger;
        DeviceBuffer<uint32_t> counts;
        bigger.reserve(ring_bytes(capacity));
        OMX_HIP(hipMemsetAsync(bigger.ptr, 0, ring_bytes(capacity), st));
        if (kind == OMX_COLUMN_REASSIGNED) {
            counts.reserve((size_t)n_streams * capacity);
            OMX_HIP(hipMemsetAsync(counts.ptr, 0, (size_t)n_streams * capacity * sizeof(uint32_t), st));
        }
        HistoryRemapArgs a{};
        a.old_ring = ring.ptr;
        a.new_ring = bigger.ptr;
        a.old_counts = kind == OMX_COLUMN_REASSIGNED ? slot_counts.ptr : nullptr;
        a.new_counts = counts.ptr;
        a.n_streams = n_streams;
        a.old_slots = ring_capacity;
        a.new_slots = capacity;
        a.start = start;
        a.keep = keep;
        a.stride_bytes = stride_elems() * elem();
        launch_history_remap(a, st);
        OMX_HIP(hipStreamSynchronize(st));  // the old buffers are freed below
        std::swap(ring.ptr, bigger.ptr);
        std::swap(ring.count, bigger.count);
        std::swap(slot_counts.ptr, counts.ptr);
        std::swap(slot_counts.count, counts.count);
    }

    // state.rs:66-123 for `n_cols` device-resident columns: src [n_streams][n_cols][src_stride], counts [n_streams][n_cols]
    int apply(uint64_t fft_size, uint64_t history_length, bool reset, uint32_t update_kind, uint64_t n_cols, const void* src,
              const uint32_t* src_counts, uint32_t src_stride, hipStream_t st) {
        last_stream = st;
        const uint64_t ppc64 = fft_size / 2 + 1;
        if (ppc64 == 0 || ppc64 > 0xFFFFFFFFull) return OMX_NONE;
        const uint32_t new_kind = n_cols ? update_kind : kind;
        const uint32_t capacity = (uint32_t)history_columns(new_kind, (uint32_t)ppc64, history_length);
        if (capacity == 0) return OMX_NONE;
        if (reset || (uint32_t)ppc64 != ppc || new_kind != kind || !ring.ptr) {
            rebuild(new_kind, capacity, (uint32_t)ppc64, st);
        } else if (capacity != ring_capacity) {
            if (capacity > ring_capacity && col_count >= ring_capacity) {
                remap(write_slot, col_count, capacity, st);
                write_slot = col_count % capacity;
            } else if (capacity < ring_capacity && col_count >= capacity) {
                const uint32_t oldest_kept = (write_slot + ring_capacity - capacity) % ring_capacity;
                remap(oldest_kept, capacity, capacity, st);
                col_count = capacity;
                write_slot = 0;
            } else {
                remap(0, ring_capacity, capacity, st);  // identity plan
            }
            ring_capacity = capacity;
        }
        if (n_cols) {
            HistoryScatterArgs a{};
            a.src = static_cast<const unsigned char*>(src);
            a.src_counts = kind == OMX_COLUMN_REASSIGNED ? src_counts : nullptr;
            a.ring = ring.ptr;
            a.slot_counts = kind == OMX_COLUMN_REASSIGNED ? slot_counts.ptr : nullptr;
            a.n_streams = n_streams;
            a.n_cols = (uint32_t)n_cols;
            a.first_col = n_cols > ring_capacity ? (uint32_t)(n_cols - ring_capacity) : 0u;  // older ones would be overwritten
            a.src_stride = src_stride;
            a.ring_stride = stride_elems();
            a.ring_slots = ring_capacity;
            a.slot0 = write_slot;
            a.elem = elem();
            launch_history_scatter(a, st);
            write_slot = (uint32_t)((write_slot + n_cols) % ring_capacity);
            col_count = (uint32_t)std::min<uint64_t>((uint64_t)col_count + n_cols, ring_capacity);
        }
        if (kind == OMX_COLUMN_REASSIGNED) launch_history_fit(slot_counts.ptr, ring_capacity, fit_state.ptr, st);
        else set_fit(1, st);
        OMX_HIP(hipGetLastError());
        return OMX_NONE;
    }
};

extern "C" {

int omx_spectrogram_history_create(uint32_t n_streams, omx_spectrogram_history** out) {
    if (!out || n_streams == 0 || n_streams > 65535) return OMX_ERR_INVALID;
    const int rc = device_ready();
    if (rc < 0) return rc;
    return guarded([&] {
        *out = new omx_spectrogram_history(n_streams);
        return (int)OMX_NONE;
    });
}
void omx_spectrogram_history_destroy(omx_spectrogram_history* h) { delete h; }

int omx_spectrogram_history_apply(omx_spectrogram_history* h, const omx_spectrogram_update* up) {
    if (!h || !up || h->n_streams != 1 || (up->n_columns && !up->column_offsets)) return OMX_ERR_INVALID;
    return guarded([&] {
        // stage the columns in the bank layout [1][n_columns][ppc] + counts
        const uint64_t ppc = up->fft_size / 2 + 1, n = up->n_columns;
        const bool reassigned = up->kind == OMX_COLUMN_REASSIGNED;
        const size_t elem = reassigned ? sizeof(omx_spectrogram_point) : 2;
        const uint64_t stride = reassigned ? ppc : ((ppc + 1) / 2) * 2;
        std::vector<unsigned char> host((size_t)(n * stride * elem), 0);
        std::vector<uint32_t> counts((size_t)n, 0);
        const unsigned char* base = reassigned ? reinterpret_cast<const unsigned char*>(up->points) : reinterpret_cast<const unsigned char*>(up->codes);
        for (uint64_t c = 0; c < n; ++c) {
            const uint64_t lo = up->column_offsets[c], len = up->column_offsets[c + 1] - lo;
            counts[c] = (uint32_t)len;
            const uint64_t written = std::min<uint64_t>(len, stride);
            if (written) std::memcpy(host.data() + (size_t)(c * stride * elem), base + lo * elem, (size_t)(written * elem));
        }
        hipStream_t st = nullptr;
        if (n) {
            h->staging.reserve(host.size());
            h->staging_counts.reserve(counts.size());
            OMX_HIP(hipMemcpyAsync(h->staging.ptr, host.data(), host.size(), hipMemcpyHostToDevice, st));
            OMX_HIP(hipMemcpyAsync(h->staging_counts.ptr, counts.data(), counts.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            OMX_HIP(hipStreamSynchronize(st));
        }
        // a classic column of zero length is "not uploaded" in the reference (render.rs:586): those do not occur on processor
        // output (every classic column carries fft_size / 2 + 1 codes)
        return h->apply(up->fft_size, up->history_length, up->reset != 0, up->kind, n, h->staging.ptr, h->staging_counts.ptr,
                        (uint32_t)stride, st);
    });
}

int omx_spectrogram_bank_history_apply(omx_spectrogram_history* h, const omx_spectrogram_bank_update* up, void* stream) {
    if (!h || !up || up->n_streams != h->n_streams) return OMX_ERR_INVALID;
    return guarded([&] {
        const bool reassigned = up->kind == OMX_COLUMN_REASSIGNED;
        return h->apply(up->fft_size, up->history_length, up->reset != 0, up->kind, up->n_columns,
                        reassigned ? static_cast<const void*>(up->d_points) : static_cast<const void*>(up->d_codes), up->d_counts,
                        (uint32_t)up->column_stride, static_cast<hipStream_t>(stream));
    });
}

int omx_spectrogram_history_get_info(omx_spectrogram_history* h, omx_spectrogram_history_info* out) {
    if (!h || !out) return OMX_ERR_INVALID;
    return guarded([&] {
        uint32_t pps = 1;
        if (h->fit_state.ptr) {
            OMX_HIP(hipMemcpyAsync(&pps, h->fit_state.ptr, sizeof(pps), hipMemcpyDeviceToHost, h->last_stream));
            OMX_HIP(hipStreamSynchronize(h->last_stream));
        }
        *out = omx_spectrogram_history_info{h->kind, h->ring_capacity, h->write_slot, h->col_count, h->ppc, pps, h->newest_slot(),
                                            h->visible_slots()};
        return (int)OMX_NONE;
    });
}

int64_t omx_spectrogram_history_slot_counts(omx_spectrogram_history* h, uint64_t stream_index, uint32_t* out, uint64_t capacity) {
    if (!h || !out || stream_index >= h->n_streams) return OMX_ERR_INVALID;
    const int rc = guarded([&] {
        const uint64_t n = std::min<uint64_t>(capacity, h->ring_capacity);
        if (n && h->kind == OMX_COLUMN_REASSIGNED) {
            OMX_HIP(hipMemcpyAsync(out, h->slot_counts.ptr + stream_index * h->ring_capacity, n * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                   h->last_stream));
            OMX_HIP(hipStreamSynchronize(h->last_stream));
        }
        return (int)OMX_NONE;
    });
    return rc < 0 ? rc : (int64_t)h->ring_capacity;
}

int omx_spectrogram_history_fetch_slot(omx_spectrogram_history* h, uint64_t stream_index, uint32_t slot, void* dst, uint64_t cap,
                                       uint64_t* n_out) {
    if (!h || !dst || stream_index >= h->n_streams || slot >= h->ring_capacity) return OMX_ERR_INVALID;
    return guarded([&] {
        uint64_t n = h->ppc;
        if (h->kind == OMX_COLUMN_REASSIGNED) {
            uint32_t c = 0;
            OMX_HIP(hipMemcpyAsync(&c, h->slot_counts.ptr + stream_index * h->ring_capacity + slot, sizeof(c), hipMemcpyDeviceToHost,
                                   h->last_stream));
            OMX_HIP(hipStreamSynchronize(h->last_stream));
            n = std::min<uint64_t>(c, h->ppc);
        }
        const uint64_t take = std::min(n, cap);
        if (take) {
            const unsigned char* src = h->ring.ptr + (stream_index * h->ring_capacity + slot) * (uint64_t)h->stride_elems() * h->elem();
            OMX_HIP(hipMemcpyAsync(dst, src, take * h->elem(), hipMemcpyDeviceToHost, h->last_stream));
            OMX_HIP(hipStreamSynchronize(h->last_stream));
        }
        if (n_out) *n_out = n;
        return (int)OMX_NONE;
    });
}

int omx_spectrogram_history_splat(omx_spectrogram_history* h, float reassigned_power_scale, const omx_splat_view* view, int on_device,
                                  void* stream, float* accum, float* db) {
    if (!h || !view || !accum || view->width == 0 || view->height == 0 || !(view->scale_factor >= 1.0f) ||
        view->freq_scale > OMX_FREQ_SCALE_ERB)
        return OMX_ERR_INVALID;
    return guarded([&] {
        hipStream_t st = static_cast<hipStream_t>(stream);
        SplatArgs a{};
        const bool drawable = h->kind == OMX_COLUMN_REASSIGNED && h->ring.ptr && h->visible_slots() > 0;
        a.points = reinterpret_cast<const omx_spectrogram_point*>(h->ring.ptr);
        a.counts = h->slot_counts.ptr;
        a.n_streams = h->n_streams;
        a.n_columns = drawable ? h->visible_slots() : 0;   // the visible slots, walked oldest -> newest
        a.column_stride = h->ppc;
        a.ring_slots = h->ring_capacity;
        // oldest visible column: age visible - 1 = (newest + hl - slot) % hl  =>  slot = (newest + hl - (visible - 1)) % hl
        a.slot0 = drawable ? (h->newest_slot() + h->ring_capacity - (h->visible_slots() - 1)) % h->ring_capacity : 0;
        a.width = view->width;
        a.height = view->height;
        a.freq_scale = view->freq_scale;
        a.extent_x = view->extent_x;
        a.extent_y = view->extent_y;
        a.scale_factor = view->scale_factor;
        const float lo = freq_scale_host(view->freq_scale, view->freq_min), hi = freq_scale_host(view->freq_scale, view->freq_max);
        a.axis_lo = lo;
        a.axis_inv = 1.0f / std::fmax(hi - lo, 1e-12f);
        a.uv_lo = view->uv_lo;
        a.inv_uv = 1.0f / std::fmax(view->uv_hi - view->uv_lo, 1e-12f);
        a.tilt_db = view->tilt_db;
        const size_t px = (size_t)h->n_streams * view->width * view->height;
        static const int form = [] {
            const char* e = getenv("OMX_SPLAT_FORM");
            return e ? atoi(e) : 0;
        }();
        if (on_device) {
            a.accum = accum;
            launch_splat(a, db, reassigned_power_scale, st, form);
            OMX_HIP(hipGetLastError());
            return (int)OMX_PRODUCED;
        }
        DeviceBuffer<float> d_accum, d_db;
        d_accum.reserve(px);
        if (db) d_db.reserve(px);
        a.accum = d_accum.ptr;
        launch_splat(a, db ? d_db.ptr : nullptr, reassigned_power_scale, st, form);
        OMX_HIP(hipGetLastError());
        OMX_HIP(hipMemcpyAsync(accum, d_accum.ptr, px * sizeof(float), hipMemcpyDeviceToHost, st));
        if (db) OMX_HIP(hipMemcpyAsync(db, d_db.ptr, px * sizeof(float), hipMemcpyDeviceToHost, st));
        OMX_HIP(hipStreamSynchronize(st));
        return (int)OMX_PRODUCED;
    });
}

}  // extern "C"
