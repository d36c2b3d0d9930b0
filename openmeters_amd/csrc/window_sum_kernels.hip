// The DC-removed window's mean in the REFERENCE's order (gfx950 only).
//   window.rs:76-79  `head.iter().chain(tail).sum::<f32>() / len`: a sequential f32 fold over the window's samples.
// The fused classic / spectrum kernels used to take this sum as a tree (per-thread partials, DPP reduction, wave partials): on a hop
// whose constant offset dwarfs its signal the two orders differ by a few ulps of N / 2, and that residue times the window's DC gain
// sat in bins 0 ... 2 up to 8e-5 of the trace maximum away from the reference.  This pre-pass computes the sum
// of every (stream, ring, hop) window in the reference's order, bit for bit; the transform kernels take it as a per-hop scalar.
//
// A sequential f32 sum cannot be split or updated incrementally (every rounding depends on the whole prefix), so the work is W dependent
// adds per hop whatever one does; what can be chosen is how the samples reach the lanes.  One lane per hop reading its own window would
// touch 64 distinct cache lines per load instruction.  Here FOUR consecutive hops of one stream form a quad that walks the union of its
// four windows together: at step s every lane of the quad adds the SAME sample x[s] — lane j's window is steps [j hop, j hop + W), so
// lane j is reset to -0.0 when the walk reaches j hop (whatever it added before is discarded by the move) and its sum is taken when
// the walk reaches j hop + W (what it adds afterwards is never looked at).  The quad loads 256 consecutive samples with sixteen 16-byte
// loads per lane (lane j: samples 16 i + 4 j ... + 3), and a step is ONE instruction: v_add_f32 with a quad_perm DPP operand that broadcasts
// the owning lane's component — no LDS, no select, no per-step predicate.  Steps between one window's end and the next one's start
// (hop > W) are skipped.  Resets and takes are wave-uniform events (every quad of a launch has the same hop and W): scalar control flow.
//
// Cost: (3 hop + W) dependent adds per lane; cfg2's spectrum (2 traces x 64 streams x 1024 hops of 4096 / 256) is 2048 wavefronts of
// 4864 steps.
#include "stft_kernels.hpp"

namespace omx {

namespace {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // ring positions are arbitrary: 4-byte alignment only

// component C of lane Q of this lane's quad, as the DPP operand of the add that consumes it
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {
    constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);  // quad_perm:[Q,Q,Q,Q]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}

// A tile = 256 consecutive samples of the quad's walk: lane j holds samples 16 i + 4 j ... + 3 of it (i < 16), sixteen 16-byte loads.
// One tile ahead of the adds is ~1 us of cover (256 steps x 4 ns): with 64-sample tiles the walk was bound by the load latency of the
// freshly written ring (75 us per launch where the adds alone take 20).
constexpr int kTileLoads = 16, kTileSteps = 16 * kTileLoads;
struct Tile {
    f4u v[kTileLoads];
};

// Whether a tile wraps around the ring's end is decided per WAVEFRONT: a per-load branch (round 6, first build) made the number of loads
// in flight path-dependent, hipcc then waits with vmcnt(0) where the paths join — i.e. for the NEXT tile's loads before the current
// tile's adds — and the walk ran at the ring's load latency (73 us per launch where the adds take 20).
__device__ __forceinline__ Tile load_tile(const float* ring, uint32_t mask, uint32_t pos, unsigned j) {
    Tile t;
    const uint32_t o0 = pos & mask;
    if (__builtin_amdgcn_ballot_w64(o0 + (uint32_t)kTileSteps > mask + 1u) == 0) {  // every quad's 256 samples lie in one piece
        const float* base = ring + o0 + 4u * j;
#pragma unroll
        for (int i = 0; i < kTileLoads; ++i) t.v[i] = *reinterpret_cast<const f4u*>(base + 16 * i);
    } else {
#pragma unroll
        for (int i = 0; i < kTileLoads; ++i) {
            const uint32_t o = pos + 16u * (unsigned)i + 4u * j;
            t.v[i] = f4u{ring[o & mask], ring[(o + 1u) & mask], ring[(o + 2u) & mask], ring[(o + 3u) & mask]};
        }
    }
    return t;
}

// steps [0, n) of the tile; n is wave-uniform
__device__ __forceinline__ float add_tile_partial(float sum, const Tile& t, uint32_t n) {
#pragma unroll
    for (int i = 0; i < kTileLoads; ++i) {
        if ((uint32_t)(16 * i) >= n) break;
#define OMX_STEP(Q, C)                                                      \
    if ((uint32_t)(16 * i + 4 * Q + C) < n) sum = quad_bcast<Q>(t.v[i][C]) + sum;
        OMX_STEP(0, 0) OMX_STEP(0, 1) OMX_STEP(0, 2) OMX_STEP(0, 3)
        OMX_STEP(1, 0) OMX_STEP(1, 1) OMX_STEP(1, 2) OMX_STEP(1, 3)
        OMX_STEP(2, 0) OMX_STEP(2, 1) OMX_STEP(2, 2) OMX_STEP(2, 3)
        OMX_STEP(3, 0) OMX_STEP(3, 1) OMX_STEP(3, 2) OMX_STEP(3, 3)
#undef OMX_STEP
    }
    return sum;
}
// All steps of a tile, written as assembly: hipcc puts `s_nop 1` between consecutive v_add_f32_dpp of one accumulator (its DPP
// hazard check covers every VGPR the instruction reads), 13.6 cycles per step where the dependent add alone takes 9.6
// (tools/microbench/dpp_chain.hip: same sums without the nops).  The hazard the ISA names is a VALU write followed by a DPP READ of that
// register: the accumulator is the plain operand here, and the DPP-read tile registers are written by loads — or by a copy the
// compiler may have placed just ahead, which the leading s_nop 1 of every 16-step group covers.
__device__ __forceinline__ float add_tile_full(float sum, const Tile& t) {
#define OMX_Q(Q) \
    "v_add_f32_dpp %0, %1, %0 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "v_add_f32_dpp %0, %2, %0 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "v_add_f32_dpp %0, %3, %0 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "v_add_f32_dpp %0, %4, %0 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#pragma unroll
    for (int i = 0; i < kTileLoads; ++i) {
        asm volatile("s_nop 1\n" OMX_Q(0) OMX_Q(1) OMX_Q(2) OMX_Q(3) : "+v"(sum) : "v"(t.v[i].x), "v"(t.v[i].y), "v"(t.v[i].z), "v"(t.v[i].w));
    }
#undef OMX_Q
    return sum;
}

}  // namespace

// One wavefront = 16 quads = 64 consecutive hops of one (stream, ring).
__global__ __launch_bounds__(64) void window_sums_seq_kernel(WindowSumArgs a) {
    const uint32_t waves_per_sr = (a.n_hops + 63u) / 64u;
    // XCD-aware map, the consumers' (spectrum_power_pow2_kernel / stft_classic_pow2_kernel): a (stream, ring) lives on one XCD
    const uint32_t xcd = blockIdx.x & 7u, bq = blockIdx.x >> 3;
    const uint32_t chunk = bq % waves_per_sr, sr = (bq / waves_per_sr) * 8u + xcd;
    if (sr >= a.n_streams * a.n_rings) return;
    const uint32_t s = sr / a.n_rings, r = sr % a.n_rings;
    const uint32_t n_hops_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.hops ? a.hops[s] : a.n_hops));
    if (chunk * 64u >= n_hops_s) return;
    const uint32_t lane = threadIdx.x, j = lane & 3u, quad = lane >> 2;
    const uint32_t live = min(4u, n_hops_s - chunk * 64u);  // lanes of the wavefront's fullest quad (its first): the walk ends with their last take
    const uint32_t hop = a.hop, W = a.window;
    const uint32_t mask = (uint32_t)(a.cap - 1u);
    const float* ring = a.ring[r] + (uint64_t)s * a.cap;
    const uint64_t tail = a.tails ? a.tails[s] : a.tail;
    const uint32_t h_quad = chunk * 64u + quad * 4u;  // first hop of the quad (of this launch)
    const uint32_t p0 = (uint32_t)(tail + (uint64_t)(a.first_hop + h_quad) * hop);  // the walk's step 0 (positions mod 2^32: cap <= 2^30 divides it)

    float sum = -0.0f, taken = 0.0f;
    uint32_t step = 0, resets = 1, takes = 0;  // lane 0's reset is the initial value
    Tile cur = load_tile(ring, mask, p0, j);
    while (true) {
        // events at `step`: window ends first (a lane's take), then window starts
        while (takes < live && (uint64_t)takes * hop + W == step) {
            taken = j == takes ? sum : taken;
            ++takes;
        }
        if (takes == live) break;
        while (resets < live && (uint64_t)resets * hop == step) {
            sum = j == resets ? -0.0f : sum;
            ++resets;
        }
        uint64_t next_event = (uint64_t)takes * hop + W;
        if (resets < live) next_event = min(next_event, (uint64_t)resets * hop);
        if (resets == takes) {  // no window open (hop > W): skip to the next start
            step = (uint32_t)next_event;
            cur = load_tile(ring, mask, p0 + step, j);
            continue;
        }
        const uint32_t n = (uint32_t)min<uint64_t>(next_event - step, (uint64_t)kTileSteps);
        const Tile nxt = load_tile(ring, mask, p0 + step + n, j);  // in flight behind this tile's adds
        asm volatile("" ::: "memory");  // (the loads stay ahead of the adds: hipcc sinks them behind the chain otherwise, to save registers)
        if (n == (uint32_t)kTileSteps) sum = add_tile_full(sum, cur);
        else sum = add_tile_partial(sum, cur, n);
        step += n;
        cur = nxt;
    }
    const uint32_t h = h_quad + j;
    if (h < n_hops_s) a.sums[((uint64_t)s * a.n_rings + r) * a.n_hops + h] = taken;
}

// ---- the same sums for a caller that feeds a few samples per call (the reference's cadence: one 256-frame block per call, meter.rs:40-69).
// A hop that completes in such a call has had W - 256 of its samples in the ring for many calls; walking the whole window again costs
// W dependent adds of LATENCY per call (16384 / 1024, the reference's default spectrum: ~68 us, measured +60 us per ingest).  Instead
// every window that has STARTED keeps its running fold between calls: G = ceil(W / hop) slots per (stream, ring), one lane per slot.
// A call adds the samples [carry_pos, head) to every open window (all lanes of a group read the same addresses), windows that
// start inside that range begin at -0.0, windows that end inside it hand their sum to the transform kernel, the others park theirs
// in their slot.  The chain per call is the number of NEW samples, not W.  Same adds in the same order: bit-identical to the walk.
// (carry_pos == tail: nothing carried — every started window is folded from its first sample, which is still in the ring.)
struct CarryTile {
    f4u v[16];
};
__device__ __forceinline__ CarryTile load_carry_tile(const float* ring, uint32_t mask, uint32_t pos) {
    CarryTile t;
    const uint32_t o0 = pos & mask;
    if (__builtin_amdgcn_ballot_w64(o0 + 64u > mask + 1u) == 0) {  // (decided per wavefront: see load_tile)
        const float* base = ring + o0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t.v[i] = *reinterpret_cast<const f4u*>(base + 4 * i);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t o = pos + 4u * (unsigned)i;
            t.v[i] = f4u{ring[o & mask], ring[(o + 1u) & mask], ring[(o + 2u) & mask], ring[(o + 3u) & mask]};
        }
    }
    return t;
}

__global__ __launch_bounds__(64) void window_sums_carry_kernel(WindowCarryArgs a) {
    const uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    const uint32_t G = a.slots;
    if (idx >= a.n_streams * a.n_rings * G) return;
    const uint32_t sr = idx / G, g = idx % G;
    const uint32_t s = sr / a.n_rings, r = sr % a.n_rings;
    const uint32_t mask = (uint32_t)(a.cap - 1u);
    const float* ring = a.ring[r] + (uint64_t)s * a.cap;
    float* slot = a.carry + (uint64_t)sr * G + (a.slot0 + g) % G;  // window k = g, g + G, ... of this call all live in this slot
    for (uint64_t k = g; k < a.n_windows; k += G) {
        const uint64_t p = a.tail + k * a.hop, end = p + a.window;
        const uint64_t from = max(p, a.carry_pos), to = min(end, a.head);
        float sum = p < a.carry_pos ? *slot : -0.0f;
        uint32_t pos = (uint32_t)from;        // (positions mod 2^32: cap <= 2^30 divides it)
        uint64_t n = to - from;
        if (n >= 64u) {
            CarryTile cur = load_carry_tile(ring, mask, pos);
            while (n >= 64u) {
                const CarryTile nxt = load_carry_tile(ring, mask, pos + 64u);  // in flight behind this tile's adds (ring memory: always readable)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    sum += cur.v[i].x;
                    sum += cur.v[i].y;
                    sum += cur.v[i].z;
                    sum += cur.v[i].w;
                }
                cur = nxt;
                pos += 64u;
                n -= 64u;
            }
        }
        if (n) {
            const CarryTile last = load_carry_tile(ring, mask, pos);
            const uint32_t m = (uint32_t)n;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (4u * (unsigned)i + 0u < m) sum += last.v[i].x;
                if (4u * (unsigned)i + 1u < m) sum += last.v[i].y;
                if (4u * (unsigned)i + 2u < m) sum += last.v[i].z;
                if (4u * (unsigned)i + 3u < m) sum += last.v[i].w;
            }
        }
        if (end <= a.head) {
            if (k >= a.first_hop && k - a.first_hop < a.n_hops) a.sums[(uint64_t)sr * a.n_hops + (k - a.first_hop)] = sum;
        } else {
            *slot = sum;
        }
    }
}

void launch_window_sums_carry(const WindowCarryArgs& a, hipStream_t stream) {
    const uint64_t lanes = (uint64_t)a.n_streams * a.n_rings * a.slots;
    if (lanes == 0 || a.n_windows == 0) return;
    window_sums_carry_kernel<<<(uint32_t)((lanes + 63u) / 64u), 64, 0, stream>>>(a);
}

void launch_window_sums(const WindowSumArgs& a, hipStream_t stream) {
    if (a.n_hops == 0 || a.n_streams == 0 || a.n_rings == 0) return;
    const uint32_t waves_per_sr = (a.n_hops + 63u) / 64u;
    const uint32_t sr = a.n_streams * a.n_rings;
    const uint32_t grid = ((sr + 7u) / 8u) * 8u * waves_per_sr;
    window_sums_seq_kernel<<<grid, 64, 0, stream>>>(a);
}

}  // namespace omx
