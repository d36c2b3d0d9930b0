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
// G consecutive hops walk together (G = 4 or 8 lanes = 1 or 2 quads): every quad of a group loads the SAME tile — identical addresses
// in one load instruction are one request to L2 — and broadcasts inside itself, so a group costs ((G - 1) hop + W) dependent adds per
// lane and fetches each sample once per G hops.  Quads alone (G = 4) re-read every sample (W + 3 hop) / (4 hop) = 4.75 times at
// 4096 / 256 (311 MB per launch: the neighbour quad read it four tiles earlier and the XCD's 128 wavefronts had moved 8 MB through its
// 4 MB L2 meanwhile); G = 8: 188 MB and 5888 steps instead of 4864 (launch_window_sums has the measurements).
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

// The aligned kernel's tile load: positions are multiples of 4 samples there (host-checked), so no 16-byte piece straddles the ring's end
// and every load wraps on its own — sixteen loads, no branch (with the wave-uniform branch above in its loop hipcc kept the NEXT tile's
// loads next to the adds that free their registers instead of ahead of the whole tile).
typedef float f4a __attribute__((ext_vector_type(4), aligned(16)));
__device__ __forceinline__ void load_tile_aligned(Tile& t, const float* ring, uint32_t mask, uint32_t pos, unsigned jq) {
#pragma unroll
    for (int i = 0; i < kTileLoads; ++i) {
        const f4a v = *reinterpret_cast<const f4a*>(ring + ((pos + 16u * (unsigned)i + 4u * jq) & mask));
        t.v[i] = f4u{v.x, v.y, v.z, v.w};
    }
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
// the 16 steps of one load group (the tile's samples 16 i ... 16 i + 15)
__device__ __forceinline__ float add_group(float sum, const f4u& g) {
#define OMX_Q(Q) \
    "v_add_f32_dpp %0, %1, %0 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "v_add_f32_dpp %0, %2, %0 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "v_add_f32_dpp %0, %3, %0 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "v_add_f32_dpp %0, %4, %0 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
    asm volatile("s_nop 1\n" OMX_Q(0) OMX_Q(1) OMX_Q(2) OMX_Q(3) : "+v"(sum) : "v"(g.x), "v"(g.y), "v"(g.z), "v"(g.w));
#undef OMX_Q
    return sum;
}
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

// The walk for shapes whose events fall on the tile's 16-step load groups (hop and W multiples of 16: every size the fused transform
// kernels serve at every hop a GUI offers): a tile is always consumed whole, group by group, and window starts / ends are handled
// between two groups — no partially consumed tiles, no 256-way predicated unroll beside the hot path (with it in the same kernel hipcc
// ran out of its 256 VGPRs and parked the NEXT tile's loads in front of every group of adds).
template <int G>
__global__ __launch_bounds__(64) void window_sums_aligned_kernel(WindowSumArgs a) {
    const uint32_t waves_per_sr = (a.n_hops + 63u) / 64u;
    const uint32_t xcd = blockIdx.x & 7u, bq = blockIdx.x >> 3;
    const uint32_t chunk = bq % waves_per_sr, sr = (bq / waves_per_sr) * 8u + xcd;
    if (sr >= a.n_streams * a.n_rings) return;
    const uint32_t s = sr / a.n_rings, r = sr % a.n_rings;
    const uint32_t n_hops_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.hops ? a.hops[s] : a.n_hops));
    if (chunk * 64u >= n_hops_s) return;
    const uint32_t lane = threadIdx.x, j = lane % (uint32_t)G, group = lane / (uint32_t)G, jq = lane & 3u;
    const uint32_t live = min((uint32_t)G, n_hops_s - chunk * 64u);
    const uint32_t hop = a.hop, W = a.window;
    const uint32_t mask = (uint32_t)(a.cap - 1u);
    const float* ring = a.ring[r] + (uint64_t)s * a.cap;
    const uint64_t tail = a.tails ? a.tails[s] : a.tail;
    const uint32_t h_group = chunk * 64u + group * (uint32_t)G;
    const uint32_t p0 = (uint32_t)(tail + (uint64_t)(a.first_hop + h_group) * hop);

    float sum = -0.0f, taken = 0.0f;
    uint32_t step = 0, resets = 1, takes = 0;
    uint32_t next_take = W, next_reset = live > 1u ? hop : 0xFFFFFFFFu;   // (steps fit 32 bits: (G - 1) hop + W < 2^32 is host-checked)
    uint32_t next_event = min(next_take, next_reset);
    // the events at `step` (rare: 2 G of them in a walk of hundreds of groups); false once the last window has been taken.
    // hop <= W (host-checked): from lane 0's start to the last take some window is always open — no gaps to skip.
    auto events = [&]() -> bool {
        while (step == next_take) {
            taken = j == takes ? sum : taken;
            ++takes;
            if (takes == live) return false;
            next_take = takes * hop + W;
        }
        while (step == next_reset) {
            sum = j == resets ? -0.0f : sum;
            ++resets;
            next_reset = resets < live ? resets * hop : 0xFFFFFFFFu;
        }
        next_event = min(next_take, next_reset);
        return true;
    };
    // one tile, group by group.  THREE tile buffers in rotation: `far` receives the loads of the tile after next before this tile's adds
    // start, so a load has two tiles of adds (~2.4 us) to land.  With one tile ahead the walk ran at the memory latency of a chip with
    // 1024 wavefronts x 16 KiB in flight — ~3 us per tile whatever the group size, 65 ... 93 us per launch where the adds take 20 ... 38:
    // the ring has left the caches behind the spectrogram kernel's 1.7 GB.  (One wavefront per SIMD: 200 VGPRs cost nothing.)
    auto walk_tile = [&](const Tile& cur, Tile& far) -> bool {
        load_tile_aligned(far, ring, mask, p0 + step + 2u * (uint32_t)kTileSteps, jq);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < kTileLoads; ++i) {
            if (__builtin_expect(step == next_event, 0)) {
                if (!events()) return false;
            }
            sum = add_group(sum, cur.v[i]);
            step += 16u;
        }
        return true;
    };
    Tile t0, t1, t2;
    load_tile_aligned(t0, ring, mask, p0, jq);
    load_tile_aligned(t1, ring, mask, p0 + (uint32_t)kTileSteps, jq);
    while (walk_tile(t0, t2) && walk_tile(t1, t0) && walk_tile(t2, t1)) {
    }
    const uint32_t h = h_group + j;
    if (h < n_hops_s) a.sums[((uint64_t)s * a.n_rings + r) * a.n_hops + h] = taken;
}

// The general walk (any hop, any window length): tiles may be consumed partially, up to the next event.
// One wavefront = 64 / G groups = 64 consecutive hops of one (stream, ring).
template <int G>
__global__ __launch_bounds__(64) void window_sums_seq_kernel(WindowSumArgs a) {
    const uint32_t waves_per_sr = (a.n_hops + 63u) / 64u;
    // XCD-aware map, the consumers' (spectrum_power_pow2_kernel / stft_classic_pow2_kernel): a (stream, ring) lives on one XCD
    const uint32_t xcd = blockIdx.x & 7u, bq = blockIdx.x >> 3;
    const uint32_t chunk = bq % waves_per_sr, sr = (bq / waves_per_sr) * 8u + xcd;
    if (sr >= a.n_streams * a.n_rings) return;
    const uint32_t s = sr / a.n_rings, r = sr % a.n_rings;
    const uint32_t n_hops_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.hops ? a.hops[s] : a.n_hops));
    if (chunk * 64u >= n_hops_s) return;
    if (a.modes && a.modes[s] != kFoldWalk) return;  // ragged banks: this stream's folds are carried (window_sums_carry_kernel)
    const uint32_t lane = threadIdx.x, j = lane % (uint32_t)G, group = lane / (uint32_t)G, jq = lane & 3u;
    const uint32_t live = min((uint32_t)G, n_hops_s - chunk * 64u);  // lanes of the wavefront's fullest group (its first): the walk ends with their last take
    const uint32_t hop = a.hop, W = a.window;
    const uint32_t mask = (uint32_t)(a.cap - 1u);
    const float* ring = a.ring[r] + (uint64_t)s * a.cap;
    const uint64_t tail = a.tails ? a.tails[s] : a.tail;
    const uint32_t h_group = chunk * 64u + group * (uint32_t)G;  // first hop of the group (of this launch)
    const uint32_t p0 = (uint32_t)(tail + (uint64_t)(a.first_hop + h_group) * hop);  // the walk's step 0 (positions mod 2^32: cap <= 2^30 divides it)

    float sum = -0.0f, taken = 0.0f;
    uint32_t step = 0, resets = 1, takes = 0;  // lane 0's reset is the initial value
    // One tile of the walk: the events at `step`, then the next tile's loads (into `nxt`: in flight behind this tile's adds), then the adds
    // of `cur`.  Returns false when the last window has been taken.  Called with the two tile buffers in alternating roles — a
    // `cur = nxt` at the end of a loop body made hipcc wait for the NEW loads ahead of every group of adds (vmcnt(15) ... vmcnt(0) through
    // the sixteen groups): 65 us per launch at the ring's load latency where the adds take 20.
    auto walk_tile = [&](Tile& cur, Tile& nxt) -> bool {
        while (true) {
            // events at `step`: window ends first (a lane's take), then window starts
            while (takes < live && (uint64_t)takes * hop + W == step) {
                taken = j == takes ? sum : taken;
                ++takes;
            }
            if (takes == live) return false;
            while (resets < live && (uint64_t)resets * hop == step) {
                sum = j == resets ? -0.0f : sum;
                ++resets;
            }
            if (resets != takes) break;
            // no window open (hop > W): skip to the next start
            step = (uint32_t)((uint64_t)resets * hop);
            cur = load_tile(ring, mask, p0 + step, jq);
        }
        uint64_t next_event = (uint64_t)takes * hop + W;
        if (resets < live) next_event = min(next_event, (uint64_t)resets * hop);
        const uint32_t n = (uint32_t)min<uint64_t>(next_event - step, (uint64_t)kTileSteps);
        nxt = load_tile(ring, mask, p0 + step + n, jq);
        asm volatile("" ::: "memory");  // (the loads stay ahead of the adds: hipcc sinks them behind the chain otherwise, to save registers)
        if (n == (uint32_t)kTileSteps) sum = add_tile_full(sum, cur);
        else sum = add_tile_partial(sum, cur, n);
        step += n;
        return true;
    };
    Tile ta = load_tile(ring, mask, p0, jq), tb;
    while (walk_tile(ta, tb) && walk_tile(tb, ta)) {
    }
    const uint32_t h = h_group + j;
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
    if (a.modes && a.modes[s] != kFoldCarry) return;
    const uint64_t tail = a.tails ? a.tails[s] : a.tail, carry_pos = a.froms ? a.froms[s] : a.carry_pos, head = a.heads ? a.heads[s] : a.head;
    const uint32_t slot0 = a.slot0s ? a.slot0s[s] : a.slot0;
    const uint64_t n_windows = a.tails ? (head - tail + a.hop - 1u) / a.hop : a.n_windows;
    const uint32_t mask = (uint32_t)(a.cap - 1u);
    const float* ring = a.ring[r] + (uint64_t)s * a.cap;
    float* slot = a.carry + (uint64_t)sr * G + (slot0 + g) % G;  // window k = g, g + G, ... of this call all live in this slot
    for (uint64_t k = g; k < n_windows; k += G) {
        const uint64_t p = tail + k * a.hop, end = p + a.window;
        const uint64_t from = max(p, carry_pos), to = min(end, head);
        float sum = p < carry_pos ? *slot : -0.0f;
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
        if (end <= head) {
            if (k >= a.first_hop && k - a.first_hop < a.n_hops) a.sums[(uint64_t)sr * a.n_hops + (k - a.first_hop)] = sum;
        } else {
            *slot = sum;
        }
    }
}

void launch_window_sums_carry(const WindowCarryArgs& a, hipStream_t stream) {
    const uint64_t lanes = (uint64_t)a.n_streams * a.n_rings * a.slots;
    if (lanes == 0 || (a.n_windows == 0 && !a.tails)) return;
    window_sums_carry_kernel<<<(uint32_t)((lanes + 63u) / 64u), 64, 0, stream>>>(a);
}

void launch_window_sums(const WindowSumArgs& a, hipStream_t stream) {
    if (a.n_hops == 0 || a.n_streams == 0 || a.n_rings == 0) return;
    const uint32_t waves_per_sr = (a.n_hops + 63u) / 64u;
    const uint32_t sr = a.n_streams * a.n_rings;
    const uint32_t grid = ((sr + 7u) / 8u) * 8u * waves_per_sr;
    // hops per group: as many as overlap one sample (W / hop), between one quad and four
    const uint32_t overlap = a.hop ? a.window / a.hop : 1u;
    // Measured at 4096 / 256, 65536 hops (round 6, kernel alone, same box): G = 4 / 8 / 16 take 72 / 77 / 79 us and the step 1.733 ... 1.741 /
    // 1.732 ... 1.738 / 1.738 ... 1.739 ms — inside each other's noise: a group of 4 re-reads every sample 4.75 times (311 MB through an
    // L2 that cannot hold the reuse distance), a group of 16 reads it twice but walks 7936 steps instead of 4864, and the kernel is the
    // chain of dependent adds either way (a build without any load runs as long: 13 us + 4.6 ... 5.3 ns per step).  8 is the middle.
    const uint32_t g = (overlap >= 8u && a.n_hops >= 8u) ? 8u : 4u;
    // (lock-step launches whose positions are multiples of 4 samples: every 16-byte load then lies inside the ring)
    const bool aligned = a.hop % 16u == 0u && a.window % 16u == 0u && (uint64_t)(g - 1u) * a.hop + a.window < 0xFFFFFF00ull && a.hop <= a.window && !a.tails &&
                         (a.tail + (uint64_t)a.first_hop * a.hop) % 4u == 0u && a.cap % 4u == 0u && reinterpret_cast<uintptr_t>(a.ring[0]) % 16u == 0u &&
                         (a.n_rings < 2u || reinterpret_cast<uintptr_t>(a.ring[1]) % 16u == 0u);
    if (aligned) {
        if (g == 8u) window_sums_aligned_kernel<8><<<grid, 64, 0, stream>>>(a);
        else window_sums_aligned_kernel<4><<<grid, 64, 0, stream>>>(a);
    } else {
        window_sums_seq_kernel<4><<<grid, 64, 0, stream>>>(a);
    }
}

}  // namespace omx
