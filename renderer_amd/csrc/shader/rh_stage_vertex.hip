// Vertex-stage code object: compiled once per vertex source with
//     hipcc --genco --offload-arch=gfx950 -ffp-contract=off -DRH_VERTEX_SOURCE='"<file>.vert.hip"' rh_stage_vertex.hip
// Exports rh_bin_count and rh_bin_pairs: both run the vertex function for every cluster (pair of consecutive triangles
// of one instance) and find the screen tiles the cluster's bounding box touches; the first only records and counts them
// (the form that waits for the count), the second writes the (tile, cluster) pair array in the same launch.
#include "rh_shader.h"
#include RH_VERTEX_SOURCE
#include "rh_geometry.h"

extern "C" __device__ const rh_module_info_t rh_module_info = {
    RH_MODULE_MAGIC, RH_DEVICE_ABI_VERSION, RH_MODULE_STAGE_VERTEX, (uint32_t)rh::N_VARYINGS, 0u,
    rh::fnv1a(RH_VERTEX_TAG), 0u, 0u, 0u
};

namespace {

// Bounding box of the part of triangle (i0, i1, i2) of `instance` inside the clip volume, grown into (bx0 .. by1).
// Returns the number of the polygon's vertices.
__device__ __noinline__ int clipped_bbox(const rh_draw_params& p, uint32_t instance, uint32_t i0, uint32_t i1, uint32_t i2,
                                         int& bx0, int& by0, int& bx1, int& by1) {
    rh::ClipPolygon poly;
    rh::clip_triangle(p, instance, i0, i1, i2, poly);
    for (int i = 2; i < poly.n; i++) {
        int x0, y0, x1, y1;
        if (rh::tri_pixel_bbox(p, poly.v[0], poly.v[i - 1], poly.v[i], x0, y0, x1, y1)) { bx0 = min(bx0, x0); by0 = min(by0, y0); bx1 = max(bx1, x1); by1 = max(by1, y1); }
    }
    return poly.n;
}

// The inclusive rectangle of tiles the bounding box of `cluster` touches; returns the number of (tile, cluster) pairs
// the cluster makes (0: none).  CLIP (the clipping kernels): a cluster with a triangle that needs clipping makes
// one pair per tile and sub-record, the numbers of sub-records packed above tx0 (see rh_tile_range).  Otherwise such a triangle is left out: the pass has
// to run again with the clipping kernels.  Either way `wants_clip` is set when a triangle of the cluster crosses a clip
// plane (the host goes back to the usual kernels when a pipeline's passes have not met one for a while).
// `lean` (out): the cluster is a lean quad (rh_geometry.h): the tile kernel sets it up itself.
template <bool CLIP>
__device__ __forceinline__ uint32_t cluster_tiles(const rh_draw_params& p, uint32_t cluster, uint32_t& tx0, uint32_t& ty0, uint32_t& tx1, uint32_t& ty1, bool& wants_clip, bool& lean) {
    lean = false;
    const uint32_t instance = cluster / p.cpi, local = cluster - instance * p.cpi;
    int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
    const uint32_t tri0 = local * 2u;
    const bool second = tri0 + 1u < p.tpi;
    bool clipped = false;
    uint32_t subs[2] = { 0u, 0u };              // sub-records of the two triangles, should the cluster turn out clipped
    auto triangle = [&](int which, const rh::SnappedVertex& a, const rh::SnappedVertex& b, const rh::SnappedVertex& c, uint32_t i0, uint32_t i1, uint32_t i2) {
        int x0, y0, x1, y1;
        const int kind = rh::classify(a, b, c);
        if (kind == rh::TRI_WHOLE) {
            subs[which] = 1u;
            if (rh::tri_pixel_bbox(p, a, b, c, x0, y0, x1, y1)) { bx0 = min(bx0, x0); by0 = min(by0, y0); bx1 = max(bx1, x1); by1 = max(by1, y1); }
        } else if (kind == rh::TRI_CLIPPED) {
            clipped = true;
            if (CLIP) {
                const int n = clipped_bbox(p, instance, i0, i1, i2, bx0, by0, bx1, by1);
                subs[which] = n >= 3 ? (uint32_t)(n - 1) / 2u : 0u;         // two triangles of the fan of n - 2 per sub-record
            }
        }
    };
    if (p.topology == 1u) {
        // strip: the two triangles share two of their four vertices
        const rh::SnappedVertex v0 = rh::run_vertex(p, tri0, instance), v1 = rh::run_vertex(p, tri0 + 1u, instance), v2 = rh::run_vertex(p, tri0 + 2u, instance);
        triangle(0, v0, v1, v2, tri0, tri0 + 1u, tri0 + 2u);
        if (second) {
            const rh::SnappedVertex v3 = rh::run_vertex(p, tri0 + 3u, instance);
            triangle(1, v1, v3, v2, tri0 + 1u, tri0 + 3u, tri0 + 2u);
            bool shape;
            lean = rh::lean_quad(p, v0, v1, v2, v3, shape);
        }
    } else {
        for (uint32_t k = 0; k < (second ? 2u : 1u); k++) {
            const uint32_t first = 3u * (tri0 + k);
            const rh::SnappedVertex v0 = rh::run_vertex(p, first, instance), v1 = rh::run_vertex(p, first + 1u, instance), v2 = rh::run_vertex(p, first + 2u, instance);
            triangle((int)k, v0, v1, v2, first, first + 1u, first + 2u);
        }
    }
    if (clipped) wants_clip = true;
    if (!CLIP) clipped = false;
    tx0 = 1u; ty0 = 0u; tx1 = 0u; ty1 = 0u;             // x0 = 1 > x1 = 0: empty
    if (bx1 < bx0 || by1 < by0) return 0u;
    tx0 = (uint32_t)bx0 / RH_TILE_W; ty0 = (uint32_t)by0 / RH_TILE_H;
    tx1 = (uint32_t)bx1 / RH_TILE_W; ty1 = (uint32_t)by1 / RH_TILE_H;
    const uint32_t tiles = (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
    if (!clipped) return tiles;
    tx0 |= RH_RANGE_SUBS(subs[0], subs[1]);
    return tiles * (subs[0] + subs[1]);
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t n = __shfl_up(v, off);
        if (lane >= off) v += n;
    }
    return v;
}

constexpr unsigned long long STATUS_VALUE = (1ull << 62) - 1ull;

} // namespace

// One thread per cluster.  Writes the cluster's inclusive tile rectangle and the per-workgroup sum of tile counts
// (the scan of those sums gives every workgroup its base offset in the (tile, cluster) pair array).
template <bool CLIP>
__device__ __forceinline__ void bin_count(const rh_bin_count_args& a) {
    const rh_draw_params& p = a.p;
    const uint32_t cluster = blockIdx.x * 256u + threadIdx.x;
    uint32_t count = 0;
    if (cluster < p.n_clusters) {
        rh_tile_range r; uint32_t tx0, ty0, tx1, ty1;
        bool wants_clip = false, lean = false;
        count = cluster_tiles<CLIP>(p, cluster, tx0, ty0, tx1, ty1, wants_clip, lean);
        r.lo = tx0 | (ty0 << 16); r.hi = tx1 | (ty1 << 16) | (lean ? 0u : RH_RANGE_GENERAL);
        a.ranges[cluster] = r;
        if (wants_clip) *a.wants_clip = 1u;
    }
    // workgroup sum: wave reduce, then 4 partials through LDS
    __shared__ uint32_t partial[4];
    uint32_t s = count;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63u) == 0) partial[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.block_sums[blockIdx.x] = partial[0] + partial[1] + partial[2] + partial[3];
}
extern "C" __global__ __launch_bounds__(256) void rh_bin_count(rh_bin_count_args a) { bin_count<false>(a); }
extern "C" __global__ __launch_bounds__(256) void rh_bin_count_clip(rh_bin_count_args a) { bin_count<true>(a); }

constexpr uint32_t HUGE_SLOTS = 8u;

// The pairs of one cluster (tile rectangle lo..hi as in rh_tile_range, row-major; a clipped cluster: its sub-records of a
// tile adjacent) written from position `o` of the pair array by `threads` threads, this one being number `t`.  The
// digits of the tile numbers are counted on the way; the lanes of a wave mostly count the same value of the higher
// digits, and adds to one LDS word go one at a time: one lane then adds for all.
__device__ __forceinline__ void write_cluster_pairs(const rh_bin_pairs_args& a, uint32_t (*digits)[256], uint32_t lo, uint32_t hi, uint32_t cluster,
                                                    unsigned long long o, uint32_t t, uint32_t threads) {
    const uint32_t sa = RH_RANGE_SUBS_A(lo), sb = RH_RANGE_SUBS_B(lo), per_tile = sa + sb ? sa + sb : 1u;
    const uint32_t x0 = RH_RANGE_X0(lo), y0 = lo >> 16, x1 = RH_RANGE_X1(hi), y1 = hi >> 16;
    const uint32_t w = x1 - x0 + 1u, n = w * (y1 - y0 + 1u) * per_tile;
    for (uint32_t i = t; i < n; i += threads) {
        if (o + i >= a.capacity) break;
        const uint32_t ti = i / per_tile, k = i - ti * per_tile, sub = k < sa ? k : (sa + sb ? 4u + (k - sa) : 0u);
        const uint32_t tile = (y0 + ti / w) * a.p.tiles_x + (x0 + ti % w);
        a.keys[o + i] = tile; a.vals[o + i] = cluster | (sub << rh::CLIP_SUB_SHIFT);
        for (uint32_t k = 0; k < a.sort_passes; k++) {
            const uint32_t d = (tile >> (8u * k)) & 0xffu;
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
            const unsigned long long same = __ballot(d == d0), active = __ballot(true);
            if (same == active) { if ((uint32_t)__builtin_ctzll(active) == (threadIdx.x & 63u)) atomicAdd(&digits[k][d0], (uint32_t)__popcll(active)); }
            else atomicAdd(&digits[k][d], 1u);
        }
    }
}

// The pair array in one launch (see rh_bin_pairs_args).  Workgroup `part` (its ticket) owns clusters
// [RH_PAIR_PART * part, RH_PAIR_PART * (part + 1)), every thread RH_PAIR_PER_THREAD consecutive ones; the pairs of a
// cluster are its tiles in row-major order, clusters in order.  (Large parts: every workgroup costs a ticket, a slot in
// the look-back chain and an add per digit value, and adds to one word go one at a time.)
template <bool CLIP>
__device__ __forceinline__ void bin_pairs(const rh_bin_pairs_args& a) {
    const rh_draw_params& p = a.p;
    constexpr uint32_t PART = RH_PAIR_PART, THREADS = RH_PAIR_THREADS, PER = RH_PAIR_PER_THREAD, WAVES = THREADS / 64u;
    static_assert(PART == THREADS * PER, "part size");
    __shared__ uint32_t s_part, s_wants_clip, s_huge_n;
    __shared__ uint4 s_huge[HUGE_SLOTS];
    __shared__ unsigned long long s_huge_off[HUGE_SLOTS];
    __shared__ uint32_t wave_sum[WAVES];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t digits[RH_SORT_MAX_PASSES][256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (a.ticketed && threadIdx.x == 0) s_part = atomicAdd(&a.state->ticket[0], 1u);
    if (threadIdx.x == 0) { s_wants_clip = 0u; s_huge_n = 0u; }
    if (threadIdx.x < 256u) for (uint32_t k = 0; k < a.sort_passes; k++) digits[k][threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t part = a.ticketed ? s_part : blockIdx.x, n_parts = (p.n_clusters + PART - 1u) / PART;
    if (part >= n_parts) return;
    const uint32_t cluster0 = part * PART + threadIdx.x * PER;
    uint32_t tx0[PER], ty0[PER], tx1[PER], ty1[PER], cnt[PER], value[PER], count = 0u;
    bool wants_clip = false;
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
        tx0[j] = 1u; ty0[j] = 0u; tx1[j] = 0u; ty1[j] = 0u; cnt[j] = 0u;
        bool lean = false;
        if (cluster0 + j < p.n_clusters) cnt[j] = cluster_tiles<CLIP>(p, cluster0 + j, tx0[j], ty0[j], tx1[j], ty1[j], wants_clip, lean);
        value[j] = (cluster0 + j) | (lean ? 0u : rh::PAIR_GENERAL);     // what the cluster's pairs carry
        count += cnt[j];
    }
    if (wants_clip) s_wants_clip = 1u;
    const uint32_t inc = wave_inclusive_scan(count, (int)lane);
    if (lane == 63u) wave_sum[wave] = inc;
    __syncthreads();
    uint32_t before = inc - count;
    unsigned long long part_total = 0ull;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; w++) { const uint32_t ws = wave_sum[w]; part_total += ws; if (w < wave) before += ws; }
    // A triangle that needs clipping (these kernels leave it out): the part's count is pushed beyond any capacity, so the
    // total of the pass is, nobody after this kernel writes anything, and the host runs the pass with the clipping kernels.
    if (!CLIP && s_wants_clip) part_total += RH_WANTS_CLIP;
    // (the clipping kernels say whether the pass needed them: the flag is in place before the part's count is published,
    // and whoever reads counts reads it after them)
    if (CLIP && s_wants_clip && threadIdx.x == 0) { atomicOr(&a.state->wants_clip, 1u); __threadfence(); }
    if (wave == 0u) {
        // the look-back: lane i polls the i-th predecessor of the current window of 64
        unsigned long long* const status = a.cluster_status;
        if (lane == 0u) __hip_atomic_store(&status[part], ((unsigned long long)(part == 0u ? RH_STATUS_PREFIX : RH_STATUS_PARTIAL) << 62) | part_total,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long base = 0ull;
        bool gave_up = false;
        for (int64_t window = (int64_t)part - 1; window >= 0; window -= 64) {
            const int64_t idx = window - (int64_t)lane;
            unsigned long long st = (unsigned long long)RH_STATUS_PREFIX << 62;             // before the first workgroup: nothing
            if (idx >= 0) {
                uint32_t spins = 0;
                while (true) {
                    st = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((st >> 62) != 0ull) break;
                    if (++spins > RH_LOOKBACK_SPINS) { gave_up = true; st = (unsigned long long)RH_STATUS_PREFIX << 62; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const unsigned long long known = __ballot((st >> 62) == (unsigned long long)RH_STATUS_PREFIX);
            // add up the lanes up to and including the first that knows its prefix
            const uint32_t last = known ? (uint32_t)__builtin_ctzll(known) : 63u;
            unsigned long long v = lane <= last ? (st & STATUS_VALUE) : 0ull;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            base += __shfl(v, 0);
            if (known) break;
        }
        if (__ballot(gave_up) && lane == 0u) { a.state->fault = 1u; a.host_total[1] = 1ull; }
        if (lane == 0u) {
            s_base = base;
            if (part != 0u) __hip_atomic_store(&status[part], ((unsigned long long)RH_STATUS_PREFIX << 62) | (base + part_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (CLIP) __threadfence();
            if (part == n_parts - 1u) {
                if (CLIP) a.host_total[2] = __hip_atomic_load(&a.state->wants_clip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.state->total = base + part_total; a.host_total[0] = base + part_total;
            }
        }
    }
    __syncthreads();
    unsigned long long off = s_base + before;
    const uint32_t passes = a.sort_passes;
    constexpr uint32_t HUGE_PAIRS = 2048u;
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
        const bool big = cnt[j] > 32u, huge = cnt[j] > HUGE_PAIRS;
        const unsigned long long first = off;
        // a clipped cluster: one pair per tile and sub-record, the sub-record's number above the cluster in the pair's value
        const uint32_t sa = RH_RANGE_SUBS_A(tx0[j]), sb = RH_RANGE_SUBS_B(tx0[j]), per_tile = sa + sb ? sa + sb : 1u, cx0 = RH_RANGE_X0(tx0[j]);
        if (!big)
            for (uint32_t ty = ty0[j]; ty <= ty1[j]; ty++)
                for (uint32_t tx = cx0; tx <= tx1[j]; tx++)
                    for (uint32_t k = 0; k < per_tile; k++, off++) {
                        if (off >= a.capacity) continue;
                        const uint32_t tile = ty * p.tiles_x + tx;
                        const uint32_t sub = k < sa ? k : (sa + sb ? 4u + (k - sa) : 0u);
                        a.keys[off] = tile; a.vals[off] = value[j] | (sub << rh::CLIP_SUB_SHIFT);
                        for (uint32_t k = 0; k < passes; k++) atomicAdd(&digits[k][(tile >> (8u * k)) & 0xffu], 1u);
                    }
        else off += cnt[j];
        // clusters with thousands of pairs (a quad over the whole target) wait for the whole workgroup, below
        bool queued = false;
        if (huge) {
            const uint32_t slot = atomicAdd(&s_huge_n, 1u);
            if (slot < HUGE_SLOTS) { s_huge[slot] = make_uint4(tx0[j] | (ty0[j] << 16), tx1[j] | (ty1[j] << 16), value[j], 0u); s_huge_off[slot] = first; queued = true; }
        }
        // clusters with more than 32 pairs are written by the whole wave, one after the other
        for (unsigned long long todo = __ballot(big && !queued); todo; todo &= todo - 1ull) {
            const int src = __builtin_ctzll(todo);
            const uint32_t lo = __shfl(tx0[j] | (ty0[j] << 16), src), hi = __shfl(tx1[j] | (ty1[j] << 16), src), cc = __shfl(value[j], src);
            const unsigned long long o = __shfl(first, src);
            write_cluster_pairs(a, digits, lo, hi, cc, o, lane, 64u);
        }
    }
    __syncthreads();
    for (uint32_t h = 0; h < min(s_huge_n, HUGE_SLOTS); h++) {
        const uint4 c = s_huge[h];
        write_cluster_pairs(a, digits, c.x, c.y, c.z, s_huge_off[h], threadIdx.x, THREADS);
    }
    __syncthreads();
    // the workgroups spread their adds over RH_DIGIT_COPIES copies of the counts
    if (threadIdx.x < 256u)
        for (uint32_t k = 0; k < passes; k++) {
            const uint32_t c = digits[k][threadIdx.x];
            if (c) atomicAdd(&a.state->digit_count[k][part % RH_DIGIT_COPIES][threadIdx.x], c);
        }
}
extern "C" __global__ __launch_bounds__(RH_PAIR_THREADS) void rh_bin_pairs(rh_bin_pairs_args a) { bin_pairs<false>(a); }
extern "C" __global__ __launch_bounds__(RH_PAIR_THREADS) void rh_bin_pairs_clip(rh_bin_pairs_args a) { bin_pairs<true>(a); }
