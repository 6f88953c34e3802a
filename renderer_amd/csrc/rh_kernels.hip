// Shader-independent device kernels of librenderer_hip.so: prefix sums, (tile, cluster) pair emission, the stable
// LSD radix sort that turns the pair array into per-tile cluster lists in primitive order, tile range extraction,
// and the texture -> buffer copy of the capture path.
#include <hip/hip_runtime.h>
#include "rh_abi.h"
#include "rh_kernels.h"

// Scratch of rh_submit_static (the entry point that allocates nothing): binning arrays and set-up records of one pass.
// Every GPU the library runs on holds its own copy until the process ends.  In host code the name is only a stand-in:
// the device address comes from hipGetSymbolAddress (rhk::static_scratch), once per device.
static __device__ __align__(256) unsigned char scratch[RH_STATIC_SCRATCH_BYTES];

namespace {

constexpr int WG = 256;
constexpr int SCAN_ITEMS = 16;                 // items per thread in the chunked scan
constexpr int SCAN_CHUNK = WG * SCAN_ITEMS;    // 4096
constexpr uint32_t SCAN_FUSED_CHUNKS = 2048;   // up to 8 M items: the apply phase adds up the chunk sums before its own itself
constexpr int SORT_ITEMS = 16;
constexpr int SORT_CHUNK = WG * SORT_ITEMS;    // 4096 pairs per workgroup, 1024 per wave

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t n = __shfl_up(v, off);
        if (lane >= off) v += n;
    }
    return v;
}

// Exclusive scan of one value per thread across a 256-thread workgroup; returns the exclusive prefix and the total.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds4, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) lds4[wave] = inc;
    __syncthreads();
    uint32_t w0 = lds4[0], w1 = lds4[1], w2 = lds4[2], w3 = lds4[3];
    uint32_t base = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
    total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

} // namespace

// ---- chunked exclusive scan (in place): one launch for one chunk, two up to SCAN_FUSED_CHUNKS chunks, else three ----------
// phase 1: per-chunk sums
extern "C" __global__ __launch_bounds__(256) void rh_scan_reduce(const uint32_t* data, uint32_t n, uint32_t* chunk_sums) {
    __shared__ uint32_t lds4[4];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        size_t idx = base + (size_t)i * WG + threadIdx.x;
        if (idx < n) s += data[idx];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// phase 2: single workgroup, exclusive scan of the chunk sums in place; writes the grand total (64-bit)
extern "C" __global__ __launch_bounds__(256) void rh_scan_chunks(uint32_t* chunk_sums, uint32_t n_chunks, unsigned long long* total_out) {
    __shared__ uint32_t lds4[4];
    unsigned long long running = 0;
    for (uint32_t base = 0; base < n_chunks; base += WG) {
        uint32_t idx = base + threadIdx.x;
        uint32_t v = idx < n_chunks ? chunk_sums[idx] : 0u;
        uint32_t total;
        uint32_t ex = block_exclusive_scan(v, lds4, total);
        if (idx < n_chunks) chunk_sums[idx] = (uint32_t)running + ex;
        running += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = running;
}

// phase 3: per-chunk exclusive scan plus the chunk's base.  With `fused` the chunk array still holds the raw sums of
// phase 1 and every workgroup adds up the sums of the chunks before its own (a few loads per thread for the chunk
// counts this is used with), which saves the single-workgroup launch of phase 2; the last workgroup writes the grand
// total.  A one-chunk scan needs no phase 1 either: its base is zero.
extern "C" __global__ __launch_bounds__(256) void rh_scan_apply(uint32_t* data, uint32_t n, const uint32_t* chunk_bases, uint32_t fused,
                                                               unsigned long long* total_out) {
    __shared__ uint32_t lds4[4];
    __shared__ unsigned long long before[4];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = (base + i < n) ? data[base + i] : 0u; s += v[i]; }
    unsigned long long chunk_base = 0;
    if (fused) {
        unsigned long long part = 0;
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += WG) part += chunk_bases[i];
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
        if ((threadIdx.x & 63) == 0) before[threadIdx.x >> 6] = part;
        __syncthreads();
        chunk_base = before[0] + before[1] + before[2] + before[3];
    } else chunk_base = chunk_bases[blockIdx.x];
    uint32_t total;
    uint32_t ex = block_exclusive_scan(s, lds4, total) + (uint32_t)chunk_base;
    if (fused && total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = chunk_base + total;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { if (base + i < n) data[base + i] = ex; ex += v[i]; }
}

// ---- pair emission ------------------------------------------------------------------------------------------------------
// One thread per cluster (same 256-cluster partition as rh_bin_count).  Cluster c writes its (tile, c) pairs at
// block_base + exclusive prefix within the block, tiles in row-major order, so the pair array is ordered by cluster.
// Clusters that touch more than 32 tiles are written cooperatively by the whole workgroup.
extern "C" __global__ __launch_bounds__(256) void rh_bin_emit(const rh_tile_range* ranges, uint32_t n_clusters, uint32_t tiles_x,
                                                             const uint32_t* block_bases, uint32_t* keys, uint32_t* vals) {
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t big_n;
    __shared__ uint32_t big_cluster[WG], big_off[WG];
    if (threadIdx.x == 0) big_n = 0;
    const uint32_t c = blockIdx.x * WG + threadIdx.x;
    uint32_t tx0 = 1, ty0 = 0, tx1 = 0, ty1 = 0, count = 0, sa = 0, sb = 0, per_tile = 1, general = 0;
    if (c < n_clusters) {
        rh_tile_range r = ranges[c];
        sa = RH_RANGE_SUBS_A(r.lo); sb = RH_RANGE_SUBS_B(r.lo); per_tile = sa + sb ? sa + sb : 1u;     // a clipped cluster: a pair per tile and sub-record
        tx0 = RH_RANGE_X0(r.lo); ty0 = r.lo >> 16; tx1 = RH_RANGE_X1(r.hi); ty1 = r.hi >> 16;
        general = (r.hi & RH_RANGE_GENERAL) ? (1u << 28) : 0u;           // rh::PAIR_GENERAL: travels in the pairs' values
        if (tx0 <= tx1) count = (tx1 - tx0 + 1) * (ty1 - ty0 + 1) * per_tile;
    }
    uint32_t total;
    uint32_t off = block_exclusive_scan(count, lds4, total) + block_bases[blockIdx.x];
    if (count > 32) {
        uint32_t slot = atomicAdd(&big_n, 1u);
        big_cluster[slot] = threadIdx.x; big_off[slot] = off;
    } else if (count > 0) {
        for (uint32_t ty = ty0; ty <= ty1; ty++)
            for (uint32_t tx = tx0; tx <= tx1; tx++)
                for (uint32_t k = 0; k < per_tile; k++) {
                    const uint32_t sub = k < sa ? k : (sa + sb ? 4u + (k - sa) : 0u);
                    keys[off] = ty * tiles_x + tx; vals[off] = c | general | (sub << 29); off++;
                }
    }
    __syncthreads();
    const uint32_t nb = big_n;
    for (uint32_t b = 0; b < nb; b++) {
        const uint32_t cc = blockIdx.x * WG + big_cluster[b];
        rh_tile_range r = ranges[cc];
        const uint32_t ba = RH_RANGE_SUBS_A(r.lo), bb = RH_RANGE_SUBS_B(r.lo), per = ba + bb ? ba + bb : 1u;
        const uint32_t x0 = RH_RANGE_X0(r.lo), y0 = r.lo >> 16, x1 = RH_RANGE_X1(r.hi), y1 = r.hi >> 16;
        const uint32_t w = x1 - x0 + 1, cnt = w * (y1 - y0 + 1) * per, o = big_off[b], gen = (r.hi & RH_RANGE_GENERAL) ? (1u << 28) : 0u;
        for (uint32_t i = threadIdx.x; i < cnt; i += WG) {
            const uint32_t t = i / per, k = i - t * per, sub = k < ba ? k : (ba + bb ? 4u + (k - ba) : 0u);
            keys[o + i] = (y0 + t / w) * tiles_x + (x0 + t % w); vals[o + i] = cc | gen | (sub << 29);
        }
    }
}

// ---- stable LSD radix sort, 8 bits per pass ------------------------------------------------------------------------------
// histogram: hist[digit * n_blocks + block]
extern "C" __global__ __launch_bounds__(256) void rh_sort_hist(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t* hist, uint32_t n_blocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_CHUNK;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; i++) {
        size_t idx = base + (size_t)i * WG + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 0xffu], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

// scatter: `offsets` is the exclusive scan of hist.  Each wave owns 1024 consecutive pairs and walks them in order,
// 64 at a time; a lane's rank among the lanes with the same digit comes from eight ballots, so equal keys keep their
// input order (stability is what preserves primitive order inside a tile).  The workgroup first sorts its 4096 pairs by
// digit inside LDS and then writes them out in that order: the pairs of one digit leave as one contiguous run instead
// of 4-byte writes all over the output.
extern "C" __global__ __launch_bounds__(256) void rh_sort_scatter(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                                                                 uint32_t n, uint32_t shift, const uint32_t* offsets, uint32_t n_blocks) {
    __shared__ uint32_t cnt[4][256];            // per wave and digit: count, then the next free local slot
    __shared__ uint32_t digit_start[256];       // first local slot of each digit
    __shared__ uint32_t global_base[256];       // where this workgroup's run of each digit starts in the output
    __shared__ uint32_t skey[SORT_CHUNK], sval[SORT_CHUNK];
    __shared__ uint32_t lds4[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += WG) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const size_t block_base = (size_t)blockIdx.x * SORT_CHUNK;
    const size_t wave_base = block_base + (size_t)wave * (SORT_CHUNK / 4);
    // per-wave digit counts
    for (int step = 0; step < SORT_CHUNK / 4 / 64; step++) {
        size_t idx = wave_base + (size_t)step * 64 + lane;
        if (idx < n) atomicAdd(&cnt[wave][(keys_in[idx] >> shift) & 0xffu], 1u);
    }
    __syncthreads();
    // local layout: digits ascending, inside a digit the waves in order (= input order)
    {
        const int d = threadIdx.x;
        const uint32_t c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
        uint32_t total;
        const uint32_t start = block_exclusive_scan(c0 + c1 + c2 + c3, lds4, total);
        digit_start[d] = start;
        global_base[d] = offsets[(size_t)d * n_blocks + blockIdx.x];
        cnt[0][d] = start; cnt[1][d] = start + c0; cnt[2][d] = start + c0 + c1; cnt[3][d] = start + c0 + c1 + c2;
    }
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int step = 0; step < SORT_CHUNK / 4 / 64; step++) {
        size_t idx = wave_base + (size_t)step * 64 + lane;
        const bool valid = idx < n;
        uint32_t key = 0, val = 0;
        if (valid) { key = keys_in[idx]; val = vals_in[idx]; }
        const uint32_t d = (key >> shift) & 0xffu;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        if (valid) {
            const uint32_t slot = cnt[wave][d] + (uint32_t)__popcll(peers & lt_mask);
            skey[slot] = key; sval[slot] = val;
        }
        // the highest lane of every digit group advances that digit's slot (reads above finish before the update:
        // same wave, program order, LDS operations complete in order)
        if (valid && (peers >> lane) == 1ull) cnt[wave][d] += (uint32_t)__popcll(peers);
    }
    __syncthreads();
    const uint32_t chunk_n = (uint32_t)(n - block_base < (size_t)SORT_CHUNK ? n - block_base : (size_t)SORT_CHUNK);
    for (uint32_t i = threadIdx.x; i < chunk_n; i += WG) {
        const uint32_t key = skey[i], d = (key >> shift) & 0xffu;
        const uint32_t dst = global_base[d] + (i - digit_start[d]);
        keys_out[dst] = key; vals_out[dst] = sval[i];
    }
}

// One sorting pass in one launch: the digit counts of the whole array are already known (rh_bin_pairs counted them), so
// a workgroup only needs, per digit value, how many pairs with that value the workgroups before it hold -- a chained
// look-back per digit value, thread d of every workgroup following digit value d (see rh_bin_state).  Workgroup `part`
// (its ticket) sorts pairs [PASS_CHUNK * part, PASS_CHUNK * (part + 1)) the way rh_sort_scatter does: every wave walks
// its 1024 consecutive pairs in order.
// status: [parts + ceil(parts / PASS_GROUP)][256] words per pass (rhk::sort_status_words), cleared; counts must stay below 2^30.
#ifndef RH_PASS_SORT_WAVES
#define RH_PASS_SORT_WAVES 4
#endif
constexpr int PASS_SORT_WAVES = RH_PASS_SORT_WAVES;                      // waves that hold and sort the pairs, 1024 each
constexpr int PASS_WG = PASS_SORT_WAVES * 64 + 256;                      // plus the four waves that follow the digit values
constexpr int PASS_STEPS = 16, PASS_CHUNK = PASS_SORT_WAVES * 64 * PASS_STEPS;
constexpr uint32_t PASS_GROUP = 32;                                      // parts per group of the two-level look-back

// Exclusive scan of v[0..256) by one wave (four consecutive values per lane).
__device__ __forceinline__ void wave_scan_256(const uint32_t* v, uint32_t* out, int lane) {
    const uint32_t a = v[4 * lane], b = v[4 * lane + 1], c = v[4 * lane + 2], d = v[4 * lane + 3];
    const uint32_t ex = wave_inclusive_scan(a + b + c + d, lane) - (a + b + c + d);
    out[4 * lane] = ex; out[4 * lane + 1] = ex + a; out[4 * lane + 2] = ex + a + b; out[4 * lane + 3] = ex + a + b + c;
}

#ifdef RH_SORT_STAMPS      // diagnostic build (tools/probe_sort.hip): wall-clock stamps of the phases of the last pass, per part
__device__ unsigned long long rh_sort_stamps[4096 * 8];
#define RH_STAMP(k, who) do { if (threadIdx.x == (who) && part < 4096u) rh_sort_stamps[part * 8u + (k)] = wall_clock64(); } while (0)
#else
#define RH_STAMP(k, who) do {} while (0)
#endif

// `ticketed` = 0: the host knows that all the workgroups of the launch fit the device at once, and a workgroup takes its
// index as its place in the chain (nobody can wait for a workgroup that never starts); otherwise places come from the
// ticket counter.
extern "C" __global__ __launch_bounds__(PASS_WG) void rh_sort_pass(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                                                                  uint32_t capacity, uint32_t shift, uint32_t pass, rh_bin_state* state, uint32_t* status,
                                                                  unsigned long long* host_total, uint32_t ticketed) {
    const unsigned long long total = state->total;
    if (total > capacity) return;
    const uint32_t n = (uint32_t)total;
    __shared__ uint32_t s_part;
    __shared__ uint32_t cnt[PASS_SORT_WAVES][256];   // per sorting wave and digit value: pairs
    __shared__ uint32_t slot[PASS_SORT_WAVES][256];  // per sorting wave and digit value: the next free local slot
    __shared__ uint32_t digit_start[256];            // first local slot of each digit value
    __shared__ uint32_t all_count[256];              // the whole array's pairs per digit value
    __shared__ uint32_t digit_base[256];             // their exclusive scan: pairs with a smaller digit value
    __shared__ uint32_t before_mine[256];            // pairs with the same digit value in earlier parts
    __shared__ uint32_t skey[PASS_CHUNK], sval[PASS_CHUNK];
    if (ticketed && threadIdx.x == 0) s_part = atomicAdd(&state->ticket[1 + pass], 1u);
    for (int i = threadIdx.x; i < PASS_SORT_WAVES * 256; i += PASS_WG) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t part = ticketed ? s_part : blockIdx.x, n_parts = gridDim.x;
    if ((size_t)part * PASS_CHUNK >= n) return;
    RH_STAMP(0, 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool sorts = wave < PASS_SORT_WAVES;       // else: thread d of the last four waves follows digit value d
    const uint32_t d = (uint32_t)(threadIdx.x - PASS_SORT_WAVES * 64) & 255u;
    const size_t block_base = (size_t)part * PASS_CHUNK;
    const size_t wave_base = block_base + (size_t)wave * (PASS_STEPS * 64);
    uint32_t key[PASS_STEPS], val[PASS_STEPS];
    if (sorts) {
        // (all the loads first, from addresses inside the array also for the lanes past its end: one round trip)
#pragma unroll
        for (int step = 0; step < PASS_STEPS; step++) {
            const size_t idx = wave_base + (size_t)step * 64 + lane;
            key[step] = keys_in[idx < n ? idx : 0];
        }
#pragma unroll
        for (int step = 0; step < PASS_STEPS; step++)
            if (wave_base + (size_t)step * 64 + lane < n) atomicAdd(&cnt[wave][(key[step] >> shift) & 0xffu], 1u);
    } else {
        uint32_t all = 0u;
#pragma unroll
        for (int k = 0; k < RH_DIGIT_COPIES; k++) all += state->digit_count[pass][k][d];
        all_count[d] = all;
    }
    __syncthreads();
    RH_STAMP(1, 0);
    if (sorts) {
        // The sorting waves order the part by digit value inside LDS (values ascending, inside a value the waves in order
        // = input order) while the other waves find the part's place in the output.
#pragma unroll
        for (int step = 0; step < PASS_STEPS; step++) {
            const size_t idx = wave_base + (size_t)step * 64 + lane;
            val[step] = vals_in[idx < n ? idx : 0];
        }
        {   // this wave's first slot of every digit value (every wave works the layout out for itself: no barrier)
            uint32_t tot[4], mine4[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                tot[i] = 0u; mine4[i] = 0u;
#pragma unroll
                for (int w = 0; w < PASS_SORT_WAVES; w++) { const uint32_t c = cnt[w][4 * lane + i]; tot[i] += c; if (w < wave) mine4[i] += c; }
            }
            const uint32_t sum = tot[0] + tot[1] + tot[2] + tot[3];
            uint32_t at = wave_inclusive_scan(sum, lane) - sum;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (wave == 0) digit_start[4 * lane + i] = at;
                slot[wave][4 * lane + i] = at + mine4[i];
                at += tot[i];
            }
        }
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
        for (int step = 0; step < PASS_STEPS; step++) {
            const size_t idx = wave_base + (size_t)step * 64 + lane;
            const bool valid = idx < n;
            const uint32_t dv = (key[step] >> shift) & 0xffu;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                unsigned long long bal = __ballot((dv >> b) & 1u);
                peers &= ((dv >> b) & 1u) ? bal : ~bal;
            }
            if (valid) {
                const uint32_t at = slot[wave][dv] + (uint32_t)__popcll(peers & lt_mask);
                skey[at] = key[step]; sval[at] = val[step];
            }
            // the highest lane of every group advances the slot (the reads above finish first: same wave, program order)
            if (valid && (peers >> lane) == 1ull) slot[wave][dv] += (uint32_t)__popcll(peers);
        }
        RH_STAMP(2, 0);
    } else {
        uint32_t c = 0u;
#pragma unroll
        for (int w = 0; w < PASS_SORT_WAVES; w++) c += cnt[w][d];
        uint32_t* const mine = status + (size_t)part * 256u + d;
        __hip_atomic_store(mine, (RH_STATUS_PARTIAL << 30) | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == PASS_SORT_WAVES) wave_scan_256(all_count, digit_base, lane);
        // The look-back, in two levels: the parts before this one in its group of PASS_GROUP (all polled at once: counts
        // only), and the groups before that, whose last parts keep a chain of their own (group count, then group prefix).
        const uint32_t group = part / PASS_GROUP, member = part % PASS_GROUP;
        uint32_t* const group_status = status + (size_t)n_parts * 256u;
        bool gave_up = false;
        auto poll = [&](const uint32_t* word, uint32_t need_prefix) -> uint32_t {
            uint32_t st = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), spins = 0;
            while ((st >> 30) == 0u || (need_prefix && (st >> 30) != RH_STATUS_PREFIX)) {
                if (++spins > RH_LOOKBACK_SPINS) { gave_up = true; return RH_STATUS_PREFIX << 30; }
                __builtin_amdgcn_s_sleep(1);
                st = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return st;
        };
        uint32_t before = 0u;                                // pairs with this digit value in earlier parts
        for (uint32_t j0 = 0; j0 < member; j0 += 8u) {
            uint32_t st[8];
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++)
                st[j] = j0 + j < member ? __hip_atomic_load(status + (size_t)(part - member + j0 + j) * 256u + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (RH_STATUS_PARTIAL << 30);
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) {
                if ((st[j] >> 30) == 0u) st[j] = poll(status + (size_t)(part - member + j0 + j) * 256u + d, 0u);
                before += st[j] & 0x3fffffffu;
            }
        }
        if (member == PASS_GROUP - 1u) {
            // last part of a group: publishes the group's count, finds the prefix of the groups before, publishes the group's prefix
            uint32_t* const ours = group_status + (size_t)group * 256u + d;
            const uint32_t group_count = before + c;
            __hip_atomic_store(ours, ((group == 0u ? RH_STATUS_PREFIX : RH_STATUS_PARTIAL) << 30) | group_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t groups_before = 0u;
            bool done = group == 0u;
            for (int64_t b = (int64_t)group - 1; !done && b >= 0; b -= 4) {
                uint32_t st[4];
#pragma unroll
                for (int j = 0; j < 4; j++) st[j] = b - j >= 0 ? __hip_atomic_load(group_status + (size_t)(b - j) * 256u + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (RH_STATUS_PREFIX << 30);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (done) break;
                    if ((st[j] >> 30) == 0u) st[j] = poll(group_status + (size_t)(b - j) * 256u + d, 0u);
                    groups_before += st[j] & 0x3fffffffu;
                    done = (st[j] >> 30) == RH_STATUS_PREFIX;
                }
            }
            if (group != 0u) __hip_atomic_store(ours, (RH_STATUS_PREFIX << 30) | (groups_before + group_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            before += groups_before;
        } else if (group != 0u) {
            before += poll(group_status + (size_t)(group - 1u) * 256u + d, 1u) & 0x3fffffffu;
        }
        if (gave_up) { state->fault = 1u; host_total[1] = 1ull; }
        before_mine[d] = before;
        RH_STAMP(3, PASS_SORT_WAVES * 64);
    }
    __syncthreads();
    RH_STAMP(4, 0);
    const uint32_t chunk_n = (uint32_t)(n - block_base < (size_t)PASS_CHUNK ? n - block_base : (size_t)PASS_CHUNK);
    for (uint32_t i = threadIdx.x; i < chunk_n; i += PASS_WG) {
        const uint32_t k = skey[i], dv = (k >> shift) & 0xffu;
        const uint32_t dst = digit_base[dv] + before_mine[dv] + (i - digit_start[dv]);
        keys_out[dst] = k; vals_out[dst] = sval[i];
    }
    RH_STAMP(5, 0);
}

// ---- capture: texture region -> buffer with a 256-byte-aligned row pitch --------------------------------------------------
// One thread per 4 bytes; rows are contiguous in both source and destination so accesses coalesce.  Padding bytes
// at the end of each destination row are written as zero.
extern "C" __global__ __launch_bounds__(256) void rh_copy_texture_to_buffer(const uint8_t* tex, uint32_t tex_pitch_bytes, uint32_t origin_x, uint32_t origin_y,
                                                                           uint32_t bytes_per_texel, uint32_t width, uint32_t height,
                                                                           uint8_t* dst, uint32_t dst_pitch_bytes) {
    const uint32_t row = blockIdx.y;
    if (row >= height) return;
    const uint32_t row_bytes = width * bytes_per_texel;
    const uint8_t* src_row = tex + (size_t)(origin_y + row) * tex_pitch_bytes + (size_t)origin_x * bytes_per_texel;
    uint8_t* dst_row = dst + (size_t)row * dst_pitch_bytes;
    for (uint32_t b = (blockIdx.x * WG + threadIdx.x) * 4u; b < dst_pitch_bytes; b += gridDim.x * WG * 4u) {
        uint32_t v = 0;
        if (b + 4 <= row_bytes && ((bytes_per_texel & 3u) == 0)) v = *reinterpret_cast<const uint32_t*>(src_row + b);
        else { for (uint32_t k = 0; k < 4; k++) if (b + k < row_bytes) v |= (uint32_t)src_row[b + k] << (8 * k); }
        *reinterpret_cast<uint32_t*>(dst_row + b) = v;
    }
}

// ---- footprint copies of a texture, regrouped by phase (rh_slot_dev::poly) ----------------------------------------------------
// One thread per texel: item (x, y) of a layer = the four taps of a bilinear sample whose base texel is (x, y); copy k
// (S = 2^k) gets the item at column (x mod S) * (width / S) + x / S of its row.  width is a multiple of 32.
extern "C" __global__ __launch_bounds__(256) void rh_polyphase_build(const uint32_t* src, uint4* dst, uint32_t width, uint32_t height, uint32_t layers) {
    const size_t texels = (size_t)width * height * layers;
    const size_t i = (size_t)blockIdx.x * WG + threadIdx.x;
    if (i >= texels) return;
    const uint32_t x = (uint32_t)(i % width);
    const size_t row = i / width;
    const uint32_t y = (uint32_t)(row % height);
    const size_t dx = x + 1 < width ? 1 : 0, dy = y + 1 < height ? width : 0;
    const uint4 v = make_uint4(src[i], src[i + dx], src[i + dy], src[i + dy + dx]);
    const size_t row_base = row * width;
#pragma unroll
    for (uint32_t k = 0; k < RH_POLY_COPIES; k++) {
        const uint32_t S = 1u << k;
        dst[k * texels + row_base + (x & (S - 1u)) * (width >> k) + (x >> k)] = v;
    }
}

// The same with 32-byte items: per channel { t(x,y) | (t(x+1,y) - t(x,y)) << 16 } and the same for the row below, as f16
// pairs (exact: 8-bit values and their differences); see rh_slot_dev::poly.
extern "C" __global__ __launch_bounds__(256) void rh_polyphase_build_h(const uint32_t* src, uint4* dst, uint32_t width, uint32_t height, uint32_t layers) {
    const size_t texels = (size_t)width * height * layers;
    const size_t i = (size_t)blockIdx.x * WG + threadIdx.x;
    if (i >= texels) return;
    const uint32_t x = (uint32_t)(i % width);
    const size_t row = i / width;
    const uint32_t y = (uint32_t)(row % height);
    const size_t dx = x + 1 < width ? 1 : 0, dy = y + 1 < height ? width : 0;
    const uint32_t t00 = src[i], t10 = src[i + dx], t01 = src[i + dy], t11 = src[i + dy + dx];
    auto pair = [](uint32_t left, uint32_t right, int c) -> uint32_t {
        const int l = (int)((left >> (8 * c)) & 0xffu), r = (int)((right >> (8 * c)) & 0xffu);
        const _Float16 lo = (_Float16)(float)l, hi = (_Float16)(float)(r - l);
        return (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
    };
    const uint4 v0 = make_uint4(pair(t00, t10, 0), pair(t01, t11, 0), pair(t00, t10, 1), pair(t01, t11, 1));
    const uint4 v1 = make_uint4(pair(t00, t10, 2), pair(t01, t11, 2), pair(t00, t10, 3), pair(t01, t11, 3));
    const size_t row_base = row * width;
#pragma unroll
    for (uint32_t k = 0; k < RH_POLY_COPIES; k++) {
        const uint32_t S = 1u << k;
        const size_t at = 2u * (k * texels + row_base + (x & (S - 1u)) * (width >> k) + (x >> k));
        dst[at] = v0; dst[at + 1] = v1;
    }
}

// ---- clearing the state area of a pass ------------------------------------------------------------------------------------
// (one launch of a few microseconds; the runtime's fill is two launches for a size that is not a multiple of its unit)
extern "C" __global__ __launch_bounds__(256) void rh_clear(uint4* p, uint32_t n16) {
    const uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ---- host-callable launchers ----------------------------------------------------------------------------------------------
namespace rhk {

static inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

hipError_t static_scratch(void** ptr, size_t* bytes) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    *ptr = p; *bytes = (size_t)RH_STATIC_SCRATCH_BYTES;
    return hipSuccess;
}

uint32_t scan_chunks_for(uint32_t n) { return div_up(n, SCAN_CHUNK); }
uint32_t sort_blocks_for(uint32_t n) { return div_up(n, SORT_CHUNK); }
uint32_t sort_parts_for(uint32_t n) { return div_up(n, PASS_CHUNK); }
size_t sort_status_words(uint32_t capacity) { const uint32_t parts = sort_parts_for(capacity); return (size_t)(parts + div_up(parts, PASS_GROUP)) * 256u; }

// In-place exclusive scan of data[0..n); chunk_tmp needs scan_chunks_for(n) words.
void exclusive_scan(uint32_t* data, uint32_t n, uint32_t* chunk_tmp, unsigned long long* total_out, hipStream_t s, bool general) {
    if (n == 0) { if (total_out) (void)hipMemsetAsync(total_out, 0, sizeof(unsigned long long), s); return; }
    const uint32_t chunks = scan_chunks_for(n);
    if (chunks <= SCAN_FUSED_CHUNKS && !general) {
        if (chunks > 1) hipLaunchKernelGGL(rh_scan_reduce, dim3(chunks), dim3(WG), 0, s, data, n, chunk_tmp);
        hipLaunchKernelGGL(rh_scan_apply, dim3(chunks), dim3(WG), 0, s, data, n, chunk_tmp, 1u, total_out);
        return;
    }
    hipLaunchKernelGGL(rh_scan_reduce, dim3(chunks), dim3(WG), 0, s, data, n, chunk_tmp);
    hipLaunchKernelGGL(rh_scan_chunks, dim3(1), dim3(WG), 0, s, chunk_tmp, chunks, total_out);
    hipLaunchKernelGGL(rh_scan_apply, dim3(chunks), dim3(WG), 0, s, data, n, chunk_tmp, 0u, (unsigned long long*)nullptr);
}

void bin_emit(const rh_tile_range* ranges, uint32_t n_clusters, uint32_t tiles_x, const uint32_t* block_bases,
              uint32_t* keys, uint32_t* vals, hipStream_t s) {
    if (n_clusters == 0) return;
    hipLaunchKernelGGL(rh_bin_emit, dim3(div_up(n_clusters, WG)), dim3(WG), 0, s, ranges, n_clusters, tiles_x, block_bases, keys, vals);
}

// Sorts n pairs by the low `key_bits` bits of the key.  Buffers ping-pong; returns 0 when the result is in
// (keys_a, vals_a) and 1 when it is in (keys_b, vals_b).  hist needs 256 * sort_blocks_for(n) words, chunk_tmp
// scan_chunks_for(256 * sort_blocks_for(n)) words.
int radix_sort_pairs(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, uint32_t key_bits,
                     uint32_t* hist, uint32_t* chunk_tmp, hipStream_t s, bool general) {
    if (n == 0) return 0;
    const uint32_t blocks = sort_blocks_for(n);
    int cur = 0;
    for (uint32_t shift = 0; shift < key_bits; shift += 8) {
        uint32_t* ki = cur ? keys_b : keys_a; uint32_t* vi = cur ? vals_b : vals_a;
        uint32_t* ko = cur ? keys_a : keys_b; uint32_t* vo = cur ? vals_a : vals_b;
        hipLaunchKernelGGL(rh_sort_hist, dim3(blocks), dim3(WG), 0, s, ki, n, shift, hist, blocks);
        exclusive_scan(hist, 256u * blocks, chunk_tmp, nullptr, s, general);
        hipLaunchKernelGGL(rh_sort_scatter, dim3(blocks), dim3(WG), 0, s, ki, vi, ko, vo, n, shift, hist, blocks);
        cur ^= 1;
    }
    return cur;
}

// The sorting passes of the chain that is queued without scans (see rh_bin_state); same result and return value as
// radix_sort_pairs.  status: RH sort passes x sort_blocks_for(capacity) x 256 words, cleared.
uint32_t sort_pass_resident_parts(uint32_t compute_units) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rh_sort_pass, PASS_WG, 0) != hipSuccess || per_cu < 1) return 0;
    return (uint32_t)per_cu * compute_units;
}

int sort_passes_chained(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t capacity, uint32_t key_bits,
                        rh_bin_state* state, uint32_t* status, unsigned long long* host_total, uint32_t resident_parts, hipStream_t s) {
    const uint32_t blocks = sort_parts_for(capacity);
    const uint32_t ticketed = blocks > resident_parts ? 1u : 0u;
    int cur = 0;
    uint32_t pass = 0;
    for (uint32_t shift = 0; shift < key_bits; shift += 8, pass++) {
        uint32_t* ki = cur ? keys_b : keys_a; uint32_t* vi = cur ? vals_b : vals_a;
        uint32_t* ko = cur ? keys_a : keys_b; uint32_t* vo = cur ? vals_a : vals_b;
        hipLaunchKernelGGL(rh_sort_pass, dim3(blocks), dim3(PASS_WG), 0, s, ki, vi, ko, vo, capacity, shift, pass, state,
                           status + (size_t)pass * sort_status_words(capacity), host_total, ticketed);
        cur ^= 1;
    }
    return cur;
}

void clear(void* p, size_t bytes, hipStream_t s) {          // bytes: a multiple of 16
    const uint32_t n16 = (uint32_t)(bytes / 16);
    if (n16) hipLaunchKernelGGL(rh_clear, dim3(div_up(n16, WG)), dim3(WG), 0, s, (uint4*)p, n16);
}

void polyphase_build(const void* src, void* dst, uint32_t width, uint32_t height, uint32_t layers, uint32_t item_bytes, hipStream_t s) {
    const size_t texels = (size_t)width * height * layers;
    if (texels == 0) return;
    if (item_bytes == 32u) hipLaunchKernelGGL(rh_polyphase_build_h, dim3(div_up(texels, WG)), dim3(WG), 0, s, (const uint32_t*)src, (uint4*)dst, width, height, layers);
    else hipLaunchKernelGGL(rh_polyphase_build, dim3(div_up(texels, WG)), dim3(WG), 0, s, (const uint32_t*)src, (uint4*)dst, width, height, layers);
}

void copy_texture_to_buffer(const uint8_t* tex, uint32_t tex_pitch_bytes, uint32_t ox, uint32_t oy, uint32_t bpt, uint32_t w, uint32_t h,
                            uint8_t* dst, uint32_t dst_pitch, hipStream_t s) {
    if (w == 0 || h == 0) return;
    uint32_t gx = div_up(dst_pitch / 4, WG); if (gx > 16) gx = 16;
    hipLaunchKernelGGL(rh_copy_texture_to_buffer, dim3(gx, h), dim3(WG), 0, s, tex, tex_pitch_bytes, ox, oy, bpt, w, h, dst, dst_pitch);
}

} // namespace rhk
