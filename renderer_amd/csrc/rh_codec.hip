// GPU side of the capture codec: LZ4 frame (LZ4F) streams compressed on the device, one stream per workgroup, so that
// only finished compressed bytes are read back over PCIe.
//
// The reference compresses captured frames on CPU worker threads, one LZ4F stream per thread/file
// (reference src/compressor.rs:88-144, via the lzzzz crate = liblz4): preferences level 0, auto-flush, block-linked,
// 64 KiB blocks, no checksums; every Write::write_all is one LZ4F_compressUpdate.  A stream's bytes depend only on its
// own write sequence, so streams are independent: each stream's state (hash table, 64 KiB dictionary, offsets) lives
// in device memory and one wavefront replays the LZ4 "fast" algorithm for its stream.  LDS holds the hash table
// (16 KiB), fifteen bits of the sequence behind every table entry (8 KiB) and a 4 KiB window of the block around the
// current position; the 64 KiB of history are read where they are, in the captured frame (through L1/L2), so a CU
// holds four or five streams instead of one.  A wavefront alone on its SIMD issues one instruction every four or five
// cycles whatever the instruction is, so what a sequence costs is the number of instructions on its path and the
// round trips it waits for: the usual sequence (a short run of literals, a match found in the first round of the walk
// whose candidate lies in the window) takes one straight piece of code with LDS reads only; everything else takes the
// general code.
//
// The match search of LZ4 is a sequential walk (position, hash, table look-up, table insert, compare, advance by a
// step that grows with the number of misses).  The walk without a match is known in advance, so the 64 lanes evaluate
// the next 64 positions of it at once: every lane hashes its position and takes as candidate the closest earlier lane
// of the batch with the same hash (that lane's insert would have been seen by the sequential walk) or else the table
// entry; the first lane whose candidate passes the tests is where the sequential walk would have stopped, and exactly
// the inserts of the lanes up to it are applied.  Match-length counting, the backward catch-up, literal copies, block
// loads and dictionary saves are lane-parallel as well; the token bookkeeping between matches is wave-uniform.
//
// The output is byte-identical to liblz4 1.9.x for the same write sequence (LZ4_compress_fast_continue with
// acceleration 1, byU32 table, 64-bit hash5; LZ4_saveDict after every update; LZ4_renormDictT).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "rh_codec.h"

namespace {

// Pointers that are read from the job record are generic as far as the compiler can tell, and a store through a
// generic pointer is a FLAT instruction: it counts as an LDS operation too, so the wait before the next ring read would
// also wait for every byte written to the output.  These types say "global memory".
typedef __attribute__((address_space(1))) uint8_t g_u8;
typedef __attribute__((address_space(1))) uint32_t g_u32;
typedef __attribute__((address_space(1))) const uint8_t g_cu8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4 g_cu32x4;

#ifdef RH_LZ4_STAMPS        // diagnostic build (tools/probe_lz4.hip): cycles of a stream's wave per phase of a sequence, summed over all streams
__device__ unsigned long long rh_lz4_stamp_acc[24];
struct Stamps {
    unsigned long long last, acc[12]; unsigned cnt[12];
    __device__ __forceinline__ void start() { last = __builtin_readcyclecounter(); for (int k = 0; k < 12; k++) { acc[k] = 0; cnt[k] = 0; } }
    // `dep`: a value the phase ends with (the stamp is taken when it is there)
    __device__ __forceinline__ void mark(int k, uint32_t dep) {
        unsigned long long t;
        asm volatile("s_nop 0 ; %1\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
        acc[k] += t - last; last = t; cnt[k]++;
    }
    __device__ __forceinline__ void flush(int lane) { if (lane == 0) for (int k = 0; k < 12; k++) { atomicAdd(&rh_lz4_stamp_acc[k], acc[k]); atomicAdd(&rh_lz4_stamp_acc[12 + k], (unsigned long long)cnt[k]); } }
};
#define RH_STAMP(k, dep) stamps.mark(k, (uint32_t)(dep))
#define RH_STAMP_ARG , Stamps& stamps
#define RH_STAMP_PASS , stamps
#else
#define RH_STAMP(k, dep)
#define RH_STAMP_ARG
#define RH_STAMP_PASS
#endif

constexpr uint32_t KB64 = 65536u;
constexpr int MINMATCH = 4, LASTLITERALS = 5, MFLIMIT = 12, LZ4_MINLENGTH = MFLIMIT + 1;
constexpr int HASHLOG = 12;
constexpr uint32_t DISTANCE_MAX = 65535u, ML_MASK = 15u, RUN_MASK = 15u;
constexpr uint32_t BLOCK_UNCOMPRESSED_FLAG = 0x80000000u;
constexpr uint32_t CUR = KB64;                // virtual offset of the current block

// Code addresses the bytes of a block and of its history with *virtual* offsets: the current block starts at CUR, the
// history byte at distance d before the block is at CUR - d, so a match index maps to one flat offset whether it lies in
// the dictionary, in the previous blocks of the same write, or in the current block.  Behind the offsets are two
// addresses: `cur` for the block, and `hist` for what lies before it -- the same address inside a write (the previous
// block precedes the current one in the source), or the end of the stream's saved dictionary for the first block of a
// write.  Lanes read unconditionally and mask afterwards, so offsets are clamped to the 64 KiB before the block and
// the few bytes behind it that every source buffer is padded with.
constexpr uint32_t WIN = 4096u, WIN_DWORDS = WIN / 4u, WIN_CHUNK = 1024u;
struct Lds {
    // A table entry is the position's index in its low IDX_BITS bits and, above them, fifteen bits derived from the four
    // bytes at the position (0: the entry was never written).  A candidate whose bits differ from the position's own
    // cannot match, so its bytes need not be read at all: the straight piece of the search reads the bytes of one
    // candidate -- the first that can match -- instead of 64.  Indices fit seventeen bits because the table is rebased
    // before every block (the stream's offset never exceeds 64 KiB at a block's start, see the kernel), which is what
    // LZ4_renormDictT does once in 2 GiB: positions further back than 64 KiB cannot be matched anyway.
    uint32_t hash[(1 << HASHLOG) + 64];       // + one private slot per lane: lanes past the end of the walk go there,
                                              //   so every lane runs the table protocol and no step needs a lane mask
    // The bytes around the current position of a block whose history precedes it in the same source: the byte at
    // virtual offset v at index v mod 4 KiB.  What the search reads of the block itself (the sequences at the walk's
    // positions, literal runs, the bytes around a nearby match) comes from here at LDS latency; only candidates further
    // back are read from memory.  The first two dwords are repeated behind the end, so that an unaligned read takes the
    // dwords i, i+1, i+2 without wrapping.
    uint32_t win[WIN_DWORDS + 2];
};
static_assert(CUR % WIN == 0, "a virtual offset's place in the window is its low bits");

struct Window {
    const uint32_t* w;
    __device__ __forceinline__ uint32_t rd32(uint32_t v) const {
        const uint32_t i = (v >> 2) & (WIN_DWORDS - 1u);
        return __builtin_amdgcn_alignbyte(w[i + 1u], w[i], v & 3u);
    }
    __device__ __forceinline__ uint64_t rd64(uint32_t v) const {
        const uint32_t i = (v >> 2) & (WIN_DWORDS - 1u), s = v & 3u;
        const uint32_t x = w[i], y = w[i + 1u], z = w[i + 2u];
        return (uint64_t)__builtin_amdgcn_alignbyte(y, x, s) | ((uint64_t)__builtin_amdgcn_alignbyte(z, y, s) << 32);
    }
    __device__ __forceinline__ uint32_t byte(uint32_t v) const { return reinterpret_cast<const uint8_t*>(w)[v & (WIN - 1u)]; }
};

constexpr uint32_t IDX_BITS = 17u, IDX_MASK = (1u << IDX_BITS) - 1u;
__device__ __forceinline__ uint32_t tag_of(uint32_t seq) { return 0x4000u | ((seq * 2654435761u) >> 18); }      // 15 bits, never 0
__device__ __forceinline__ uint32_t entry_of(uint32_t index, uint32_t seq) { return index | (tag_of(seq) << IDX_BITS); }

constexpr uint32_t READ_SLACK = 16;           // bytes behind the end of a source that may be read (and are ignored)

struct Ring {
    g_cu8* cur;
    g_cu8* hist;
    uint32_t hi;                              // last virtual offset that may be read (the block's end + slack - 8)
    bool split;                               // hist != cur: a read that starts in the history may end in the block
    __device__ __forceinline__ g_cu8* at(uint32_t v) const {
        v = v < CUR - KB64 ? CUR - KB64 : (v > hi ? hi : v);
        return (v >= CUR ? cur : hist) + (int32_t)(v - CUR);
    }
    __device__ __forceinline__ uint32_t rd32(uint32_t v) const {
        if (split) return (uint32_t)*at(v) | ((uint32_t)*at(v + 1u) << 8) | ((uint32_t)*at(v + 2u) << 16) | ((uint32_t)*at(v + 3u) << 24);
        // (global loads take any address: one instruction whatever the alignment)
        uint32_t x; __builtin_memcpy(&x, (const void*)at(v), 4);
        return x;
    }
    __device__ __forceinline__ uint64_t rd64(uint32_t v) const {
        if (split) return (uint64_t)rd32(v) | ((uint64_t)rd32(v + 4u) << 32);
        uint64_t x; __builtin_memcpy(&x, (const void*)at(v), 8);
        return x;
    }
    __device__ __forceinline__ uint32_t byte(uint32_t v) const { return *at(v); }
    // sixteen bytes of the block's own source with one load (hist == cur); they may end READ_SLACK bytes past the block
    __device__ __forceinline__ u32x4 rd128(uint32_t v) const {
        v = v < CUR - KB64 ? CUR - KB64 : (v > hi - 8u ? hi - 8u : v);
        u32x4 x; __builtin_memcpy(&x, (const void*)(cur + (int32_t)(v - CUR)), 16);
        return x;
    }
    __device__ __forceinline__ uint32_t hash5(uint32_t v) const { return hash5_of(rd64(v)); }
    // LZ4_hash5: ((sequence << 24) * 889523592379) >> (64 - HASHLOG), i.e. bits 28..39 of x * P for the five low bytes x
    // of the sequence.  With x = a + b * 2^32 and P = p0 + p1 * 2^32 those bits need one 32x32 product (both halves) and two
    // 8-bit products instead of a full 64-bit multiplication (32-bit multiplies issue at a quarter of the rate).
    static __device__ __forceinline__ uint32_t hash5_of(uint64_t s8) {
        static_assert(HASHLOG == 12, "bits 28..39");
        const uint32_t p0 = (uint32_t)(889523592379ULL & 0xffffffffu), p1 = (uint32_t)(889523592379ULL >> 32);
        const uint32_t a = (uint32_t)s8, b = (uint32_t)(s8 >> 32) & 0xffu;
        const uint32_t lo = a * p0, hi = __umulhi(a, p0);
        const uint32_t top = (hi + __umul24(a & 0xffu, p1 & 0xffu) + __umul24(b, p0 & 0xffu)) & 0xffu;     // bits 32..39
        return (lo >> 28) | (top << 4);
    }
};

// Number of equal bytes of ring[ip..] and ring[match..], stopping at `limit` (a virtual offset bounding ip).  256 bytes
// per step.
__device__ __forceinline__ uint32_t count_match(const Ring& R, uint32_t ip, uint32_t match, uint32_t limit, int lane, uint32_t n = 0) {
    for (;;) {
        const uint32_t p = ip + n + (uint32_t)lane * 4u;
        uint32_t avail = p < limit ? limit - p : 0u; if (avail > 4u) avail = 4u;
        // (unconditional reads: any address is inside the ring; `avail` bounds what counts)
        const uint32_t d = R.rd32(p) ^ R.rd32(match + n + (uint32_t)lane * 4u);
        uint32_t same = d ? (uint32_t)__builtin_ctz(d) >> 3 : 4u;
        if (same > avail) same = avail;
        const unsigned long long stop = __ballot(same < 4u);
        if (stop) {
            const int first = __builtin_ctzll(stop);
            return n + (uint32_t)first * 4u + (uint32_t)__builtin_amdgcn_readlane((int)same, first);
        }
        n += 256u;
    }
}

// Cooperative copy of n bytes from the ring (virtual offset v) to global memory: bytes up to the first 4-byte boundary
// of dst, then whole dwords, then the tail.
template <class Src>
__device__ __forceinline__ void copy_out(g_u8* dst, const Src& R, uint32_t v, uint32_t n, int lane) {
    if (n <= 64u) {                                   // the usual literal run: one byte per lane
        if ((uint32_t)lane < n) dst[lane] = (uint8_t)R.byte(v + (uint32_t)lane);
        return;
    }
    uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u);
    if (head > n) head = n;
    if ((uint32_t)lane < head) dst[lane] = (uint8_t)R.byte(v + (uint32_t)lane);
    const uint32_t words = (n - head) >> 2;
    g_u32* d32 = (g_u32*)(dst + head);
    for (uint32_t i = (uint32_t)lane; i < words; i += 64u) d32[i] = R.rd32(v + head + 4u * i);
    const uint32_t done = head + 4u * words;
    if (done + (uint32_t)lane < n) dst[done + lane] = (uint8_t)R.byte(v + done + (uint32_t)lane);
}

struct Stream {
    uint32_t current_offset;
    uint32_t dict_size;       // bytes of valid history held before the block
};

// Distance walked by the match search after k misses: it advances by 1 for the first 65 positions, then by
// (63 + j) >> 6 after the j-th miss (LZ4: step = searchMatchNb++ >> skipTrigger, searchMatchNb starting at 64).
__device__ __forceinline__ uint32_t walk(uint32_t k) {
    if (k == 0u) return 0u;
    const uint32_t m = 62u + k, q = m >> 6, r = m & 63u;
    return 1u + 32u * q * (q - 1u) + q * (r + 1u);
}

// LZ4_compress_generic (limitedOutput, byU32) on the block [CUR, CUR + n) of the ring.  `ext_dict` selects the
// usingExtDict rules (first block of a write; history = saved dictionary) against withPrefix64k (later blocks of the
// same write).  Returns the compressed size, or 0 when it does not fit in max_out bytes.
//
// The sequential algorithm between two matches is: [insert ip-2; test-and-insert ip] (after a match) or [insert the
// block's first byte] (at the start), then the miss walk from the next position.  One round evaluates 64 consecutive
// steps of that sequence: the `pre` leading lanes of the first round take the special positions, the others the walk.
__device__ uint32_t compress_block(Lds& L, const Ring& R, Stream& st, uint32_t n, g_u8* out, uint32_t max_out, bool ext_dict,
                                   bool dict_small, int lane) {
    const uint32_t start_index = st.current_offset;
    const uint32_t dict_size = st.dict_size;
    // index -> virtual offset: off = index - start_index + CUR
    const uint32_t bias = CUR - start_index;
    const uint32_t prefix_idx_limit = start_index - dict_size;
    // Virtual offset of the first reachable history byte.  Within one write the prefix can be longer than the 64 KiB held
    // here, but a match never reaches further back than 65535 bytes before the block (offsets are at most 65535 and
    // the catch-up loop stops at the block's first byte), so clamping does not change any decision.
    const uint32_t dict_start = CUR - (dict_size < KB64 ? dict_size : KB64);
    const uint32_t iend = CUR + n;
    const uint32_t mflimit_plus_one = iend - MFLIMIT + 1;
    const uint32_t matchlimit = iend - LASTLITERALS;
    const unsigned long long below = (1ull << lane) - 1ull;                 // lanes before this one
    const unsigned long long above = ~((2ull << lane) - 1ull);              // lanes after this one
    uint32_t ip = CUR, anchor = CUR;
    uint32_t op = 0;                                                        // bytes written to `out`
    uint32_t low_limit = ext_dict ? CUR : dict_start;
    uint32_t match = 0, token_pos = 0, token = 0;
    uint32_t pre = 1;                                                       // 1: block start, 2: after a match

    st.dict_size += n;
    st.current_offset += n;

    if ((int)n < LZ4_MINLENGTH) goto last_literals;

    for (;;) {
        // ---- match search
        bool found = false, immediate = false;
        {   const uint32_t w0 = ip + 1u;                                    // the miss walk starts behind ip
            for (uint32_t r0 = 0; ; r0 += 64u) {
                const uint32_t idx = r0 + (uint32_t)lane;
                uint32_t p; bool valid = true, tests = true;
                if (idx < pre) {
                    // pre == 1: insert the block's first position; pre == 2: insert ip - 2, then test-and-insert ip
                    p = (pre == 2u && idx == 0u) ? ip - 2u : ip;
                    tests = (pre == 2u && idx == 1u);
                } else {
                    const uint32_t k = idx - pre;
                    // the sequential walk leaves for the last literals when the position after this one passes the
                    // limit, before it inserts or tests this one
                    if (r0 == 0u) { p = w0 + k; valid = p + 1u <= mflimit_plus_one; }      // the first 65 steps advance by one
                    else { p = w0 + walk(k); valid = w0 + walk(k + 1u) <= mflimit_plus_one; }
                }
                const unsigned long long vm = __ballot(valid);
                // (a lane past the end reads wrapped ring bytes: harmless, its hash is replaced by its private slot)
                const uint64_t s8 = R.rd64(p);
                const uint32_t seq = (uint32_t)s8;
                const uint32_t h = valid ? Ring::hash5_of(s8) : (uint32_t)(1 << HASHLOG) + (uint32_t)lane;
                const uint32_t mine = p - bias;
                // Candidate of a lane: the table entry, unless an earlier lane of this round has the same hash (the
                // sequential walk would have seen that lane's insert).  Lanes with equal hashes address the same table
                // slot, so the slot itself finds each group's first lane: save the entry, store all-ones, atomic-min
                // the lane numbers in, read the winner back, and let the winners restore the entry.  (A lane's own
                // store must not be forwarded to its later load, other lanes write the same slot: the compiler
                // barriers make every access a real LDS operation; a volatile pointer would lose the LDS address space.)
                uint32_t* const table = L.hash;
                const uint32_t old_entry = table[h];
                const uint32_t old = old_entry & IDX_MASK;
                table[h] = 0xffffffffu;
                asm volatile("" ::: "memory");
                atomicMin(&table[h], (uint32_t)lane);
                asm volatile("" ::: "memory");
                // the four bytes at the table's candidate are asked for now: the read is under way during the rest of the
                // election (a candidate inside the round needs no read: its bytes are that lane's own sequence)
                const uint32_t old4 = R.rd32(old + bias);
                const uint32_t leader = table[h];
                asm volatile("" ::: "memory");
                if (leader == (uint32_t)lane) table[h] = old_entry;         // the table is as it was
                const unsigned long long repeats = __ballot(leader != (uint32_t)lane);   // lanes with an earlier lane of the same hash (never a lane past the end)
                const int first_invalid = (~vm) ? __builtin_ctzll(~vm) : 64;
                uint32_t mi = old;
                unsigned long long same = 1ull << lane;                     // (general path) lanes of this round with the same hash
                // Up to the first repeating lane f every candidate is the table entry; f's own candidate is its group's
                // first lane.  A repeated hash is nearly always a repeated sequence, i.e. f is a match, so the walk
                // usually stops at or before f and the lanes behind it never matter.
                const int f = repeats ? __builtin_ctzll(repeats) : 64;
                const uint32_t from_leader = (uint32_t)__shfl((int)mine, (int)leader);          // all lanes take part
                const uint32_t leader4 = (uint32_t)__shfl((int)seq, (int)leader);
                uint32_t mi4 = old4;                                        // the four bytes at the candidate
                if (lane == f) { mi = from_leader; mi4 = leader4; }
                auto test = [&](uint32_t cand, uint32_t cand4) {
                    // (the bytes were read unconditionally: the result only counts when the candidate is in reach)
                    const bool ok = !(dict_small && cand < prefix_idx_limit) && (cand + DISTANCE_MAX >= mine);
                    return valid && tests && ok && cand4 == seq;
                };
                unsigned long long hits = __ballot(test(mi, mi4));
                bool general = false;
                if (repeats) {
                    const unsigned long long upto_f = f >= 63 ? ~0ull : (2ull << f) - 1ull;
                    if (hits & upto_f) hits &= upto_f;                      // decided inside the exact prefix
                    else {
                        // The walk continues behind f: resolve every group exactly.  One step per hash value that
                        // occurs more than once: its lanes learn their group.
                        general = true;
                        unsigned long long losers = repeats;
                        while (losers) {
                            const uint32_t hl = (uint32_t)__builtin_amdgcn_readlane((int)h, __builtin_ctzll(losers));
                            const bool in_group = valid && h == hl;
                            const unsigned long long group = __ballot(in_group);
                            if (in_group) same = group;
                            losers &= ~group;
                        }
                        const unsigned long long prev = valid ? (same & below) : 0ull;
                        const int prev_lane = prev ? 63 - __builtin_clzll(prev) : lane;
                        const uint32_t from_lane = (uint32_t)__shfl((int)mine, prev_lane);      // all lanes take part
                        const uint32_t lane4 = (uint32_t)__shfl((int)seq, prev_lane);
                        mi = prev ? from_lane : old;
                        mi4 = prev ? lane4 : old4;
                        hits = __ballot(test(mi, mi4));
                    }
                }
                const int last_insert = hits ? __builtin_ctzll(hits) : first_invalid - 1;
                // the inserts of the lanes up to the stop, in order: per hash value the last of them stays
                if (general) {
                    const unsigned long long inserting = last_insert < 0 ? 0ull : (last_insert >= 63 ? ~0ull : (2ull << last_insert) - 1ull);
                    if (valid && lane <= last_insert && (same & above & inserting) == 0ull) table[h] = entry_of(mine, seq);
                } else {
                    // hashes are distinct among the lanes up to f, except f and its group's first lane
                    const uint32_t leader_of_f = (uint32_t)__builtin_amdgcn_readlane((int)leader, f < 64 ? f : 0);
                    const bool shadowed = f <= last_insert && (uint32_t)lane == leader_of_f;
                    if (valid && lane <= last_insert && !shadowed) table[h] = entry_of(mine, seq);
                }
                if (hits) {
                    const int e = __builtin_ctzll(hits);
                    ip = (uint32_t)__builtin_amdgcn_readlane((int)p, e);
                    const uint32_t match_index = (uint32_t)__builtin_amdgcn_readlane((int)mi, e);
                    match = match_index + bias;
                    if (ext_dict) low_limit = match_index < start_index ? dict_start : CUR;
                    found = true;
                    immediate = (pre == 2u && r0 == 0u && e == 1);         // the test right behind the previous match
                    break;
                }
                if (first_invalid < 64) break;
            }
        }
        if (!found) goto last_literals;
        // Everything that follows the search reads the ring at addresses known now: the bytes before ip and match (catch
        // up), the literals behind the anchor, the dwords behind the four matching bytes (match length).  The loads are
        // issued together so that they cost one LDS round trip instead of three dependent ones.  (Ring addresses wrap,
        // so the unconditional reads are always inside the ring; what they return is only used where it is meaningful.)
        const uint32_t fwd_ip = ip + MINMATCH, fwd_match = match + MINMATCH;
        const uint32_t cu_a = R.byte(ip - 1u - (uint32_t)lane), cu_b = R.byte(match - 1u - (uint32_t)lane);
        const uint32_t lit_byte = R.byte(anchor + (uint32_t)lane);
        const uint32_t cm_a = R.rd32(fwd_ip + (uint32_t)lane * 4u), cm_b = R.rd32(fwd_match + (uint32_t)lane * 4u);
        uint32_t caught_up = 0;
        if (immediate) { token_pos = op++; token = 0; }
        else {
            // ---- catch up: extend the match backwards, 64 bytes per round (the first round from the loads above)
            for (bool first = true; ip > anchor; first = false) {
                const uint32_t d = (uint32_t)lane + 1u;
                const bool room = (int32_t)(ip - anchor) >= (int32_t)d && (int32_t)(match - low_limit) >= (int32_t)d;
                const bool equal = first ? cu_a == cu_b : R.byte(ip - d) == R.byte(match - d);      // (first: wave-uniform)
                const bool ok = room && equal;
                const unsigned long long stop = __ballot(!ok);
                const uint32_t c = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                ip -= c; match -= c; caught_up += c;
                if (c < 64u) break;
            }
            const uint32_t lit = ip - anchor;
            token_pos = op++;
            if (op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > max_out) return 0;
            if (lit >= RUN_MASK) {
                uint32_t len = lit - RUN_MASK;
                token = RUN_MASK << 4;
                // length bytes: floor(len / 255) bytes of 255, then len % 255
                const uint32_t n255 = len / 255u;
                for (uint32_t i = (uint32_t)lane; i < n255; i += 64u) out[op + i] = 255;
                op += n255;
                if (lane == 0) out[op] = (uint8_t)(len - n255 * 255u);
                op++;
            } else token = lit << 4;
            if (lit <= 64u) { if ((uint32_t)lane < lit) out[op + lane] = (uint8_t)lit_byte; }      // the usual literal run
            else copy_out(out + op, R, anchor, lit, lane);
            op += lit;
        }
        {   const uint32_t offset = ip - match;
            if (lane == 0) { out[op] = (uint8_t)offset; out[op + 1] = (uint8_t)(offset >> 8); }
            op += 2;
        }
        {   // match length: the ring is flat in virtual offsets, so a match that starts in the history simply runs on
            // into the current block.  The bytes caught up backwards and the first four are known to match; counting
            // goes on behind them, the first 256 bytes from the loads above.
            uint32_t fwd;
            {   const uint32_t q = fwd_ip + (uint32_t)lane * 4u;
                uint32_t avail = q < matchlimit ? matchlimit - q : 0u; if (avail > 4u) avail = 4u;
                const uint32_t x = cm_a ^ cm_b;
                uint32_t same = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
                if (same > avail) same = avail;
                const unsigned long long stop = __ballot(same < 4u);
                if (stop) { const int first = __builtin_ctzll(stop); fwd = (uint32_t)first * 4u + (uint32_t)__builtin_amdgcn_readlane((int)same, first); }
                else fwd = count_match(R, fwd_ip, fwd_match, matchlimit, lane, 256u);
            }
            uint32_t match_code = caught_up + fwd;
            ip += match_code + MINMATCH;
            if (op + (1 + LASTLITERALS) + (match_code + 240) / 255 > max_out) return 0;
            if (match_code >= ML_MASK) {
                token += ML_MASK;
                match_code -= ML_MASK;
                const uint32_t n255 = match_code / 255u;
                for (uint32_t i = (uint32_t)lane; i < n255; i += 64u) out[op + i] = 255;
                op += n255;
                if (lane == 0) out[op] = (uint8_t)(match_code - n255 * 255u);
                op++;
            } else token += match_code;
            if (lane == 0) out[token_pos] = (uint8_t)token;
        }
        anchor = ip;
        if (ip >= mflimit_plus_one) break;
        pre = 2;
    }
last_literals:
    {   const uint32_t last_run = iend - anchor;
        if (op + last_run + 1 + ((last_run + 255 - RUN_MASK) / 255) > max_out) return 0;
        if (last_run >= RUN_MASK) {
            uint32_t acc = last_run - RUN_MASK;
            if (lane == 0) out[op] = (uint8_t)(RUN_MASK << 4);
            op++;
            const uint32_t n255 = acc / 255u;
            for (uint32_t i = (uint32_t)lane; i < n255; i += 64u) out[op + i] = 255;
            op += n255;
            if (lane == 0) out[op] = (uint8_t)(acc - n255 * 255u);
            op++;
        } else { if (lane == 0) out[op] = (uint8_t)(last_run << 4); op++; }
        copy_out(out + op, R, anchor, last_run, lane);
        op += last_run;
    }
    return op;
}

// Backward catch-up and match length of a match the short form could not decide, from memory: the bytes before ip and
// match and the dwords behind the four matching bytes are asked for together, then 64 / 256 bytes per step.  Every load
// is consumed in here.
__device__ __noinline__ uint2 long_match(const Ring& R, uint32_t ip, uint32_t match, uint32_t anchor, uint32_t low_limit, uint32_t matchlimit,
                                         bool immediate, int lane) {
    const uint32_t fwd_ip = ip + MINMATCH, fwd_match = match + MINMATCH;
    const uint32_t cu_a = R.byte(ip - 1u - (uint32_t)lane), cu_b = R.byte(match - 1u - (uint32_t)lane);
    const uint32_t cm_a = R.rd32(fwd_ip + (uint32_t)lane * 4u), cm_b = R.rd32(fwd_match + (uint32_t)lane * 4u);
    uint32_t caught_up = 0, fwd;
    if (!immediate)
        for (bool first = true; ip > anchor; first = false) {
            const uint32_t d = (uint32_t)lane + 1u;
            const bool room = (int32_t)(ip - anchor) >= (int32_t)d && (int32_t)(match - low_limit) >= (int32_t)d;
            const bool equal = first ? cu_a == cu_b : R.byte(ip - d) == R.byte(match - d);      // (first: wave-uniform)
            const unsigned long long stop = __ballot(!(room && equal));
            const uint32_t c = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
            ip -= c; match -= c; caught_up += c;
            if (c < 64u) break;
        }
    {   const uint32_t q = fwd_ip + (uint32_t)lane * 4u;
        uint32_t avail = q < matchlimit ? matchlimit - q : 0u; if (avail > 4u) avail = 4u;
        const uint32_t x = cm_a ^ cm_b;
        uint32_t same = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
        if (same > avail) same = avail;
        const unsigned long long stop = __ballot(same < 4u);
        if (stop) { const int first = __builtin_ctzll(stop); fwd = (uint32_t)first * 4u + (uint32_t)__builtin_amdgcn_readlane((int)same, first); }
        else fwd = count_match(R, fwd_ip, fwd_match, matchlimit, lane, 256u);
    }
    return make_uint2(caught_up, fwd);
}

// compress_block for a block whose history lies before it in the same source (every block of a write but the first:
// the withPrefix64k rules).  What the search reads of the block itself comes from the window in LDS, which follows the
// position a kilobyte at a time; the chunk behind it is always on its way, so the sequential reads cost LDS latency.
// With the test's four bytes a lane brings the sixteen bytes from four before its table candidate: a match whose
// candidate came from the table, or lies inside the window, and that is short (as nearly all are in rendered frames:
// backwards at most three bytes, forwards at most seven behind the first four) is decided without a trip to memory.
__device__ uint32_t compress_block_w(Lds& L, const Ring& R, Stream& st, uint32_t n, g_u8* out, uint32_t max_out, bool dict_small, int lane RH_STAMP_ARG) {
    const uint32_t start_index = st.current_offset;
    const uint32_t dict_size = st.dict_size;
    const uint32_t bias = CUR - start_index;
    const uint32_t prefix_idx_limit = start_index - dict_size;
    const uint32_t low_limit = CUR - (dict_size < KB64 ? dict_size : KB64);
    const uint32_t iend = CUR + n;
    const uint32_t mflimit_plus_one = iend - MFLIMIT + 1;
    const uint32_t matchlimit = iend - LASTLITERALS;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long above = ~((2ull << lane) - 1ull);
    uint32_t ip = CUR, anchor = CUR;
    uint32_t op = 0;
    uint32_t match = 0, token_pos = 0, token = 0;
    uint32_t pre = 1;
    uint32_t* const table = L.hash;
    const Window W{L.win};

    // The window holds [max(w_filled - WIN, w_start), w_filled); the chunk behind w_filled was asked for when the chunk
    // before it was stored.
    uint32_t w_filled = CUR, w_start = CUR;
    u32x4 w_next;
    auto w_load = [&](uint32_t from) -> u32x4 {
        uint32_t v = from + (uint32_t)lane * 16u;
        if (v > iend) v = iend;                                             // (a source is readable READ_SLACK bytes past its end)
        u32x4 x; __builtin_memcpy(&x, (const void*)(R.cur + (int32_t)(v - CUR)), 16);
        return x;
    };
    auto w_store = [&](uint32_t from, const u32x4& x) {
        const uint32_t i = ((from & (WIN - 1u)) >> 2) + (uint32_t)lane * 4u;
        *reinterpret_cast<u32x4*>(&L.win[i]) = x;
        if (i == 0u) { L.win[WIN_DWORDS] = x.x; L.win[WIN_DWORDS + 1u] = x.y; }
    };
    // makes [lo, hi) readable from the window (hi - lo <= WIN - WIN_CHUNK)
    auto w_cover = [&](uint32_t lo, uint32_t hi) {
        if (hi > iend + READ_SLACK) hi = iend + READ_SLACK;
        if (hi > w_filled + 2u * WIN_CHUNK) {
            // far ahead (a long match, or a walk with long steps): the whole window from lo's chunk on, in one trip
            const uint32_t base = lo & ~(WIN_CHUNK - 1u);
            const u32x4 a = w_load(base), b = w_load(base + WIN_CHUNK), c = w_load(base + 2u * WIN_CHUNK), d = w_load(base + 3u * WIN_CHUNK);
            w_store(base, a); w_store(base + WIN_CHUNK, b); w_store(base + 2u * WIN_CHUNK, c); w_store(base + 3u * WIN_CHUNK, d);
            w_start = base; w_filled = base + WIN;
            w_next = w_load(w_filled);
        }
        while (w_filled < hi) {
            w_store(w_filled, w_next);
            w_filled += WIN_CHUNK;
            w_next = w_load(w_filled);
        }
        asm volatile("" ::: "memory");
    };
    w_next = w_load(CUR);

    st.dict_size += n;
    st.current_offset += n;

    if ((int)n < LZ4_MINLENGTH) goto last_literals;

    for (;;) {
        // ---- The usual sequence in one straight piece of code: after a match, away from the block's end, the walk's first
        // round [insert ip - 2; test-and-insert ip; ip + 1 ...] finds a hit that the lanes before the first repeated hash
        // decide, the match is short, and the run before it fits one byte per lane.  Nothing is written (the election
        // leaves the table as it was) until all of that is known; if any of it fails, the general code below does the
        // sequence from the start.
        bool committed = false;
        uint32_t first_round = 0;
        if (pre == 2u && ip + 400u <= mflimit_plus_one && op + 96u <= max_out) {
            w_cover(ip - 8u, ip + 352u);
            const uint32_t p = ip + (uint32_t)lane - (lane == 0 ? 2u : 1u);
            const uint64_t s8 = W.rd64(p);
            const uint32_t seq = (uint32_t)s8;
            const uint32_t h = Ring::hash5_of(s8);
            const uint32_t my_tag = tag_of(seq);
            const uint32_t mine = p - bias;
            const uint32_t old_entry = table[h];
            const uint32_t old = old_entry & IDX_MASK, old_tag = old_entry >> IDX_BITS;
            table[h] = 0xffffffffu;
            asm volatile("" ::: "memory");
            atomicMin(&table[h], (uint32_t)lane);
            asm volatile("" ::: "memory");
            const uint32_t leader = table[h];
            asm volatile("" ::: "memory");
            if (leader == (uint32_t)lane) table[h] = old_entry;
            const unsigned long long repeats = __ballot(leader != (uint32_t)lane);
            const int f = repeats ? __builtin_ctzll(repeats) : 64;
            const uint32_t from_leader = (uint32_t)__shfl((int)mine, (int)leader);
            const uint32_t leader4 = (uint32_t)__shfl((int)seq, (int)leader);
            // lanes that can be the hit: the table's candidate is in reach and its bits are the position's own (or unknown);
            // the first repeated hash's candidate is the lane that leads its group, whose bytes are at hand
            uint32_t mi = old;
            bool maybe = old_tag == my_tag || old_tag == 0u, sure = false;
            if (lane == f) { mi = from_leader; maybe = sure = leader4 == seq; }
            maybe = maybe && lane != 0 && !(dict_small && mi < prefix_idx_limit) && (mi + DISTANCE_MAX >= mine);
            unsigned long long cands = __ballot(maybe);
            if (f < 63) cands &= (2ull << f) - 1ull;                         // behind the first repeated hash nothing is known
            const unsigned long long sure_mask = __ballot(sure);
            const uint32_t held = w_filled - w_start > WIN ? w_filled - WIN : w_start;
            RH_STAMP(7, (uint32_t)cands | (uint32_t)(cands >> 32));
            while (cands) {
                // the first of them: its candidate's bytes from four before it (window, or one read from memory)
                const int e = __builtin_ctzll(cands);
                const uint32_t ip_hit = ip + (uint32_t)e - 1u;
                const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)mi, e) + bias;
                uint32_t cb, c4; uint64_t cf;
                if (m >= held + 4u) { cb = W.rd32(m - 4u); c4 = W.rd32(m); cf = W.rd64(m + MINMATCH); }
                else if (m >= CUR - KB64 + 4u) {
                    const u32x4 c16 = R.rd128(m - 4u);
                    cb = c16.x; c4 = c16.y; cf = (uint64_t)c16.z | ((uint64_t)c16.w << 32);
                } else break;                                               // (no bytes before the history's start: general code)
                const uint32_t lit_byte = W.byte(ip + (uint32_t)lane);
                const uint32_t cur_back = W.rd32(ip_hit - 4u);
                const uint64_t cur_fwd = W.rd64(ip_hit + MINMATCH);
                const uint32_t seq_e = (uint32_t)__builtin_amdgcn_readlane((int)seq, e);
                if (!((sure_mask >> e) & 1ull) && (uint32_t)__builtin_amdgcn_readfirstlane((int)c4) != seq_e) { cands &= cands - 1ull; continue; }
                cb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cb ^ cur_back));
                const uint64_t y = cf ^ cur_fwd;
                const uint32_t y_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)y), y_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(y >> 32));
                // backwards: at most to the anchor and to the history's first byte; four equal bytes with room left
                // mean that the catch-up goes on (general code); forwards likewise with eight (the block's end is far)
                const int32_t ra = (int32_t)(ip_hit - ip), rb = (int32_t)(m - low_limit);
                const uint32_t room = (uint32_t)(ra < rb ? ra : (rb < 0 ? 0 : rb));
                uint32_t back = cb ? (uint32_t)__builtin_clz(cb) >> 3 : 4u;
                const bool more_back = back < room && back == 4u;
                if (back > room) back = room;
                uint32_t fw = y_lo ? (uint32_t)__builtin_ctz(y_lo) >> 3 : (y_hi ? 4u + ((uint32_t)__builtin_ctz(y_hi) >> 3) : 8u);
                if (more_back) break;
                if (fw == 8u) {
                    // the match goes on: the next 256 bytes of both sides, when the candidate is inside the window too
                    if (m < held + 4u) break;
                    const uint32_t d = W.rd32(ip_hit + 12u + (uint32_t)lane * 4u) ^ W.rd32(m + 12u + (uint32_t)lane * 4u);
                    const unsigned long long stop = __ballot(d != 0u);
                    if (!stop) break;                                       // (longer still: general code)
                    const int first = __builtin_ctzll(stop);
                    fw += (uint32_t)first * 4u + ((uint32_t)__builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)d, first)) >> 3);
                }
                // ---- decided: the inserts of the lanes up to the hit (per hash value the last of them stays) ...
                const uint32_t leader_of_f = (uint32_t)__builtin_amdgcn_readlane((int)leader, f < 64 ? f : 0);
                const bool shadowed = f <= e && (uint32_t)lane == leader_of_f;
                if (lane <= e && !shadowed) table[h] = mine | (my_tag << IDX_BITS);
                // ... and the sequence: token, one length byte for a run of 15 or more, the run, the offset
                const uint32_t lit = ip_hit - back - ip;
                const uint32_t long_run = lit >= RUN_MASK ? 1u : 0u;
                const uint32_t match_code = back + fw;
                const uint32_t long_match = match_code >= ML_MASK ? 1u : 0u;
                const uint32_t tok = ((long_run ? RUN_MASK : lit) << 4) + (long_match ? ML_MASK : match_code);
                const uint32_t offset = ip_hit - m;
                g_u8* const o = out + op;
                if ((uint32_t)lane <= long_run) o[lane] = (uint8_t)(lane == 0 ? tok : lit - RUN_MASK);
                if ((uint32_t)lane < lit) o[1u + long_run + (uint32_t)lane] = (uint8_t)lit_byte;
                // (a match code of 15 or more: one length byte, its length is below 270 here)
                if ((uint32_t)lane < 2u + long_match) o[1u + long_run + lit + (uint32_t)lane] = (uint8_t)(lane == 2 ? match_code - ML_MASK : offset >> (8 * lane));
                op += 3u + long_run + lit + long_match;
                ip = ip_hit + MINMATCH + fw;
                anchor = ip;
                committed = true;
                break;
            }
            if (committed) {
                RH_STAMP(8, ip);
                if (ip >= mflimit_plus_one) break;
                continue;
            }
            if (!cands && !repeats) {
                // nothing in this round can match and its hashes are all different: every lane inserts, and the general
                // code goes on with the walk's second round
                table[h] = mine | (my_tag << IDX_BITS);
                first_round = 64u;
            }
            RH_STAMP(9, ip);
        }
        // ---- match search (see compress_block): 64 steps of the sequential walk per round
        bool found = false, immediate = false, near = false, from_table = false;
        uint32_t c_back = 0, c_lo = 0, c_hi = 0;
        {   const uint32_t w0 = ip + 1u;
            for (uint32_t r0 = first_round; ; r0 += 64u) {
                uint32_t p, lo, hi; bool valid, tests;
                if (r0 == 0u) {
                    // the first `pre` lanes take the positions that come before the walk: [the block's first position]
                    // or [insert ip - 2; test-and-insert ip]; the walk's first 65 steps advance by one
                    const bool special = (uint32_t)lane < pre;
                    const uint32_t pw = w0 + (uint32_t)lane - pre;
                    p = special ? ((pre == 2u && lane == 0) ? ip - 2u : ip) : pw;
                    valid = special || pw + 1u <= mflimit_plus_one;
                    tests = !special || (pre == 2u && lane == 1);
                    lo = ip - 8u; hi = w0 + 64u + 16u;
                } else {
                    const uint32_t k = r0 + (uint32_t)lane - pre;
                    p = w0 + walk(k); valid = w0 + walk(k + 1u) <= mflimit_plus_one; tests = true;
                    lo = w0 + walk(r0 - pre) - 8u; hi = w0 + walk(r0 + 64u - pre) + 16u;
                }
                w_cover(lo, hi);
                const unsigned long long vm = __ballot(valid);
                const uint64_t s8 = W.rd64(p);
                const uint32_t seq = (uint32_t)s8;
                // (every lane hashes, so that the window reads are not split by a branch around the hash)
                uint32_t hashed = Ring::hash5_of(s8);
                asm volatile("" : "+v"(hashed));
                const uint32_t h = valid ? hashed : (uint32_t)(1 << HASHLOG) + (uint32_t)lane;
                const uint32_t mine = p - bias;
                RH_STAMP(0, h);
                const uint32_t old_entry = table[h];
                const uint32_t old = old_entry & IDX_MASK;
                table[h] = 0xffffffffu;
                asm volatile("" ::: "memory");
                atomicMin(&table[h], (uint32_t)lane);
                asm volatile("" ::: "memory");
                // the bytes around the table's candidate are asked for now: the read is under way during the rest of the
                // election (a candidate within four bytes of the history's start has no bytes before it)
                const uint32_t cv = old + bias;
                const bool cand_edge = cv < CUR - KB64 + 4u;
                const u32x4 cand16 = R.rd128(cand_edge ? cv : cv - 4u);
                const uint32_t leader = table[h];
                asm volatile("" ::: "memory");
                if (leader == (uint32_t)lane) table[h] = old_entry;
                const unsigned long long repeats = __ballot(leader != (uint32_t)lane);
                const int first_invalid = (~vm) ? __builtin_ctzll(~vm) : 64;
                uint32_t mi = old;
                unsigned long long same = 1ull << lane;
                const int f = repeats ? __builtin_ctzll(repeats) : 64;
                const uint32_t from_leader = (uint32_t)__shfl((int)mine, (int)leader);
                const uint32_t leader4 = (uint32_t)__shfl((int)seq, (int)leader);
                uint32_t mi4 = cand_edge ? cand16.x : cand16.y;
                if (lane == f) { mi = from_leader; mi4 = leader4; }
                auto test = [&](uint32_t cand, uint32_t cand4) {
                    const bool ok = !(dict_small && cand < prefix_idx_limit) && (cand + DISTANCE_MAX >= mine);
                    return valid && tests && ok && cand4 == seq;
                };
                unsigned long long hits = __ballot(test(mi, mi4));
                RH_STAMP(1, (uint32_t)hits | (uint32_t)(hits >> 32));
                bool general = false;
                if (repeats) {
                    const unsigned long long upto_f = f >= 63 ? ~0ull : (2ull << f) - 1ull;
                    if (hits & upto_f) hits &= upto_f;
                    else {
                        general = true;
                        unsigned long long losers = repeats;
                        while (losers) {
                            const uint32_t hl = (uint32_t)__builtin_amdgcn_readlane((int)h, __builtin_ctzll(losers));
                            const bool in_group = valid && h == hl;
                            const unsigned long long group = __ballot(in_group);
                            if (in_group) same = group;
                            losers &= ~group;
                        }
                        const unsigned long long prev = valid ? (same & below) : 0ull;
                        const int prev_lane = prev ? 63 - __builtin_clzll(prev) : lane;
                        const uint32_t from_lane = (uint32_t)__shfl((int)mine, prev_lane);
                        const uint32_t lane4 = (uint32_t)__shfl((int)seq, prev_lane);
                        if (prev) { mi = from_lane; mi4 = lane4; } else { mi = old; mi4 = cand_edge ? cand16.x : cand16.y; }
                        hits = __ballot(test(mi, mi4));
                    }
                }
                const int last_insert = hits ? __builtin_ctzll(hits) : first_invalid - 1;
                if (general) {
                    const unsigned long long inserting = last_insert < 0 ? 0ull : (last_insert >= 63 ? ~0ull : (2ull << last_insert) - 1ull);
                    if (valid && lane <= last_insert && (same & above & inserting) == 0ull) table[h] = entry_of(mine, seq);
                } else {
                    const uint32_t leader_of_f = (uint32_t)__builtin_amdgcn_readlane((int)leader, f < 64 ? f : 0);
                    const bool shadowed = f <= last_insert && (uint32_t)lane == leader_of_f;
                    if (valid && lane <= last_insert && !shadowed) table[h] = entry_of(mine, seq);
                }
                RH_STAMP(2, last_insert);
                if (hits) {
                    const int e = __builtin_ctzll(hits);
                    ip = (uint32_t)__builtin_amdgcn_readlane((int)p, e);
                    const uint32_t match_index = (uint32_t)__builtin_amdgcn_readlane((int)mi, e);
                    match = match_index + bias;
                    found = true;
                    immediate = (pre == 2u && r0 == 0u && e == 1);
                    // where the bytes around the candidate are: in the window, or in the registers of the lane that hit
                    const uint32_t held = w_filled - w_start > WIN ? w_filled - WIN : w_start;
                    near = match >= held + 4u;
                    from_table = match_index == (uint32_t)__builtin_amdgcn_readlane((int)old, e) && __builtin_amdgcn_readlane((int)cand_edge, e) == 0;
                    c_back = (uint32_t)__builtin_amdgcn_readlane((int)cand16.x, e);
                    c_lo = (uint32_t)__builtin_amdgcn_readlane((int)cand16.z, e);
                    c_hi = (uint32_t)__builtin_amdgcn_readlane((int)cand16.w, e);
                    break;
                }
                if (first_invalid < 64) break;
            }
        }
        if (!found) goto last_literals;
        // ---- lengths
        uint32_t caught_up = 0, fwd = 0;
        bool decided = false;
        const uint32_t lit_byte = W.byte(anchor + (uint32_t)lane);          // (used when the run is short and inside the window)
        if (near || from_table) {
            const uint32_t cur_back = W.rd32(ip - 4u);
            const uint64_t cur_fwd = W.rd64(ip + MINMATCH);
            if (near) { c_back = W.rd32(match - 4u); const uint64_t c = W.rd64(match + MINMATCH); c_lo = (uint32_t)c; c_hi = (uint32_t)(c >> 32); }
            bool more = false;
            uint32_t back = 0;
            if (!immediate) {
                const int32_t ra = (int32_t)(ip - anchor), rb = (int32_t)(match - low_limit);
                const uint32_t room = (uint32_t)(ra < rb ? (ra < 0 ? 0 : ra) : (rb < 0 ? 0 : rb));
                const uint32_t x = cur_back ^ c_back;
                back = x ? (uint32_t)__builtin_clz(x) >> 3 : 4u;
                if (back >= room) back = room; else if (back == 4u) more = true;
            }
            const uint64_t y = cur_fwd ^ ((uint64_t)c_lo | ((uint64_t)c_hi << 32));
            uint32_t fw = y ? (uint32_t)__builtin_ctzll(y) >> 3 : 8u;
            const uint32_t avail = matchlimit - (ip + MINMATCH);
            if (fw >= avail) fw = avail; else if (fw == 8u) more = true;
            if (!more) { decided = true; caught_up = back; fwd = fw; }
        }
        RH_STAMP(3, fwd + caught_up + lit_byte);
        if (!decided) {
            // (what a function returns counts as different from lane to lane: say that it is not)
            const uint2 r = long_match(R, ip, match, anchor, low_limit, matchlimit, immediate, lane);
            caught_up = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.x); fwd = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.y);
        }
        ip -= caught_up; match -= caught_up;
        RH_STAMP(4, caught_up + fwd);
        // ---- the sequence: token, literal run, offset, match length
        if (immediate) { token_pos = op++; token = 0; }
        else {
            const uint32_t lit = ip - anchor;
            token_pos = op++;
            if (op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > max_out) return 0;
            if (lit >= RUN_MASK) {
                uint32_t len = lit - RUN_MASK;
                token = RUN_MASK << 4;
                const uint32_t n255 = len / 255u;
                for (uint32_t i = (uint32_t)lane; i < n255; i += 64u) out[op + i] = 255;
                op += n255;
                if (lane == 0) out[op] = (uint8_t)(len - n255 * 255u);
                op++;
            } else token = lit << 4;
            // (a hit far along the walk leaves the anchor behind the window, however short the run after the catch-up)
            const uint32_t held = w_filled - w_start > WIN ? w_filled - WIN : w_start;
            if (anchor < held) copy_out(out + op, R, anchor, lit, lane);
            else if (lit <= 64u) { if ((uint32_t)lane < lit) out[op + lane] = (uint8_t)lit_byte; }
            else copy_out(out + op, W, anchor, lit, lane);
            op += lit;
        }
        {   const uint32_t offset = ip - match;
            if (lane == 0) { out[op] = (uint8_t)offset; out[op + 1] = (uint8_t)(offset >> 8); }
            op += 2;
        }
        {   uint32_t match_code = caught_up + fwd;
            ip += match_code + MINMATCH;
            if (op + (1 + LASTLITERALS) + (match_code + 240) / 255 > max_out) return 0;
            if (match_code >= ML_MASK) {
                token += ML_MASK;
                match_code -= ML_MASK;
                const uint32_t n255 = match_code / 255u;
                for (uint32_t i = (uint32_t)lane; i < n255; i += 64u) out[op + i] = 255;
                op += n255;
                if (lane == 0) out[op] = (uint8_t)(match_code - n255 * 255u);
                op++;
            } else token += match_code;
            if (lane == 0) out[token_pos] = (uint8_t)token;
        }
        anchor = ip;
        RH_STAMP(5, ip);
        if (ip >= mflimit_plus_one) break;
        pre = 2;
    }
last_literals:
    {   const uint32_t last_run = iend - anchor;
        if (op + last_run + 1 + ((last_run + 255 - RUN_MASK) / 255) > max_out) return 0;
        if (last_run >= RUN_MASK) {
            uint32_t acc = last_run - RUN_MASK;
            if (lane == 0) out[op] = (uint8_t)(RUN_MASK << 4);
            op++;
            const uint32_t n255 = acc / 255u;
            for (uint32_t i = (uint32_t)lane; i < n255; i += 64u) out[op + i] = 255;
            op += n255;
            if (lane == 0) out[op] = (uint8_t)(acc - n255 * 255u);
            op++;
        } else { if (lane == 0) out[op] = (uint8_t)(last_run << 4); op++; }
        copy_out(out + op, R, anchor, last_run, lane);
        op += last_run;
    }
    return op;
}

} // namespace

// One workgroup (one wave) per job.  A job is a sequence of writes to one stream; every write is one
// LZ4F_compressUpdate: full 64 KiB blocks, then the remainder, then the dictionary save.  Every source is readable up
// to READ_SLACK bytes behind its end.
extern "C" __global__ __launch_bounds__(64) void rh_lz4f_compress(const rh_lz4_job* jobs, rh_lz4_stream_state* states, uint32_t top_priority) {
    extern __shared__ uint8_t lds_raw[];
    Lds& L = *reinterpret_cast<Lds*>(lds_raw);
    const int lane = (int)threadIdx.x;
    // A stream is one long chain of dependent steps: whenever its wave can issue, it should -- before the waves of the
    // rendering kernels that share its SIMD, which have others to stand in for them.
    if (top_priority) __builtin_amdgcn_s_setprio(3);
    // (a reference, not a copy: a local copy with its run-time indexed arrays would live in per-lane memory, and
    // everything derived from it -- block sizes, positions, every branch of the match loop -- would be treated as
    // varying across the lanes instead of running on the scalar unit)
    const rh_lz4_job& job = jobs[blockIdx.x];
    rh_lz4_stream_state& S = states[job.stream];
    g_u8* const dict = (g_u8*)S.dict;

    // bring the hash table into LDS
    for (int i = lane; i < (1 << HASHLOG); i += 64) L.hash[i] = S.hash[i];
    Stream st; st.current_offset = __builtin_amdgcn_readfirstlane(S.current_offset); st.dict_size = __builtin_amdgcn_readfirstlane(S.dict_size);
    __syncthreads();

    g_u8* out = (g_u8*)job.out;
    uint64_t op = 0;
#ifdef RH_LZ4_STAMPS
    Stamps stamps; stamps.start();
#endif
    for (uint32_t wi = 0; wi < job.n_writes; wi++) {
        g_cu8* src = (g_cu8*)job.write_ptr[wi];
        const uint64_t size = job.write_len[wi];
        if (size == 0) continue;
        bool first_block = true;
        for (uint64_t done = 0; done < size; ) {
            const uint32_t n = __builtin_amdgcn_readfirstlane((uint32_t)((size - done) < KB64 ? (size - done) : KB64));
            // LZ4_renormDictT, before every block instead of once in 2 GiB: the stream's offset goes back to 64 KiB and the
            // table's indices with it (an index that falls below zero is further back than any match can reach: its entry
            // is cleared), so that an index of the block fits IDX_BITS bits.  The decisions of the search only compare
            // indices with one another, so they do not change.
            if (st.current_offset > KB64) {
                const uint32_t delta = st.current_offset - KB64;
                __syncthreads();
                for (int i = lane; i < (1 << HASHLOG); i += 64) { const uint32_t e = L.hash[i]; L.hash[i] = (e & IDX_MASK) < delta ? 0u : e - delta; }
                st.current_offset = KB64;
                if (st.dict_size > KB64) st.dict_size = KB64;
            }
            if (first_block) {
                // a dictionary shorter than 4 bytes is dropped (LZ4_compress_fast_continue)
                if (st.dict_size - 1u < 3u) st.dict_size = 0;
            }
            // the block where it is; before it the previous block of the same write, or the stream's saved dictionary
            Ring R;
            R.cur = src + done;
            R.hist = first_block ? (g_cu8*)(dict + KB64) : R.cur;
            R.split = first_block;
            R.hi = CUR + n + READ_SLACK - 8u;
            __syncthreads();
            const bool small = st.dict_size < KB64 && st.dict_size < st.current_offset;
            const bool ext = first_block;
            g_u8* blk = out + op;
            Stream before = st;
            RH_STAMP(6, n);
            uint32_t c = first_block ? compress_block(L, R, st, n, blk + 4, n - 1, ext, small, lane)
                                     : compress_block_w(L, R, st, n, blk + 4, n - 1, small, lane RH_STAMP_PASS);
            if (c == 0) { st = before; st.dict_size += n; st.current_offset += n; }      // state advances even when the block is stored
            if (ext) st.dict_size = n;                      // external-dictionary mode: the block becomes the new dictionary
            __syncthreads();
            if (c == 0) {
                c = n;
                if (lane == 0) { const uint32_t h = n | BLOCK_UNCOMPRESSED_FLAG; blk[0] = (uint8_t)h; blk[1] = (uint8_t)(h >> 8); blk[2] = (uint8_t)(h >> 16); blk[3] = (uint8_t)(h >> 24); }
                copy_out(blk + 4, R, CUR, n, lane);
            } else if (lane == 0) { blk[0] = (uint8_t)c; blk[1] = (uint8_t)(c >> 8); blk[2] = (uint8_t)(c >> 16); blk[3] = (uint8_t)(c >> 24); }
            op += 4u + c;
            done += n;
            first_block = false;
            if (done >= size) {
                // LZ4_saveDict at the end of the update: the last `keep` bytes of what the write's blocks saw as [history |
                // data] -- after the first block that is the write's own data -- go to the end of the dictionary buffer
                const uint32_t keep = st.dict_size < KB64 ? st.dict_size : KB64;
                copy_out(dict + (KB64 - keep), R, CUR + n - keep, keep, lane);
                st.dict_size = keep;
                // (the next write of this job reads the dictionary through other lanes than the ones that wrote it)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
    }

#ifdef RH_LZ4_STAMPS
    stamps.mark(6, 0u); stamps.flush(lane);
#endif
    // persist the stream state (the dictionary is already in place)
    __syncthreads();
    for (int i = lane; i < (1 << HASHLOG); i += 64) S.hash[i] = L.hash[i];
    if (lane == 0) { S.current_offset = st.current_offset; S.dict_size = st.dict_size; *job.out_len = op; }
}

// ---- LZ4F block decoding (the inverse; reference src/decompressor.rs:186-249 reads its stream files through
// lz4f::BufReadDecompressor) ----------------------------------------------------------------------------------------------
// One wavefront per stream walks the sequences of its blocks in order; what is parallel is inside a sequence: the
// look-ahead (the next 64 compressed bytes arrive with one LDS read: token, length bytes, short literal runs and the
// offset are picked out of the lanes), the literal copy and the match copy (an overlapping match repeats with the period
// of its offset, so every byte's source lies before the match).  Decoded bytes go to a 64 KiB ring in LDS, which is
// also where matches read from, and leave for memory one block at a time in 16-byte pieces.
namespace {

constexpr uint32_t IN_WINDOW = 8192u, IN_HALF = IN_WINDOW / 2u;

struct DecodeLds {
    uint8_t ring[KB64];                 // byte at stream position p at index p mod 64 KiB
    uint8_t in[IN_WINDOW];              // compressed byte at source position q at index q mod 8 KiB
};

} // namespace

extern "C" __global__ __launch_bounds__(64) void rh_lz4f_decode(rh_lz4_decode_job* jobs, rh_lz4_decode_state* states) {
    extern __shared__ uint8_t lds_raw[];
    DecodeLds& L = *reinterpret_cast<DecodeLds*>(lds_raw);
    const uint32_t lane = threadIdx.x;
    rh_lz4_decode_job& job = jobs[blockIdx.x];
    rh_lz4_decode_state& S = states[job.stream];
    g_cu8* const src = (g_cu8*)job.src;
    g_u8* const out = (g_u8*)job.out;
    const uint32_t src_len = __builtin_amdgcn_readfirstlane(job.src_len), n_blocks = __builtin_amdgcn_readfirstlane(job.n_blocks);
    const uint32_t trailer = __builtin_amdgcn_readfirstlane(job.block_trailer);
    const uint64_t out_cap = job.out_cap, before = job.decoded_before;
    const uint32_t ring0 = (uint32_t)before & (KB64 - 1u);                // ring index of out[0]

    // the stream's window so far
    if (before != 0)
        for (uint32_t i = lane * 16u; i < KB64; i += 64u * 16u) *reinterpret_cast<uint4*>(L.ring + i) = *reinterpret_cast<const uint4*>(S.window + i);
    // compressed bytes arrive in halves of the input window: [loaded - 8 KiB, loaded) is held
    uint32_t loaded = 0;
    auto refill = [&](uint32_t cursor) {
        while (loaded < src_len + 64u && cursor + IN_HALF > loaded) {             // (the source is readable up to a multiple of 4096)
#pragma unroll
            for (uint32_t k = 0; k < IN_HALF / (64u * 16u); k++) {
                const uint32_t q = loaded + (k * 64u + lane) * 16u;
                *reinterpret_cast<u32x4*>(L.in + (q & (IN_WINDOW - 1u))) = *(g_cu32x4*)(src + q);
            }
            loaded += IN_HALF;
        }
    };
    auto in_byte = [&](uint32_t q) -> uint32_t { return L.in[q & (IN_WINDOW - 1u)]; };

    uint32_t cursor = 0;                // source position
    uint64_t op = 0;                    // bytes decoded by this job
    uint64_t good = 0;                  // ... up to the end of the last block decoded completely
    uint32_t status = RH_LZ4_DECODE_OK, blocks_done = 0;
    refill(0);
    __syncthreads();
    for (uint32_t blk = 0; blk < n_blocks && status == RH_LZ4_DECODE_OK; blk++) {
        refill(cursor);
        const uint32_t header = __builtin_amdgcn_readfirstlane(in_byte(cursor) | (in_byte(cursor + 1u) << 8) | (in_byte(cursor + 2u) << 16) | (in_byte(cursor + 3u) << 24));
        const uint32_t bsize = header & 0x7fffffffu;
        cursor += 4u;
        const uint32_t bend = cursor + bsize;
        const uint64_t block_start = op;
        if (bsize == 0u || bsize > src_len || bend > src_len) { status = RH_LZ4_DECODE_MALFORMED; break; }
        if (header & BLOCK_UNCOMPRESSED_FLAG) {
            if (bsize > KB64) { status = RH_LZ4_DECODE_MALFORMED; break; }
            if (op + bsize > out_cap) { status = RH_LZ4_DECODE_NO_ROOM; break; }
            for (uint32_t done = 0; done < bsize; ) {
                refill(cursor);
                const uint32_t piece = min(bsize - done, loaded - cursor);
                for (uint32_t i = lane; i < piece; i += 64u) L.ring[(ring0 + (uint32_t)op + i) & (KB64 - 1u)] = (uint8_t)in_byte(cursor + i);
                cursor += piece; done += piece; op += piece;
            }
        } else {
            while (cursor < bend) {
                refill(cursor);
                // look-ahead: lane i holds the compressed byte at cursor + i
                const uint32_t la = in_byte(cursor + lane);
                const uint32_t token = (uint32_t)__builtin_amdgcn_readfirstlane((int)la);
                uint32_t at = 1u;                                               // look-ahead lane of the next unread byte
                uint32_t lit = token >> 4;
                bool inside = true;                                             // everything read so far came from the look-ahead
                if (lit == 15u) {
                    // length bytes: 255s, then one byte below 255
                    const unsigned long long ends = __ballot(lane >= at && la != 255u);
                    if (ends) {
                        const uint32_t e = (uint32_t)__builtin_ctzll(ends);
                        lit = 15u + 255u * (e - at) + (uint32_t)__builtin_amdgcn_readlane((int)la, (int)e);
                        at = e + 1u;
                    } else {
                        // longer than the look-ahead: byte by byte
                        inside = false;
                        uint32_t q = cursor + 1u, x;
                        do {
                            if (q >= bend) { status = RH_LZ4_DECODE_MALFORMED; break; }
                            if (q + 64u > loaded) refill(q);
                            x = (uint32_t)__builtin_amdgcn_readfirstlane((int)in_byte(q)); q++; lit += x;
                        } while (x == 255u);
                        if (status != RH_LZ4_DECODE_OK) break;
                        cursor = q; at = 0u;
                    }
                }
                if (inside) { if (cursor + at > bend) { status = RH_LZ4_DECODE_MALFORMED; break; } }
                const uint32_t lit_pos = inside ? cursor + at : cursor;          // source position of the first literal
                if (lit > bend - lit_pos) { status = RH_LZ4_DECODE_MALFORMED; break; }
                if (op - block_start + lit > KB64) { status = RH_LZ4_DECODE_MALFORMED; break; }
                if (op + lit > out_cap) { status = RH_LZ4_DECODE_NO_ROOM; break; }
                const uint32_t rop = ring0 + (uint32_t)op;
                if (inside && at + lit + 2u <= 64u) {
                    // the literals and the offset are in the look-ahead
                    if (lane >= at && lane < at + lit) L.ring[(rop + lane - at) & (KB64 - 1u)] = (uint8_t)la;
                    at += lit;
                    cursor += at; at = 0u;
                    op += lit;
                    if (cursor >= bend) break;                                   // last sequence of the block: literals only
                    // (the offset bytes are the two look-ahead lanes behind the literals)
                } else {
                    uint32_t q = lit_pos;
                    for (uint32_t done = 0; done < lit; ) {
                        refill(q);
                        const uint32_t piece = min(lit - done, loaded - q);
                        for (uint32_t i = lane; i < piece; i += 64u) L.ring[(rop + done + i) & (KB64 - 1u)] = (uint8_t)in_byte(q + i);
                        q += piece; done += piece;
                    }
                    cursor = q; op += lit;
                    if (cursor >= bend) break;
                    refill(cursor);
                }
                // offset and match length: a fresh look-ahead unless the old one still covers them
                if (bend - cursor < 2u) { status = RH_LZ4_DECODE_MALFORMED; break; }
                const uint32_t lb = in_byte(cursor + lane);
                const uint32_t offset = (uint32_t)__builtin_amdgcn_readfirstlane((int)lb) | ((uint32_t)__builtin_amdgcn_readlane((int)lb, 1) << 8);
                uint32_t ml = token & 15u, used = 2u;
                if (ml == 15u) {
                    const unsigned long long ends = __ballot(lane >= 2u && lb != 255u);
                    if (ends) {
                        const uint32_t e = (uint32_t)__builtin_ctzll(ends);
                        ml = 15u + 255u * (e - 2u) + (uint32_t)__builtin_amdgcn_readlane((int)lb, (int)e);
                        used = e + 1u;
                    } else {
                        uint32_t q = cursor + 2u, x;
                        do {
                            if (q >= bend) { status = RH_LZ4_DECODE_MALFORMED; break; }
                            if (q + 64u > loaded) refill(q);
                            x = (uint32_t)__builtin_amdgcn_readfirstlane((int)in_byte(q)); q++; ml += x;
                        } while (x == 255u);
                        if (status != RH_LZ4_DECODE_OK) break;
                        used = q - cursor;
                    }
                }
                ml += 4u;
                cursor += used;
                if (cursor > bend) { status = RH_LZ4_DECODE_MALFORMED; break; }
                if (offset == 0u || (uint64_t)offset > before + op) { status = RH_LZ4_DECODE_BAD_OFFSET; break; }
                if (op - block_start + ml > KB64) { status = RH_LZ4_DECODE_MALFORMED; break; }
                if (op + ml > out_cap) { status = RH_LZ4_DECODE_NO_ROOM; break; }
                // match copy
                const uint32_t dst = ring0 + (uint32_t)op, from = dst - offset;
                if (offset >= 64u) {
                    // 64 bytes per step: their sources were written by earlier steps (LDS operations of a wave finish in order)
                    for (uint32_t i = lane; i < ml; i += 64u) L.ring[(dst + i) & (KB64 - 1u)] = L.ring[(from + i) & (KB64 - 1u)];
                } else {
                    // whole periods per step: lane % offset with a multiplication (exact for lanes and offsets below 64)
                    const uint32_t recip = (65536u + offset - 1u) / offset;
                    const uint32_t phase = lane - ((lane * recip) >> 16) * offset;
                    const uint32_t per_step = (64u / offset) * offset;
                    const uint32_t v = L.ring[(from + phase) & (KB64 - 1u)];       // the period itself never changes
                    for (uint32_t base = 0; base < ml; base += per_step)
                        if (lane < per_step && base + lane < ml) L.ring[(dst + base + lane) & (KB64 - 1u)] = (uint8_t)v;
                }
                op += ml;
            }
            if (status != RH_LZ4_DECODE_OK) break;
            if (cursor != bend) { status = RH_LZ4_DECODE_MALFORMED; break; }
        }
        cursor = bend + trailer;
        // the block leaves for memory
        __syncthreads();
        {
            const uint32_t n = (uint32_t)(op - block_start);
            const uint32_t r = ring0 + (uint32_t)block_start;
            g_u8* const o = out + block_start;
            uint32_t head = (uint32_t)((16u - ((uintptr_t)o & 15u)) & 15u); if (head > n) head = n;
            if (lane < head) o[lane] = L.ring[(r + lane) & (KB64 - 1u)];
            const uint32_t pieces = (n - head) >> 4;
            for (uint32_t i = lane; i < pieces; i += 64u) {
                const uint32_t at = r + head + i * 16u;
                u32x4 v;
                if ((at & 15u) == 0u && ((at & (KB64 - 1u)) + 16u <= KB64)) v = *reinterpret_cast<const u32x4*>(L.ring + (at & (KB64 - 1u)));
                else {
                    uint32_t w[4];
                    for (int k = 0; k < 4; k++) {
                        const uint32_t a = at + 4u * (uint32_t)k;
                        w[k] = (uint32_t)L.ring[a & (KB64 - 1u)] | ((uint32_t)L.ring[(a + 1u) & (KB64 - 1u)] << 8) | ((uint32_t)L.ring[(a + 2u) & (KB64 - 1u)] << 16) | ((uint32_t)L.ring[(a + 3u) & (KB64 - 1u)] << 24);
                    }
                    v = u32x4{w[0], w[1], w[2], w[3]};
                }
                *(__attribute__((address_space(1))) u32x4*)(o + head + i * 16u) = v;
            }
            const uint32_t done = head + pieces * 16u;
            if (done + lane < n) o[done + lane] = L.ring[(r + done + lane) & (KB64 - 1u)];
        }
        blocks_done++; good = op;
    }
    __syncthreads();
    // persist the window
    for (uint32_t i = lane * 16u; i < KB64; i += 64u * 16u) *reinterpret_cast<uint4*>(S.window + i) = *reinterpret_cast<const uint4*>(L.ring + i);
    if (lane == 0) { job.out_len = good; job.status = status; job.blocks_done = blocks_done; }
}

namespace rhk {

int lz4f_decode_launch(rh_lz4_decode_job* d_jobs, uint32_t n_jobs, rh_lz4_decode_state* d_states, hipStream_t s) {
    if (n_jobs == 0) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rh_lz4f_decode), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DecodeLds));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(rh_lz4f_decode, dim3(n_jobs), dim3(64), sizeof(DecodeLds), s, d_jobs, d_states);
    return (int)hipGetLastError();
}

size_t lz4_lds_bytes() { return sizeof(Lds); }

int lz4f_compress_launch(const rh_lz4_job* d_jobs, uint32_t n_jobs, rh_lz4_stream_state* d_states, hipStream_t s) {
    if (n_jobs == 0) return 0;
    // the attribute belongs to the current device (a process may hold several rh_device objects): set per launch, a
    // launch is one round of packets
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rh_lz4f_compress), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
    if (e != hipSuccess) return (int)e;
    static const uint32_t top_priority = (getenv("RH_LZ4_PRIORITY") && !atoi(getenv("RH_LZ4_PRIORITY"))) ? 0u : 1u;
    hipLaunchKernelGGL(rh_lz4f_compress, dim3(n_jobs), dim3(64), sizeof(Lds), s, d_jobs, d_states, top_priority);
    return (int)hipGetLastError();
}

} // namespace rhk
