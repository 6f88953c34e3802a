// GPU side of the capture codec: LZ4 frame (LZ4F) streams compressed on the device, one stream per workgroup, so that
// only finished compressed bytes are read back over PCIe.
//
// The reference compresses captured frames on CPU worker threads, one LZ4F stream per thread/file
// (reference src/compressor.rs:88-144, via the lzzzz crate = liblz4): preferences level 0, auto-flush, block-linked,
// 64 KiB blocks, no checksums; every Write::write_all is one LZ4F_compressUpdate.  A stream's bytes depend only on its
// own write sequence, so streams are independent -- that is the parallelism used here: each stream's state (hash
// table, 64 KiB dictionary, offsets) lives in device memory and a workgroup replays the sequential LZ4 "fast"
// algorithm for its stream with the current 64 KiB block, the 64 KiB of history before it and the hash table resident
// in LDS.  The wave's lanes cooperate on the data-parallel parts (block loads, match-length counting, literal copies,
// dictionary saves); the match search itself is inherently serial and runs wave-uniformly.
//
// The output is byte-identical to liblz4 1.9.x for the same write sequence (LZ4_compress_fast_continue with
// acceleration 1, byU32 table, 64-bit hash5; LZ4_saveDict after every update; LZ4_renormDictT).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rh_codec.h"

namespace {

constexpr uint32_t KB64 = 65536u;
constexpr int MINMATCH = 4, LASTLITERALS = 5, MFLIMIT = 12, LZ4_MINLENGTH = MFLIMIT + 1;
constexpr int HASHLOG = 12, SKIP_TRIGGER = 6;
constexpr uint32_t DISTANCE_MAX = 65535u, ML_MASK = 15u, RUN_MASK = 15u;
constexpr uint32_t BLOCK_UNCOMPRESSED_FLAG = 0x80000000u;
constexpr int CUR = KB64;                     // byte offset of the current block inside `buf`

// LDS: [ 64 KiB history | 64 KiB current block + 16 B slack ] followed by the hash table.  History byte at distance d
// before the block is buf[CUR - d], so a match index maps to one flat offset whether it lies in the dictionary, in the
// previous blocks of the same write, or in the current block.
struct Lds {
    uint8_t buf[2 * KB64 + 16];
    uint32_t hash[1 << HASHLOG];
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint32_t rd32(const uint8_t* buf, uint32_t off) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(buf);
    const uint32_t i = off >> 2;
    return __builtin_amdgcn_alignbyte(w[i + 1], w[i], off & 3u);
}
__device__ __forceinline__ uint64_t rd64(const uint8_t* buf, uint32_t off) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(buf);
    const uint32_t i = off >> 2, s = off & 3u;
    const uint32_t a = w[i], b = w[i + 1], c = w[i + 2];
    return (uint64_t)__builtin_amdgcn_alignbyte(b, a, s) | ((uint64_t)__builtin_amdgcn_alignbyte(c, b, s) << 32);
}
__device__ __forceinline__ uint32_t hash5(const uint8_t* buf, uint32_t off) {
    const uint64_t prime5bytes = 889523592379ULL;
    return (uint32_t)(((rd64(buf, off) << 24) * prime5bytes) >> (64 - HASHLOG));
}

// Number of equal bytes of buf[ip..] and buf[match..], stopping at `limit` (an offset bounding ip).  256 bytes per step.
__device__ __forceinline__ uint32_t count_match(const uint8_t* buf, uint32_t ip, uint32_t match, uint32_t limit, int lane) {
    uint32_t n = 0;
    for (;;) {
        const uint32_t p = ip + n + (uint32_t)lane * 4u;
        uint32_t avail = p < limit ? limit - p : 0u; if (avail > 4u) avail = 4u;
        uint32_t same = 0;
        if (avail) {
            const uint32_t d = rd32(buf, p) ^ rd32(buf, match + n + (uint32_t)lane * 4u);
            same = d ? (uint32_t)__builtin_ctz(d) >> 3 : 4u;
            if (same > avail) same = avail;
        }
        const unsigned long long stop = __ballot(same < 4u);
        if (stop) {
            const int first = __builtin_ctzll(stop);
            return n + (uint32_t)first * 4u + (uint32_t)__builtin_amdgcn_readlane((int)same, first);
        }
        n += 256u;
    }
}

// Cooperative copy of n bytes from LDS to global memory.
__device__ __forceinline__ void copy_out(uint8_t* dst, const uint8_t* src, uint32_t n, int lane) {
    for (uint32_t i = (uint32_t)lane; i < n; i += 64u) dst[i] = src[i];
}

struct Stream {
    uint32_t current_offset;
    uint32_t dict_size;       // bytes of valid history held at the end of the history half of `buf`
};

// LZ4_compress_generic (limitedOutput, byU32) on the block cur = buf[CUR .. CUR + n).  `ext_dict` selects the
// usingExtDict rules (first block of a write; history = saved dictionary) against withPrefix64k (later blocks of the
// same write).  Returns the compressed size, or 0 when it does not fit in max_out bytes.
__device__ uint32_t compress_block(Lds& L, Stream& st, uint32_t n, uint8_t* out, uint32_t max_out, bool ext_dict, bool dict_small, int lane) {
    const uint8_t* buf = L.buf;
    const uint32_t start_index = st.current_offset;
    const uint32_t dict_size = st.dict_size;
    // index -> flat offset: off = index - start_index + CUR
    const uint32_t bias = (uint32_t)CUR - start_index;
    const uint32_t prefix_idx_limit = start_index - dict_size;
    // Flat offset of the first reachable history byte.  Within one write the prefix can be longer than the 64 KiB held
    // here, but a match never reaches further back than 65535 bytes before the block (offsets are at most 65535 and
    // the catch-up loop stops at the block's first byte), so clamping does not change any decision.
    const uint32_t dict_start = (uint32_t)CUR - (dict_size < KB64 ? dict_size : KB64);
    const uint32_t iend = (uint32_t)CUR + n;
    const uint32_t mflimit_plus_one = iend - MFLIMIT + 1;
    const uint32_t matchlimit = iend - LASTLITERALS;
    uint32_t ip = CUR, anchor = CUR;
    uint32_t op = 0;                                                        // bytes written to `out`
    uint32_t low_limit = ext_dict ? (uint32_t)CUR : dict_start;
    uint32_t match = 0, token_pos = 0, token = 0, forward_h;

    st.dict_size += n;
    st.current_offset += n;

    if ((int)n < LZ4_MINLENGTH) goto last_literals;

    if (lane == 0) L.hash[hash5(buf, ip)] = ip - bias;
    ip++; forward_h = hash5(buf, ip);

    for (;;) {
        {   uint32_t forward_ip = ip;
            uint32_t step = 1, search_match_nb = 1u << SKIP_TRIGGER;
            for (;;) {
                const uint32_t h = forward_h;
                const uint32_t current = forward_ip - bias;
                const uint32_t match_index = uni(L.hash[h]);
                ip = forward_ip;
                forward_ip += step;
                step = (search_match_nb++ >> SKIP_TRIGGER);
                if (forward_ip > mflimit_plus_one) goto last_literals;
                match = match_index + bias;
                if (ext_dict) low_limit = match_index < start_index ? dict_start : (uint32_t)CUR;
                forward_h = hash5(buf, forward_ip);
                if (lane == 0) L.hash[h] = current;
                if (dict_small && match_index < prefix_idx_limit) continue;
                if (match_index + DISTANCE_MAX < current) continue;
                if (uni(rd32(buf, match)) == uni(rd32(buf, ip))) break;
            }
        }
        // catch up
        while (ip > anchor && match > low_limit && uni(buf[ip - 1]) == uni(buf[match - 1])) { ip--; match--; }
        {   const uint32_t lit = ip - anchor;
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
            copy_out(out + op, buf + anchor, lit, lane);
            op += lit;
        }
next_match:
        {   const uint32_t offset = ip - match;
            if (lane == 0) { out[op] = (uint8_t)offset; out[op + 1] = (uint8_t)(offset >> 8); }
            op += 2;
        }
        {   // match length: buf is flat, so a match that starts in the history simply runs on into the current block
            uint32_t match_code = count_match(buf, ip + MINMATCH, match + MINMATCH, matchlimit, lane);
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
        if (lane == 0) L.hash[hash5(buf, ip - 2)] = ip - 2 - bias;
        {   const uint32_t h = hash5(buf, ip);
            const uint32_t current = ip - bias;
            const uint32_t match_index = uni(L.hash[h]);
            match = match_index + bias;
            if (ext_dict) low_limit = match_index < start_index ? dict_start : (uint32_t)CUR;
            if (lane == 0) L.hash[h] = current;
            if ((dict_small ? match_index >= prefix_idx_limit : true) && match_index + DISTANCE_MAX >= current &&
                uni(rd32(buf, match)) == uni(rd32(buf, ip))) {
                token_pos = op++; token = 0;
                goto next_match;
            }
        }
        forward_h = hash5(buf, ++ip);
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
        copy_out(out + op, buf + anchor, last_run, lane);
        op += last_run;
    }
    return op;
}

} // namespace

// One workgroup (one wave) per job.  A job is a sequence of writes to one stream; every write is one
// LZ4F_compressUpdate: full 64 KiB blocks, then the remainder, then the dictionary save.
extern "C" __global__ __launch_bounds__(64) void rh_lz4f_compress(const rh_lz4_job* jobs, rh_lz4_stream_state* states) {
    extern __shared__ uint8_t lds_raw[];
    Lds& L = *reinterpret_cast<Lds*>(lds_raw);
    const int lane = (int)threadIdx.x;
    const rh_lz4_job job = jobs[blockIdx.x];
    rh_lz4_stream_state& S = states[job.stream];

    // bring the stream state into LDS
    for (int i = lane; i < (1 << HASHLOG); i += 64) L.hash[i] = S.hash[i];
    Stream st; st.current_offset = S.current_offset; st.dict_size = S.dict_size;
    {   const uint32_t d = st.dict_size;              // <= 64 KiB, stored at the end of S.dict
        for (uint32_t i = (uint32_t)lane * 16u; i < d; i += 64u * 16u) {
            const uint32_t off = KB64 - ((d + 15u) & ~15u) + i;
            *reinterpret_cast<uint4*>(L.buf + off) = *reinterpret_cast<const uint4*>(S.dict + off);
        }
    }
    __syncthreads();

    uint8_t* out = job.out;
    uint64_t op = 0;
    for (uint32_t wi = 0; wi < job.n_writes; wi++) {
        const uint8_t* src = job.write_ptr[wi];
        const uint64_t size = job.write_len[wi];
        if (size == 0) continue;
        bool first_block = true;
        for (uint64_t done = 0; done < size; ) {
            const uint32_t n = (uint32_t)((size - done) < KB64 ? (size - done) : KB64);
            // LZ4_renormDictT
            if (st.current_offset + n > 0x80000000u) {
                const uint32_t delta = st.current_offset - KB64;
                for (int i = lane; i < (1 << HASHLOG); i += 64) { const uint32_t h = L.hash[i]; L.hash[i] = h < delta ? 0u : h - delta; }
                st.current_offset = KB64;
                if (st.dict_size > KB64) st.dict_size = KB64;
            }
            if (first_block) {
                // a dictionary shorter than 4 bytes is dropped (LZ4_compress_fast_continue)
                if (st.dict_size - 1u < 3u) st.dict_size = 0;
            }
            // load the block behind the history (byte loop handles unaligned tails; sources are 4-byte aligned)
            {   const uint8_t* s = src + done;
                if ((reinterpret_cast<uintptr_t>(s) & 15u) == 0) {
                    for (uint32_t i = (uint32_t)lane * 16u; i < n; i += 64u * 16u)
                        *reinterpret_cast<uint4*>(L.buf + CUR + i) = (i + 16u <= n) ? *reinterpret_cast<const uint4*>(s + i) : make_uint4(0, 0, 0, 0);
                    const uint32_t tail = n & ~15u;
                    for (uint32_t i = tail + (uint32_t)lane; i < n; i += 64u) L.buf[CUR + i] = s[i];
                } else {
                    for (uint32_t i = (uint32_t)lane; i < n; i += 64u) L.buf[CUR + i] = s[i];
                }
            }
            __syncthreads();
            const bool small = st.dict_size < KB64 && st.dict_size < st.current_offset;
            const bool ext = first_block;
            uint8_t* blk = out + op;
            Stream before = st;
            uint32_t c = compress_block(L, st, n, blk + 4, n - 1, ext, small, lane);
            if (c == 0) { st = before; st.dict_size += n; st.current_offset += n; }      // state advances even when the block is stored
            if (ext) st.dict_size = n;                      // external-dictionary mode: the block becomes the new dictionary
            __syncthreads();
            if (c == 0) {
                c = n;
                if (lane == 0) { const uint32_t h = n | BLOCK_UNCOMPRESSED_FLAG; blk[0] = (uint8_t)h; blk[1] = (uint8_t)(h >> 8); blk[2] = (uint8_t)(h >> 16); blk[3] = (uint8_t)(h >> 24); }
                copy_out(blk + 4, L.buf + CUR, n, lane);
            } else if (lane == 0) { blk[0] = (uint8_t)c; blk[1] = (uint8_t)(c >> 8); blk[2] = (uint8_t)(c >> 16); blk[3] = (uint8_t)(c >> 24); }
            op += 4u + c;
            done += n;
            first_block = false;
            // the block just compressed becomes history: slide the last 64 KiB of [history | block] down
            {   const uint32_t keep = st.dict_size < KB64 ? st.dict_size : KB64;      // == LZ4_saveDict size after the last block
                // move buf[CUR + n - keep .. CUR + n) to buf[CUR - keep .. CUR), ascending, 1 KiB per step, reads before writes
                const uint32_t src_off = CUR + n - keep, dst_off = CUR - keep;
                if (n > 0 && src_off != dst_off) {
                    for (uint32_t i = 0; i < keep; i += 64u * 16u) {
                        const uint32_t j = i + (uint32_t)lane * 16u;
                        uint8_t tmp[16];
                        const bool act = j < keep;
                        if (act) for (int k = 0; k < 16; k++) tmp[k] = (j + k < keep) ? L.buf[src_off + j + k] : 0;
                        __syncthreads();
                        if (act) for (int k = 0; k < 16; k++) if (j + k < keep) L.buf[dst_off + j + k] = tmp[k];
                        __syncthreads();
                    }
                }
                if (done >= size) st.dict_size = keep;      // LZ4_saveDict at the end of the update
            }
            __syncthreads();
        }
    }

    // persist the stream state
    __syncthreads();
    for (int i = lane; i < (1 << HASHLOG); i += 64) S.hash[i] = L.hash[i];
    {   const uint32_t d = st.dict_size;
        for (uint32_t i = (uint32_t)lane * 16u; i < ((d + 15u) & ~15u); i += 64u * 16u) {
            const uint32_t off = KB64 - ((d + 15u) & ~15u) + i;
            *reinterpret_cast<uint4*>(S.dict + off) = *reinterpret_cast<const uint4*>(L.buf + off);
        }
    }
    if (lane == 0) { S.current_offset = st.current_offset; S.dict_size = st.dict_size; *job.out_len = op; }
}

namespace rhk {

size_t lz4_lds_bytes() { return sizeof(Lds); }

int lz4f_compress_launch(const rh_lz4_job* d_jobs, uint32_t n_jobs, rh_lz4_stream_state* d_states, hipStream_t s) {
    if (n_jobs == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rh_lz4f_compress), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(rh_lz4f_compress, dim3(n_jobs), dim3(64), sizeof(Lds), s, d_jobs, d_states);
    return (int)hipGetLastError();
}

} // namespace rhk
