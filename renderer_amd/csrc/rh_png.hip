// GPU PNG encoder for captured RGBA8 frames (reference src/png_encoder.rs:15-38: 8-bit RGBA, rows de-padded): the
// frame never leaves device memory uncompressed, only the finished IDAT byte stream is read back.
//
// The reference hands the rows to the `png` crate, whose deflate output is not pinned by anything; what is pinned is the
// decoded image.  The stream produced here is a valid zlib stream made for parallel hardware:
//   * every image row is one deflate block (fixed Huffman codes) written by one wavefront and closed with an empty
//     stored block, which byte-aligns it, so rows are encoded independently and simply concatenated;
//   * every row is its own IDAT chunk (consecutive IDAT chunks concatenate), so rows also carry independent CRCs;
//   * matches are the two that rendered frames are full of: "same as the pixel to the left" (distance 4) and "same as
//     the pixel above" (distance = one row), found per pixel and merged into runs of up to 64 pixels; other pixels are
//     literals.  Row filter type 0 throughout.
//   * Adler-32 is summed per row and combined; it travels in a last 4-byte IDAT chunk.
// Three launches: encode rows -> scan sizes / combine Adler (one workgroup) -> frame the chunks (length, type, CRC) at
// their final offsets.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rh_png.h"

namespace {

constexpr uint32_t CRC_POLY = 0xedb88320u;
constexpr uint32_t ADLER_MOD = 65521u;

__device__ __forceinline__ uint32_t brev_bits(uint32_t code, int nbits) { return __brev(code) >> (32 - nbits); }

// Fixed Huffman code of a literal byte, bit-reversed for LSB-first packing: 0..143 -> 8 bits, 144..255 -> 9 bits.
__device__ __forceinline__ void literal_code(uint32_t v, uint32_t& bits, int& n) {
    if (v < 144u) { bits = brev_bits(0x30u + v, 8); n = 8; }
    else { bits = brev_bits(0x190u + (v - 144u), 9); n = 9; }
}

// Length symbol + extra bits of a match of `len` bytes (3..257), fixed Huffman, LSB-first.
__device__ __forceinline__ void length_code(uint32_t len, uint32_t& bits, int& n) {
    const uint32_t l = len - 3u;
    uint32_t idx, e;
    if (l < 8u) { idx = l; e = 0; }
    else { e = (31u - (uint32_t)__clz((int)l)) - 2u; idx = 4u * e + 4u + ((l >> e) & 3u); }
    const uint32_t sym = 257u + idx;
    uint32_t code; int nc;
    if (sym < 280u) { code = brev_bits(sym - 256u, 7); nc = 7; }
    else { code = brev_bits(0xc0u + (sym - 280u), 8); nc = 8; }
    bits = code | ((l & ((1u << e) - 1u)) << nc);
    n = nc + (int)e;
}

struct BitWriter {
    uint32_t* stage;          // LDS, 96 dwords
    uint32_t* out;            // global, dword aligned
    uint32_t out_dw;          // dwords written
    uint32_t carry_bits;      // valid bits in stage[0] carried over from the previous flush
};

// Every lane contributes `n` bits (`bits`, LSB first; n <= 40) in lane order; whole dwords go to global memory.
__device__ __forceinline__ void emit(BitWriter& w, uint64_t bits, int n, int lane) {
    // exclusive prefix sum of n across the wave
    uint32_t inc = (uint32_t)n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    const uint32_t total = __shfl(inc, 63);
    const uint32_t pos = w.carry_bits + inc - (uint32_t)n;
    if (n > 0) {
        const uint32_t d = pos >> 5, sh = pos & 31u;
        atomicOr(&w.stage[d], (uint32_t)(bits << sh));
        const uint64_t hi = sh ? (bits >> (32u - sh)) : (bits >> 32);       // the bits beyond the first dword
        if ((uint32_t)hi) atomicOr(&w.stage[d + 1], (uint32_t)hi);
        if (hi >> 32) atomicOr(&w.stage[d + 2], (uint32_t)(hi >> 32));
    }
    __syncthreads();
    const uint32_t all = w.carry_bits + total, ndw = all >> 5;
    for (uint32_t i = (uint32_t)lane; i < ndw; i += 64u) w.out[w.out_dw + i] = w.stage[i];
    const uint32_t carry = w.stage[ndw];
    __syncthreads();
    for (uint32_t i = (uint32_t)lane; i <= ndw + 2u && i < 96u; i += 64u) w.stage[i] = 0;
    __syncthreads();
    if (lane == 0) w.stage[0] = carry;
    __syncthreads();
    w.out_dw += ndw; w.carry_bits = all & 31u;
}

__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b) {      // a * b mod P, reflected CRC-32 polynomial
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1u)) == 0u) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}

// x^(n * 2^k) mod P
__device__ uint32_t x2nmodp(uint64_t n, uint32_t k) {
    uint32_t table = 1u << 30;                 // x^1
    for (uint32_t i = 0; i < k; i++) table = multmodp(table, table);       // x^(2^k)
    uint32_t p = 1u << 31;                     // x^0
    while (n) {
        if (n & 1u) p = multmodp(table, p);
        n >>= 1;
        table = multmodp(table, table);
    }
    return p;
}

// crc of A||B from crc(A), crc(B) and |B| in bytes
__device__ __forceinline__ uint32_t crc_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    return multmodp(x2nmodp(len_b, 3), crc_a) ^ crc_b;
}

} // namespace

// ---- 1: one wavefront per row -----------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(64) void rh_png_rows(rh_png_args a) {
    __shared__ uint32_t stage[96];
    const int lane = (int)threadIdx.x;
    const uint32_t y = blockIdx.x;
    const uint32_t* row = reinterpret_cast<const uint32_t*>(a.image + (size_t)y * a.padded_bytes_per_row);
    const uint32_t* above = y ? reinterpret_cast<const uint32_t*>(a.image + (size_t)(y - 1) * a.padded_bytes_per_row) : nullptr;
    for (int i = lane; i < 96; i += 64) stage[i] = 0;
    __syncthreads();
    BitWriter w; w.stage = stage; w.out = a.slots + (size_t)y * a.slot_dwords; w.out_dw = 0; w.carry_bits = 0;
    const uint32_t stride = a.width * 4u + 1u;            // bytes of one filtered row: filter byte + pixels
    const bool use_up = above != nullptr && a.up_ok;

    {   // row prologue: zlib header (first row), block header BFINAL=0 BTYPE=01, the filter-type byte as a literal 0
        uint64_t bits = 0; int n = 0;
        if (lane == 0) {
            if (y == 0) { bits = 0x0178u; n = 16; }                       // 0x78 0x01: deflate, 32 KiB window, no dictionary
            bits |= (uint64_t)2u << n; n += 3;
            uint32_t lc; int ln; literal_code(0u, lc, ln);
            bits |= (uint64_t)lc << n; n += ln;
        }
        emit(w, bits, n, lane);
    }
    uint32_t s1 = 0, s2 = 0;                               // Adler partial sums of this lane's bytes
    for (uint32_t base = 0; base < a.width; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        const bool active = i < a.width;
        uint32_t px = 0; bool up = false, left = false;
        if (active) {
            px = row[i];
            if (use_up) up = px == above[i];
            if (i > 0) left = px == row[i - 1];
        }
        const int cls = !active ? 3 : (up ? 1 : (left ? 2 : 0));         // 0 literal, 1 copy from above, 2 copy from the left
        const int prev_cls = __shfl_up(cls, 1);
        const bool start = active && (lane == 0 || cls == 0 || cls != prev_cls);
        const unsigned long long starts = __ballot(start), act = __ballot(active);
        uint64_t bits = 0; int n = 0;
        if (start) {
            if (cls == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) { uint32_t lc; int ln; literal_code((px >> (8 * k)) & 0xffu, lc, ln); bits |= (uint64_t)lc << n; n += ln; }
            } else {
                const unsigned long long later = starts & ~((2ull << lane) - 1ull);
                const int end = later ? __builtin_ctzll(later) : __popcll(act);
                const uint32_t len = 4u * (uint32_t)(end - lane);
                uint32_t lc; int ln; length_code(len, lc, ln);
                bits = lc; n = ln;
                if (cls == 2) { bits |= (uint64_t)brev_bits(3u, 5) << n; n += 5; }                      // distance 4: code 3, no extra bits
                else { bits |= (uint64_t)a.up_code << n; n += (int)a.up_code_bits; }                      // distance = one row
            }
        }
        emit(w, bits, n, lane);
        if (active) {
            const uint32_t b0 = px & 0xffu, b1 = (px >> 8) & 0xffu, b2 = (px >> 16) & 0xffu, b3 = px >> 24;
            const uint32_t wgt = stride - (1u + 4u * i);   // weight of b0: bytes from it to the end of the row, itself included
            s1 += b0 + b1 + b2 + b3;
            s2 = (s2 + wgt * b0 + (wgt - 1u) * b1 + (wgt - 2u) * b2 + (wgt - 3u) * b3) % ADLER_MOD;
        }
    }
    {   // row epilogue: end-of-block (7 zero bits), empty stored block (BFINAL on the last row), pad to a byte, 00 00 ff ff
        uint64_t bits = 0; int n = 0;
        if (lane == 0) {
            n = 7;
            bits |= (uint64_t)(y + 1 == a.height ? 1u : 0u) << n; n += 3;
            n += (int)((8u - ((w.carry_bits + (uint32_t)n) & 7u)) & 7u);
        }
        emit(w, bits, n, lane);
        emit(w, lane == 0 ? 0xffff0000ull : 0ull, lane == 0 ? 32 : 0, lane);
    }
    // the tail (carry_bits is a multiple of 8 now)
    if (lane == 0 && w.carry_bits) w.out[w.out_dw] = stage[0];
    // Adler partials of the row
    s1 %= ADLER_MOD;
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
    if (lane == 0) {
        a.row_bytes[y] = w.out_dw * 4u + w.carry_bits / 8u;
        a.row_s1[y] = s1 % ADLER_MOD;
        a.row_s2[y] = s2 % ADLER_MOD;
    }
}

// ---- 2: one workgroup: chunk offsets, Adler-32, the trailing IDAT chunk ---------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void rh_png_scan(rh_png_args a) {
    __shared__ uint32_t nb[256], s1[256], s2[256];
    __shared__ unsigned long long offs[256];
    uint64_t off = 0;
    uint32_t A = 1, B = 0;
    const uint32_t stride = a.width * 4u + 1u;
    for (uint32_t base = 0; base < a.height; base += 256u) {
        const uint32_t y = base + threadIdx.x;
        if (y < a.height) { nb[threadIdx.x] = a.row_bytes[y]; s1[threadIdx.x] = a.row_s1[y]; s2[threadIdx.x] = a.row_s2[y]; }
        __syncthreads();
        if (threadIdx.x == 0) {
            // serial over the rows of this batch (a few thousand additions per frame)
            const uint32_t cnt = min(256u, a.height - base);
            for (uint32_t k = 0; k < cnt; k++) {
                offs[k] = off;
                off += 12u + nb[k];
                B = (uint32_t)((B + (uint64_t)(stride % ADLER_MOD) * A + s2[k]) % ADLER_MOD);
                A = (A + s1[k]) % ADLER_MOD;
            }
        }
        __syncthreads();
        if (y < a.height) a.row_offset[y] = offs[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const uint32_t adler = (B << 16) | A;
    uint8_t* t = a.out + off;
    const uint8_t chunk[12] = { 0, 0, 0, 4, 'I', 'D', 'A', 'T', (uint8_t)(adler >> 24), (uint8_t)(adler >> 16), (uint8_t)(adler >> 8), (uint8_t)adler };
    uint32_t crc = 0xffffffffu;
    for (int i = 0; i < 12; i++) {
        t[i] = chunk[i];
        if (i >= 4) { crc ^= chunk[i]; for (int k = 0; k < 8; k++) crc = (crc & 1u) ? (crc >> 1) ^ CRC_POLY : crc >> 1; }
    }
    crc ^= 0xffffffffu;
    t[12] = (uint8_t)(crc >> 24); t[13] = (uint8_t)(crc >> 16); t[14] = (uint8_t)(crc >> 8); t[15] = (uint8_t)crc;
    *a.total_bytes = off + 16u;
}

// ---- 3: one workgroup per row: [length]["IDAT"][data][crc] at the final offset --------------------------------------------
extern "C" __global__ __launch_bounds__(256) void rh_png_frame(rh_png_args a) {
    __shared__ uint32_t table[256];
    __shared__ uint32_t part_crc[256];
    __shared__ uint32_t part_len[256];
    const uint32_t y = blockIdx.x, tid = threadIdx.x;
    {   uint32_t c = tid;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
        table[tid] = c;
    }
    const uint32_t n = a.row_bytes[y];
    const uint8_t* src = reinterpret_cast<const uint8_t*>(a.slots + (size_t)y * a.slot_dwords);
    uint8_t* dst = a.out + a.row_offset[y];
    if (tid < 4) dst[tid] = (uint8_t)(n >> (24 - 8 * tid));
    if (tid == 0) { dst[4] = 'I'; dst[5] = 'D'; dst[6] = 'A'; dst[7] = 'T'; }
    for (uint32_t i = tid; i < n; i += 256u) dst[8 + i] = src[i];
    __syncthreads();
    // CRC over "IDAT" + data: contiguous slices, then a combining tree
    const uint32_t total = n + 4u, slice = (total + 255u) / 256u;
    const uint32_t begin = min(tid * slice, total), end = min(begin + slice, total);
    uint32_t crc = 0xffffffffu;
    for (uint32_t i = begin; i < end; i++) {
        const uint32_t byte = i < 4u ? (uint32_t)("IDAT"[i]) : src[i - 4u];
        crc = table[(crc ^ byte) & 0xffu] ^ (crc >> 8);
    }
    part_crc[tid] = crc ^ 0xffffffffu; part_len[tid] = end - begin;          // crc of an empty slice is 0
    __syncthreads();
    for (uint32_t step = 1; step < 256u; step <<= 1) {
        if ((tid & (2u * step - 1u)) == 0u) {
            const uint32_t lb = part_len[tid + step];
            if (lb) {
                part_crc[tid] = part_len[tid] ? crc_combine(part_crc[tid], part_crc[tid + step], lb) : part_crc[tid + step];
                part_len[tid] += lb;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t c = part_crc[0];
        uint8_t* p = dst + 8 + n;
        p[0] = (uint8_t)(c >> 24); p[1] = (uint8_t)(c >> 16); p[2] = (uint8_t)(c >> 8); p[3] = (uint8_t)c;
    }
}

namespace rhk {

void png_encode_launch(const rh_png_args& a, hipStream_t s) {
    hipLaunchKernelGGL(rh_png_rows, dim3(a.height), dim3(64), 0, s, a);
    hipLaunchKernelGGL(rh_png_scan, dim3(1), dim3(256), 0, s, a);
    hipLaunchKernelGGL(rh_png_frame, dim3(a.height), dim3(256), 0, s, a);
}

} // namespace rhk
