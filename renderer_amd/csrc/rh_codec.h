// Device/host shared structures of the GPU LZ4 frame compressor (rh_codec.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RH_LZ4_MAX_WRITES 8

// Persistent per-stream compressor state (LZ4_stream_t_internal + the 64 KiB LZ4F dictionary buffer).
struct rh_lz4_stream_state {
    uint32_t hash[4096];        // per entry: the position's index (17 bits; the table is rebased before every block) and, above it,
                                // 15 bits derived from the four bytes at the position (0: never written)
    uint8_t dict[65536];        // the last dict_size bytes are valid, ending at dict + 65536
    uint32_t current_offset;
    uint32_t dict_size;
    uint32_t pad[2];
};

// One kernel workgroup: a sequence of writes (each = one LZ4F_compressUpdate) appended to one stream.
struct rh_lz4_job {
    uint32_t stream;
    uint32_t n_writes;
    const uint8_t* write_ptr[RH_LZ4_MAX_WRITES];    // device addresses, 16-byte aligned
    uint64_t write_len[RH_LZ4_MAX_WRITES];
    uint8_t* out;                                   // device address where this job's blocks go
    unsigned long long* out_len;                    // bytes produced
};

// ---- decoding ---------------------------------------------------------------------------------------------------------
// Persistent per-stream decoder state: the last 64 KiB of decoded bytes (block-linked frames: a match may reach that far
// back, across blocks and launches), as the ring the kernel keeps them in (byte at stream position p at index p mod 64 KiB).
struct rh_lz4_decode_state {
    uint8_t window[65536];
};

#define RH_LZ4_DECODE_OK 0u
#define RH_LZ4_DECODE_MALFORMED 1u          /* a sequence runs past its block, or the block past 64 KiB of output */
#define RH_LZ4_DECODE_BAD_OFFSET 2u         /* a match reaches before the first decoded byte or further than 65535 back */
#define RH_LZ4_DECODE_NO_ROOM 3u            /* more output than the job has room for */

// One kernel workgroup: `n_blocks` consecutive LZ4F data blocks of one stream, headers included, starting at `src`
// (16-byte aligned, readable up to the next multiple of 4096 bytes past the last block), decoded to `out`.
struct rh_lz4_decode_job {
    uint32_t stream;
    uint32_t n_blocks;
    const uint8_t* src;
    uint32_t src_len;               // bytes of the blocks
    uint32_t block_trailer;         // bytes behind every block that are skipped (4 with block checksums)
    uint8_t* out;                   // room for out_cap bytes; its address is congruent to decoded_before modulo 16
    uint64_t out_cap;
    uint64_t decoded_before;        // bytes of this stream decoded by earlier jobs (position of out[0] in the stream)
    // results
    uint64_t out_len;               // bytes decoded, up to the end of the last block decoded completely
    uint32_t status;                // RH_LZ4_DECODE_*
    uint32_t blocks_done;           // blocks decoded completely
};

namespace rhk {
int lz4f_decode_launch(rh_lz4_decode_job* d_jobs, uint32_t n_jobs, rh_lz4_decode_state* d_states, hipStream_t s);
size_t lz4_lds_bytes();
int lz4f_compress_launch(const rh_lz4_job* d_jobs, uint32_t n_jobs, rh_lz4_stream_state* d_states, hipStream_t s);
}
