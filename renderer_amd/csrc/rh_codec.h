// Device/host shared structures of the GPU LZ4 frame compressor (rh_codec.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RH_LZ4_MAX_WRITES 8

// Persistent per-stream compressor state (LZ4_stream_t_internal + the 64 KiB LZ4F dictionary buffer).
struct rh_lz4_stream_state {
    uint32_t hash[4096];
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

namespace rhk {
size_t lz4_lds_bytes();
int lz4f_compress_launch(const rh_lz4_job* d_jobs, uint32_t n_jobs, rh_lz4_stream_state* d_states, hipStream_t s);
}
