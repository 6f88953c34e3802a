// Host-side launchers for the shader-independent kernels in rh_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rh_abi.h"

// Limits of the entry point that allocates nothing (rh_submit_static): its passes work in a static device array.
#define RH_STATIC_MAX_CLUSTERS (64u * 1024u)            // triangle pairs of one pass
#define RH_STATIC_MAX_PAIRS (128u * 1024u)              // (tile, cluster) pairs of one pass
#define RH_STATIC_MAX_TILES (16u * 1024u)               // 32x32 tiles of the target (3840x2160 has 8160)
#define RH_STATIC_RECORD_BYTES (16u * 1024u * 1024u)    // set-up records / extended records of one pass, each
#define RH_STATIC_SCRATCH_BYTES (40u * 1024u * 1024u)

namespace rhk {

// Device address and size of the library's static scratch array on the CURRENT device (every GPU holds its own copy at
// its own address).
hipError_t static_scratch(void** ptr, size_t* bytes);

uint32_t scan_chunks_for(uint32_t n);
uint32_t sort_blocks_for(uint32_t n);

// `general` forces the three-launch form that serves any n (short scans otherwise take one or two launches).
void exclusive_scan(uint32_t* data, uint32_t n, uint32_t* chunk_tmp, unsigned long long* total_out, hipStream_t s, bool general = false);

void clear(void* p, size_t bytes, hipStream_t s);
void bin_emit(const rh_tile_range* ranges, uint32_t n_clusters, uint32_t tiles_x, const uint32_t* block_bases,
              uint32_t* keys, uint32_t* vals, hipStream_t s);

int radix_sort_pairs(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, uint32_t key_bits,
                     uint32_t* hist, uint32_t* chunk_tmp, hipStream_t s, bool general = false);

// The same sort queued as one launch per pass, for pair arrays built by rh_bin_pairs (the digit counts are in `state`):
// status needs passes * sort_status_words(capacity) cleared words; capacity below 2^30.
uint32_t sort_parts_for(uint32_t n);
size_t sort_status_words(uint32_t capacity);
int sort_passes_chained(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t capacity, uint32_t key_bits,
                        rh_bin_state* state, uint32_t* status, unsigned long long* host_total, uint32_t resident_parts, hipStream_t s);
// How many workgroups of a sorting pass the device holds at once (launches of no more than that need no tickets: see
// rh_sort_pass); 0 when unknown.
uint32_t sort_pass_resident_parts(uint32_t compute_units);

// Fills the RH_POLY_COPIES footprint copies of an RGBA8 texture (dst: RH_POLY_COPIES * width * height * layers items of 16 bytes).
void polyphase_build(const void* src, void* dst, uint32_t width, uint32_t height, uint32_t layers, uint32_t item_bytes, hipStream_t s);

void copy_texture_to_buffer(const uint8_t* tex, uint32_t tex_pitch_bytes, uint32_t ox, uint32_t oy, uint32_t bpt, uint32_t w, uint32_t h,
                            uint8_t* dst, uint32_t dst_pitch, hipStream_t s);

} // namespace rhk
