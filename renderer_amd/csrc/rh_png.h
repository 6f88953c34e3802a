// Device/host shared arguments of the GPU PNG encoder (rh_png.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct rh_png_args {
    const uint8_t* image;             // RGBA8 rows, padded_bytes_per_row apart (device memory)
    uint32_t width, height;
    uint32_t padded_bytes_per_row;
    uint32_t up_ok;                   // 1 when a row (4 * width + 1 bytes) fits the 32 KiB deflate window
    uint32_t up_code, up_code_bits;   // distance code + extra bits (LSB first) of the distance "one row"
    uint32_t slot_dwords;             // per-row staging slot size
    uint32_t* slots;                  // [height * slot_dwords]
    uint32_t* row_bytes;              // [height] deflate bytes of each row
    uint32_t* row_s1;                 // [height] Adler partial sums
    uint32_t* row_s2;
    unsigned long long* row_offset;   // [height] offset of each row's IDAT chunk in `out`
    uint8_t* out;                     // the IDAT chunks, back to back, then the 16-byte Adler chunk
    unsigned long long* total_bytes;  // bytes written to `out`
};

namespace rhk {
void png_encode_launch(const rh_png_args& a, hipStream_t s);
}
