// PNG encode/decode for the C++ host mirror: PngEncoder (reference src/png_encoder.rs:15-38: 8-bit RGBA, rows
// de-padded) and the decoder the quads example needs for its texture (reference examples/quads_main_thread.rs:166-179,
// 8-bit RGBA, no transformations).  The reference uses the `png` crate; here the container is written by hand and
// zlib does the deflate/inflate, so encoded bytes differ from the crate's while the decoded pixels are identical.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "renderer.hpp"

namespace renderer {

namespace png_detail {
inline void be32(std::vector<uint8_t>& out, uint32_t v) { for (int i = 3; i >= 0; i--) out.push_back((uint8_t)(v >> (8 * i))); }
inline void chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n) {
    be32(out, (uint32_t)n);
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4)));
}
inline uint32_t rd32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
} // namespace png_detail

struct PngEncoder {
    // Err("VideoFrame could not be written because image_data is None.") becomes the empty vector with *error set
    static std::vector<uint8_t> encode_to_bytes(const VideoFrame& frame, const char** error = nullptr) {
        std::vector<uint8_t> out;
        if (!frame.image_data) { if (error) *error = "VideoFrame could not be written because image_data is None."; return out; }
        if (error) *error = nullptr;
        const auto pixels = frame.bytes();
        return encode_rows(pixels.first, frame.width, frame.height, frame.unpadded_bytes_per_row, frame.padded_bytes_per_row);
    }
    // the frame's device buffer encoded on the GPU (rh_png_encode): only the finished PNG bytes are read back
    static std::vector<uint8_t> encode_on_device(Renderer& renderer, const VideoFrame& frame, const char** error = nullptr) {
        std::vector<uint8_t> out;
        if (!frame.image_data) { if (error) *error = "VideoFrame could not be written because image_data is None."; return out; }
        if (error) *error = nullptr;
        out.resize(rh_png_bound(frame.width, frame.height));
        size_t n = 0;
        check(rh_png_encode(renderer.device, frame.image_data, frame.width, frame.height, frame.padded_bytes_per_row, out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
    static std::vector<uint8_t> encode_rows(const uint8_t* rows, uint32_t width, uint32_t height, size_t unpadded, size_t padded) {
        using namespace png_detail;
        std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
        uint8_t ihdr[13];
        for (int i = 0; i < 4; i++) { ihdr[i] = (uint8_t)(width >> (24 - 8 * i)); ihdr[4 + i] = (uint8_t)(height >> (24 - 8 * i)); }
        ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;          // 8 bit, RGBA, deflate, adaptive filtering, no interlace
        chunk(out, "IHDR", ihdr, 13);
        std::vector<uint8_t> raw((unpadded + 1) * height);
        for (uint32_t y = 0; y < height; y++) {                                       // filter type 0 + the row without its padding
            raw[(unpadded + 1) * y] = 0;
            memcpy(&raw[(unpadded + 1) * y + 1], rows + padded * y, unpadded);
        }
        uLongf bound = compressBound((uLong)raw.size());
        std::vector<uint8_t> z(bound);
        if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), Z_BEST_SPEED) != Z_OK) throw Panic("deflate failed");
        chunk(out, "IDAT", z.data(), bound);
        chunk(out, "IEND", nullptr, 0);
        return out;
    }
};

// Decodes an 8-bit RGBA, non-interlaced PNG; ancillary chunks (gAMA, cHRM, ...) are skipped, which is what the
// reference's decoder settings do.
inline std::vector<uint8_t> decode_png_rgba8(const std::vector<uint8_t>& file, uint32_t& width, uint32_t& height) {
    using namespace png_detail;
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) throw Panic("not a PNG file");
    std::vector<uint8_t> idat;
    size_t pos = 8; bool have_header = false;
    while (pos + 12 <= file.size()) {
        const uint32_t n = rd32(&file[pos]);
        const std::string type(reinterpret_cast<const char*>(&file[pos + 4]), 4);
        if (pos + 12 + n > file.size()) throw Panic("truncated PNG chunk");
        const uint8_t* d = &file[pos + 8];
        if (type == "IHDR") {
            width = rd32(d); height = rd32(d + 4);
            if (d[8] != 8 || d[9] != 6 || d[12] != 0) throw Panic("only 8-bit RGBA non-interlaced PNGs are supported");
            have_header = true;
        } else if (type == "IDAT") idat.insert(idat.end(), d, d + n);
        else if (type == "IEND") break;
        pos += 12 + n;
    }
    if (!have_header) throw Panic("PNG without IHDR");
    const size_t stride = (size_t)width * 4;
    std::vector<uint8_t> raw((stride + 1) * height);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) throw Panic("inflate failed");
    std::vector<uint8_t> out(stride * height);
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t f = raw[(stride + 1) * y];
        const uint8_t* in = &raw[(stride + 1) * y + 1];
        uint8_t* cur = &out[stride * y];
        const uint8_t* up = y ? &out[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= 4 ? cur[i - 4] : 0, b = up ? up[i] : 0, c = (up && i >= 4) ? up[i - 4] : 0;
            int pred = 0;
            if (f == 1) pred = a; else if (f == 2) pred = b; else if (f == 3) pred = (a + b) / 2;
            else if (f == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (f != 0) throw Panic("bad PNG filter type");
            cur[i] = (uint8_t)(in[i] + pred);
        }
    }
    return out;
}

} // namespace renderer
