// Shader-author interface: what a `*.vert.hip` / `*.frag.hip` source sees.
//
// A vertex source defines
//     struct Varyings { ... floats only ... };
//     RH_DEVICE float4 vertex_main(const rh::VertexContext& c, Varyings& v);
// A fragment source defines
//     RH_DEVICE void fragment_main(const rh::FragmentContext& c, const Varyings& v, rh::Outputs& out);
// They replace the GLSL `main`s the reference loads as SPIR-V (reference src/program.rs:43-48;
// examples/quads/hello.vert, hello.frag).  Bindings are addressed by (set, binding) exactly as in the GLSL
// `layout(set=, binding=)` qualifiers; attributes by `layout(location=)`.
//
// All arithmetic here is written as explicit single IEEE-754 binary32 operations (compile with
// -ffp-contract=off) so results are reproducible bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include "../rh_abi.h"

#define RH_DEVICE __device__ __forceinline__

namespace rh {

constexpr float INV_255 = 1.0f / 255.0f;

enum : uint32_t { FORMAT_R_U8 = 0, FORMAT_BGRA_U8 = 1, FORMAT_RGBA_U8 = 2, FORMAT_RGBA_F16 = 3, FORMAT_RGBA_F32 = 4 };
enum : uint32_t { FILTER_LINEAR = 0, FILTER_NEAREST = 1 };

// `static_*` members are compile-time constants in the specialised kernel bodies (0 = decide at run time): after
// inlining, the branches on format and filter fold away.
struct Sampler { uint32_t filter; uint32_t static_filter; };

typedef float v2f __attribute__((ext_vector_type(2)));                       // maps to packed f32 VALU instructions
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));   // two adjacent texels, 4-byte aligned

RH_DEVICE v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// The form of footprint copies the bilinear body reads (rh_slot_dev::poly): 32 = f16 pairs read by v_fma_mix_f32, 16 = the
// texels as bytes, 48 = both, picked per primitive (PolyCopy::pick).  Identical results; the code object tells the host
// which to build (rh_module_info_t).
#ifndef RH_POLY_ITEM_BYTES
#define RH_POLY_ITEM_BYTES 48
#endif
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// fma(a, hi half, lo half) of a dword holding two f16: one v_fma_mix_f32
RH_DEVICE float lerp_h2(float a, uint32_t w) { const h2 h = __builtin_bit_cast(h2, w); return __builtin_fmaf(a, (float)h.y, (float)h.x); }

// Byte k of a dword as f32 (exact).  Written as the instruction itself: left to the compiler, a difference of two
// converted bytes becomes an integer subtraction plus a conversion, two instructions outside the f32 add/mul/fma
// class.  A SIMD issues one instruction of that class beside one of any other kind (from two different waves), so the
// count outside the class is what bounds the shading pass; the values are the same either way.
RH_DEVICE float ubyte0(uint32_t v) { float r; asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(v)); return r; }
RH_DEVICE float ubyte1(uint32_t v) { float r; asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(v)); return r; }
RH_DEVICE float ubyte2(uint32_t v) { float r; asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(v)); return r; }
RH_DEVICE float ubyte3(uint32_t v) { float r; asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(v)); return r; }
// clamp(x, lo, hi) in one instruction; a NaN x gives lo (as fminf(fmaxf(x, lo), hi) does)
RH_DEVICE float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

struct Texture2D {
    const uint8_t* data; uint32_t width, height, layers, format;
    uint32_t static_rgba8;      // 1: known to be RGBA8, at least 2x2 and at most 16 MiB (texel offsets exact in f32)
    // Footprint copies regrouped by phase (rh_slot_dev::poly), when the kernel body was built for them: the copy picked for
    // the primitive being shaded (PolyCopy, prepared once per primitive by the tile kernel), else null.
    const uint8_t* poly_base; float poly_invS, poly_C1, poly_C2, poly_C3; bool poly_wide;
    uint32_t static_poly;       // 1: the kernel body was built for textures that all have footprint copies: poly_base is set

    // Texel (x, y) of `layer` (generic path, any supported format): 8-bit channels on the 0..255 scale, float channels
    // as they are (a half converts to float exactly).
    RH_DEVICE float4 fetch(int x, int y, uint32_t layer) const {
        const uint32_t format = static_rgba8 ? (uint32_t)FORMAT_RGBA_U8 : this->format;
        const uint32_t idx = (layer * height + (uint32_t)y) * width + (uint32_t)x;
        if (format == FORMAT_R_U8) return make_float4((float)data[idx], 0.0f, 0.0f, 255.0f);
        if (format == FORMAT_RGBA_F32) return *reinterpret_cast<const float4*>(data + ((size_t)idx << 4));
        if (format == FORMAT_RGBA_F16) {
            const _Float16* h = reinterpret_cast<const _Float16*>(data + ((size_t)idx << 3));
            return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        }
        const uint32_t t = *reinterpret_cast<const uint32_t*>(data + (idx << 2));
        const float c0 = (float)(t & 0xffu), c1 = (float)((t >> 8) & 0xffu), c2 = (float)((t >> 16) & 0xffu), c3 = (float)(t >> 24);
        if (format == FORMAT_BGRA_U8) return make_float4(c2, c1, c0, c3);
        return make_float4(c0, c1, c2, c3);
    }

    // texture(sampler2D(t, s), uv): clamp-to-edge, one mip level, unnormalised coordinate = uv * size
    // (the only sampler the reference creates: src/texture.rs:141-158).  Textures are limited to 4 GiB so that texel
    // byte offsets fit 32 bits.
    //
    // Linear filtering in the weight-clamp form: taps ib and min(ib + 1, W - 1) with ib = clamp(floor(u), 0, W - 2)
    // and weight clamp(u - ib, 0, 1), u = s*W - 0.5; beyond the edge texel centres the weight saturates and the result
    // is the edge texel exactly.  Rows are combined with lerp(t, b, w) = fma(w, b, fma(-w, t, t)).
    RH_DEVICE float4 sample(Sampler s, float2 uv, uint32_t layer = 0) const {
        const int W = (int)width, H = (int)height;
        const uint32_t format = static_rgba8 ? (uint32_t)FORMAT_RGBA_U8 : this->format;
        const uint32_t filter = s.static_filter ? s.static_filter - 1u : s.filter;
        if (layer >= layers) layer = layers - 1;
        // 8-bit texels are filtered on the 0..255 scale and normalised once at the end; float texels are not scaled
        const float scale = (format == FORMAT_RGBA_F16 || format == FORMAT_RGBA_F32) ? 1.0f : INV_255;
        if (filter == FILTER_NEAREST) {
            float u = floorf(uv.x * (float)W), v = floorf(uv.y * (float)H);
            u = fminf(fmaxf(u, -1.0f), (float)W); v = fminf(fmaxf(v, -1.0f), (float)H);
            int i = min(max((int)u, 0), W - 1), j = min(max((int)v, 0), H - 1);
            float4 c = fetch(i, j, layer);
            return make_float4(c.x * scale, c.y * scale, c.z * scale, c.w * scale);
        }
        const float u = __builtin_fmaf(uv.x, (float)W, -0.5f), v = __builtin_fmaf(uv.y, (float)H, -0.5f);
        const float ibf = floorf(clampf(u, 0.0f, (float)(W >= 2 ? W - 2 : 0)));
        const float jbf = floorf(clampf(v, 0.0f, (float)(H >= 2 ? H - 2 : 0)));
        const float a = clampf(u - ibf, 0.0f, 1.0f), b = clampf(v - jbf, 0.0f, 1.0f);
        float4 r;
        if (static_rgba8 || ((format == FORMAT_RGBA_U8 || format == FORMAT_BGRA_U8) && W >= 2 && H >= 2)) {
            // 4-byte texels: the taps of a row are adjacent and the rows are one pitch apart: two 8-byte loads
            uint32_t t00, t10, t01, t11;
#if defined(RH_EXP) && RH_EXP == 1      /* diagnostic build: no texture read, no filter */
            if (static_rgba8 && static_poly) return make_float4(a, b, a * b, 0.8f);
#endif
            if (static_rgba8 && static_poly) {
                // Item x = xq * S + ph of row j sits at item ph * (W / S) + xq of the row in copy log2 S, i.e. at byte
                //   B * (j * W + (x - xq * S) * (W / S) + xq) = x * (B * W / S) + xq * (B - B * W) + j * (B * W)      (B bytes per item)
                // -- three products of small integers, every partial sum a multiple of B below 2^24 * B: exact in f32.
                // (poly_C1 = B W / S, poly_C2 = B W, poly_C3 = B - B W: uniform, from PolyCopy::pick)
                const float xqf = floorf(ibf * poly_invS);
                const uint32_t off = (uint32_t)__builtin_fmaf(ibf, poly_C1, __builtin_fmaf(xqf, poly_C3, jbf * poly_C2));
                if (poly_wide) {
                    // 32-byte items: channels 0, 1 and 2, 3 as { left | right - left } f16 pairs of the upper, of the lower row
                    const uint4* const item = reinterpret_cast<const uint4*>(poly_base + (size_t)layer * (height * (width << 5)) + off);
                    const uint4 f0 = item[0], f1 = item[1];
#define RH_LERP_H2(ch, UP, LO) { const float top = lerp_h2(a, UP), bot = lerp_h2(a, LO); \
                                 r.ch = __builtin_fmaf(b, bot, __builtin_fmaf(-b, top, top)) * scale; }
                    RH_LERP_H2(x, f0.x, f0.y) RH_LERP_H2(y, f0.z, f0.w) RH_LERP_H2(z, f1.x, f1.y) RH_LERP_H2(w, f1.z, f1.w)
#undef RH_LERP_H2
                    return r;
                }
                // 16-byte items: the four taps as they are
                const uint4 f = *reinterpret_cast<const uint4*>(poly_base + (size_t)layer * (height * (width << 4)) + off);
                t00 = f.x; t10 = f.y; t01 = f.z; t11 = f.w;
#if defined(RH_EXP) && RH_EXP == 2      /* diagnostic build: the read, but no filter */
                return make_float4(__uint_as_float(t00 ^ t10) * a, __uint_as_float(t01 ^ t11) * b, 0.5f, 0.8f);
#endif
            } else {
                uint32_t off;
                if (static_rgba8) off = layer * (height * (width << 2)) + (uint32_t)__builtin_fmaf(jbf, (float)(W << 2), ibf * 4.0f);   // exact: below 2^24
                else off = (__umul24(layer * height + (uint32_t)(int)jbf, width) + (uint32_t)(int)ibf) << 2;
                const u32x2_a4 r0 = *reinterpret_cast<const u32x2_a4*>(data + off);
                const u32x2_a4 r1 = *reinterpret_cast<const u32x2_a4*>(data + (width << 2) + off);
                t00 = r0.x; t10 = r0.y; t01 = r1.x; t11 = r1.y;
            }
#define RH_LERP_BYTE(ch, UB) { const float p00 = UB(t00), p10 = UB(t10), p01 = UB(t01), p11 = UB(t11); \
                               const float top = __builtin_fmaf(a, p10 - p00, p00); \
                               const float bot = __builtin_fmaf(a, p11 - p01, p01); \
                               r.ch = __builtin_fmaf(b, bot, __builtin_fmaf(-b, top, top)) * scale; }
            RH_LERP_BYTE(x, ubyte0) RH_LERP_BYTE(y, ubyte1) RH_LERP_BYTE(z, ubyte2) RH_LERP_BYTE(w, ubyte3)
#undef RH_LERP_BYTE
        } else {
            const int ib = (int)ibf, jb = (int)jbf;
            const int i1 = min(ib + 1, W - 1), j1 = min(jb + 1, H - 1);
            float4 t00 = fetch(ib, jb, layer), t10 = fetch(i1, jb, layer), t01 = fetch(ib, j1, layer), t11 = fetch(i1, j1, layer);
            if (format == FORMAT_BGRA_U8) {       // fetch() already swizzled: undo so that the common swizzle below applies
                float x;
                x = t00.x; t00.x = t00.z; t00.z = x; x = t10.x; t10.x = t10.z; t10.z = x;
                x = t01.x; t01.x = t01.z; t01.z = x; x = t11.x; t11.x = t11.z; t11.z = x;
            }
#define RH_LERP2(ch) { const float top = __builtin_fmaf(a, t10.ch - t00.ch, t00.ch); \
                       const float bot = __builtin_fmaf(a, t11.ch - t01.ch, t01.ch); \
                       r.ch = __builtin_fmaf(b, bot, __builtin_fmaf(-b, top, top)) * scale; }
            RH_LERP2(x) RH_LERP2(y) RH_LERP2(z) RH_LERP2(w)
#undef RH_LERP2
        }
        if (format == FORMAT_BGRA_U8) return make_float4(r.z, r.y, r.x, r.w);
        return r;
    }
};

// The footprint copy of one texture slot picked for a primitive: copy k holds the items of every row regrouped by their
// phase modulo S = 2^k, with S about the distance in texels between the samples of neighbouring fragments.  Any k gives
// the same texels; the choice only decides how many cache lines a wave touches.
struct PolyCopy {
    // invS = 2^-k for copy k; with B the bytes of an item: C1 = B * width * 2^-k, C2 = B * width, C3 = B - B * width (see
    // Texture2D::sample); wide: the copy holds the 32-byte items
    const uint8_t* base = nullptr; float invS = 1.0f, C1 = 0.0f, C2 = 0.0f, C3 = 0.0f; bool wide = RH_POLY_ITEM_BYTES != 16;
    // grad_log2 = floor(log2(1.41 * |d coordinate / dx| per pixel)).  Everything is uniform and is kept on the scalar unit (readfirstlane: the clamp would otherwise be
    // done, and its results kept, in vector registers); the power of two divides by exponent arithmetic (width >= 32,
    // k <= 5: nowhere near a denormal).
    // Which items: the 32-byte ones cost two loads per fragment and pay where the lanes of a wave read neighbouring
    // items -- a magnified or barely minified texture (copy 0), or one minified by a power of two, whose copy puts the
    // texels that far apart side by side.  At any other ratio neighbouring fragments' items are scattered whatever the
    // copy, every load is a cache line of its own, and the 16-byte items' one load is what counts (a 256-texel texture
    // over 24-px quads: twice as fast).
    // (`wide_items`: the caller's choice for the primitive, wide_for() below, one for all its texture slots)
    RH_DEVICE void pick(const rh_slot_dev& slot, int grad_log2, bool wide_items) {
        int k = grad_log2 + (31 - __builtin_clz(slot.width | 1u));
        k = k < 0 ? 0 : (k > RH_POLY_COPIES - 1 ? RH_POLY_COPIES - 1 : k);
        k = __builtin_amdgcn_readfirstlane(k);
        const bool both = RH_POLY_ITEM_BYTES == 48;
        wide = both ? wide_items : RH_POLY_ITEM_BYTES == 32;
        // slot.size_bytes: bytes of one copy (of the 32-byte form when both are there: those come first, the 16-byte ones behind them)
        const uint32_t narrow_size = both ? slot.size_bytes >> 1 : slot.size_bytes;
        const size_t at = wide ? (size_t)((uint32_t)k * slot.size_bytes)
                               : (size_t)(both ? (uint32_t)RH_POLY_COPIES * slot.size_bytes : 0u) + (size_t)((uint32_t)k * narrow_size);
        base = reinterpret_cast<const uint8_t*>(slot.poly) + at;
        invS = __uint_as_float((uint32_t)(127 - k) << 23);
        const uint32_t B = wide ? 32u : 16u;
        const float bw = (float)(slot.width * B);
        const uint32_t bw_bits = __builtin_amdgcn_readfirstlane(__float_as_uint(bw));
        C1 = __uint_as_float(bw_bits - ((uint32_t)k << 23));        // (a slot with copies has a width)
        C2 = __uint_as_float(bw_bits);
        C3 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)B - bw)));
    }
    // The choice for a primitive whose gradient hint is (grad_log2, near), going by a texture 2^log2_width texels wide.
    static RH_DEVICE bool wide_for(int grad_log2, bool near, uint32_t log2_width, bool width_is_pow2) {
        if (RH_POLY_ITEM_BYTES != 48) return RH_POLY_ITEM_BYTES == 32;
        return grad_log2 + (int)log2_width <= 0 || (near && width_is_pow2);
    }
};

struct BindingContext {
    const rh_draw_params* p;
    uint32_t static_rgba8 = 0, static_filter = 0, static_poly = 0;
    uint32_t static_wide = 0;            // (static_poly) 1: every slot's copies are the 32-byte ones, 2: the 16-byte ones, 0: PolyCopy::wide says
    PolyCopy poly[RH_DEV_MAX_SLOTS];     // (static_poly) per texture slot; slots a shader does not sample cost nothing
    // A read-only storage buffer as an array of T.  An element that does not lie inside the bound buffer reads as
    // zeros (what wgpu's robust buffer access guarantees at worst), so a draw with more instances than the buffer
    // holds cannot touch memory outside it.
    template <typename T> struct StorageView {
        const T* data; uint32_t count;
        // (an unconditional load from a clamped index, then a select: no branch around the load; every allocation of the
        // library has spare bytes behind it, so element 0 of an empty buffer can be read)
        RH_DEVICE T operator[](uint32_t i) const { T zero{}; const T v = data[i < count ? i : (count ? count - 1u : 0u)]; return i < count ? v : zero; }
    };
    template <typename T> RH_DEVICE StorageView<T> storage(uint32_t set, uint32_t binding) const {
        const rh_slot_dev& s = p->slots[set * 4 + binding];
        return StorageView<T>{ reinterpret_cast<const T*>(s.ptr), s.size_bytes / (uint32_t)sizeof(T) };
    }
    template <typename T> RH_DEVICE const T* uniform(uint32_t set, uint32_t binding) const {
        return reinterpret_cast<const T*>(p->slots[set * 4 + binding].ptr);
    }
    RH_DEVICE Texture2D texture2d(uint32_t set, uint32_t binding) const {
        const rh_slot_dev& s = p->slots[set * 4 + binding];
        const PolyCopy& c = poly[set * 4 + binding];
        return Texture2D{ reinterpret_cast<const uint8_t*>(s.ptr), s.width, s.height, s.layers, s.format, static_rgba8,
                          c.base, c.invS, c.C1, c.C2, c.C3, static_wide ? static_wide == 1u : c.wide, static_poly };
    }
    RH_DEVICE Sampler sampler(uint32_t set, uint32_t binding) const {
        return Sampler{ p->slots[set * 4 + binding].filter, static_filter };
    }
};

struct VertexContext : BindingContext {
    uint32_t vertex_index, instance_index;
    RH_DEVICE float attribute1(uint32_t location) const { return p->attr[location][(size_t)vertex_index * p->attr_size[location]]; }
    RH_DEVICE float2 attribute2(uint32_t location) const {
        const float* a = p->attr[location] + (size_t)vertex_index * p->attr_size[location];
        return make_float2(a[0], a[1]);
    }
    RH_DEVICE float3 attribute3(uint32_t location) const {
        const float* a = p->attr[location] + (size_t)vertex_index * p->attr_size[location];
        return make_float3(a[0], a[1], a[2]);
    }
    RH_DEVICE float4 attribute4(uint32_t location) const {
        const float* a = p->attr[location] + (size_t)vertex_index * p->attr_size[location];
        return make_float4(a[0], a[1], a[2], a[3]);
    }
};

struct FragmentContext : BindingContext {};

struct Outputs { float4 color[RH_DEV_MAX_TARGETS]; };

RH_DEVICE constexpr uint32_t fnv1a(const char* s) {
    uint32_t h = 2166136261u;
    while (*s) { h ^= (uint32_t)(unsigned char)*s++; h *= 16777619u; }
    return h;
}

} // namespace rh
