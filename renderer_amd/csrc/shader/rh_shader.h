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

enum : uint32_t { FORMAT_R_U8 = 0, FORMAT_BGRA_U8 = 1, FORMAT_RGBA_U8 = 2 };
enum : uint32_t { FILTER_LINEAR = 0, FILTER_NEAREST = 1 };

struct Sampler { uint32_t filter; };

struct Texture2D {
    const uint8_t* data; uint32_t width, height, layers, format;

    // Texel (x, y) of `layer`, channels on the 0..255 scale.
    RH_DEVICE float4 fetch(int x, int y, uint32_t layer) const {
        size_t idx = ((size_t)layer * height + (size_t)y) * width + (size_t)x;
        if (format == FORMAT_R_U8) return make_float4((float)data[idx], 0.0f, 0.0f, 255.0f);
        uint32_t t = reinterpret_cast<const uint32_t*>(data)[idx];
        float c0 = (float)(t & 0xffu), c1 = (float)((t >> 8) & 0xffu), c2 = (float)((t >> 16) & 0xffu), c3 = (float)(t >> 24);
        if (format == FORMAT_BGRA_U8) return make_float4(c2, c1, c0, c3);
        return make_float4(c0, c1, c2, c3);
    }

    // texture(sampler2D(t, s), uv): clamp-to-edge, one mip level, unnormalised coordinate = uv * size
    // (the only sampler the reference creates: src/texture.rs:141-158).
    RH_DEVICE float4 sample(Sampler s, float2 uv, uint32_t layer = 0) const {
        const int W = (int)width, H = (int)height;
        if (layer >= layers) layer = layers - 1;
        if (s.filter == FILTER_NEAREST) {
            float u = floorf(uv.x * (float)W), v = floorf(uv.y * (float)H);
            u = fminf(fmaxf(u, -1.0f), (float)W); v = fminf(fmaxf(v, -1.0f), (float)H);
            int i = min(max((int)u, 0), W - 1), j = min(max((int)v, 0), H - 1);
            float4 c = fetch(i, j, layer);
            return make_float4(c.x * INV_255, c.y * INV_255, c.z * INV_255, c.w * INV_255);
        }
        float u = __builtin_fmaf(uv.x, (float)W, -0.5f), v = __builtin_fmaf(uv.y, (float)H, -0.5f);
        float fu = floorf(u), fv = floorf(v);
        float a = u - fu, b = v - fv;
        fu = fminf(fmaxf(fu, -1.0f), (float)W); fv = fminf(fmaxf(fv, -1.0f), (float)H);
        int i0 = min(max((int)fu, 0), W - 1), i1 = min(max((int)fu + 1, 0), W - 1);
        int j0 = min(max((int)fv, 0), H - 1), j1 = min(max((int)fv + 1, 0), H - 1);
        float4 t00 = fetch(i0, j0, layer), t10 = fetch(i1, j0, layer), t01 = fetch(i0, j1, layer), t11 = fetch(i1, j1, layer);
        float4 r;
#define RH_LERP2(ch) { float top = __builtin_fmaf(a, t10.ch - t00.ch, t00.ch); \
                       float bot = __builtin_fmaf(a, t11.ch - t01.ch, t01.ch); \
                       r.ch = __builtin_fmaf(b, bot - top, top) * INV_255; }
        RH_LERP2(x) RH_LERP2(y) RH_LERP2(z) RH_LERP2(w)
#undef RH_LERP2
        return r;
    }
};

struct BindingContext {
    const rh_draw_params* p;
    template <typename T> RH_DEVICE const T* storage(uint32_t set, uint32_t binding) const {
        return reinterpret_cast<const T*>(p->slots[set * 4 + binding].ptr);
    }
    template <typename T> RH_DEVICE const T* uniform(uint32_t set, uint32_t binding) const {
        return reinterpret_cast<const T*>(p->slots[set * 4 + binding].ptr);
    }
    RH_DEVICE Texture2D texture2d(uint32_t set, uint32_t binding) const {
        const rh_slot_dev& s = p->slots[set * 4 + binding];
        return Texture2D{ reinterpret_cast<const uint8_t*>(s.ptr), s.width, s.height, s.layers, s.format };
    }
    RH_DEVICE Sampler sampler(uint32_t set, uint32_t binding) const {
        return Sampler{ p->slots[set * 4 + binding].filter };
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
