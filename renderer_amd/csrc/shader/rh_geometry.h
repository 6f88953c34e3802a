// Vertex fetch + viewport transform + sub-pixel snapping, shared by the binning kernels (vertex stage) and the
// tile kernel (fragment stage, which re-runs the vertex function during triangle setup instead of reading
// post-transform vertices back from HBM).
//
// Requires `Varyings` and `vertex_main` to be declared before inclusion.
#pragma once
#include "rh_shader.h"

namespace rh {

constexpr int N_VARYINGS = (int)(sizeof(Varyings) / sizeof(float));
static_assert(sizeof(Varyings) % sizeof(float) == 0, "Varyings must contain floats only");
static_assert(N_VARYINGS >= 1 && N_VARYINGS <= 8, "1..8 varying floats supported");

// Guard band: snapped coordinates are clamped to [-8192 px, 24576 px) so that every edge delta fits 24 signed bits.
constexpr float GB_LO = -8192.0f * 256.0f;
constexpr float GB_HI = 24576.0f * 256.0f - 1.0f;

struct SnappedVertex {
    int32_t X, Y;          // 24.8 fixed point, pixel centres on multiples of 256
    float rw;              // 1 / w
    bool ok;               // w > 0 and position not NaN
    Varyings var;
};

RH_DEVICE int32_t snap(float w) {
    float t = (w - 0.5f) * 256.0f;
    t = fminf(fmaxf(t, GB_LO), GB_HI);
    return (int32_t)rintf(t);
}

RH_DEVICE SnappedVertex run_vertex(const rh_draw_params& p, uint32_t vertex, uint32_t instance) {
    VertexContext c; c.p = &p; c.vertex_index = vertex; c.instance_index = instance;
    SnappedVertex r;
    float4 pos = vertex_main(c, r.var);
    float x = pos.x, y = pos.y, w = pos.w;
    r.ok = (w > 0.0f) && !(x != x) && !(y != y);
    float rw = 1.0f;
    if (w != 1.0f && r.ok) { rw = 1.0f / w; x = x * rw; y = y * rw; }
    float xw = __builtin_fmaf(x, p.vp_hw, p.vp_ox);
    float yw = __builtin_fmaf(y, -p.vp_hh, p.vp_oy);
    r.X = r.ok ? snap(xw) : 0; r.Y = r.ok ? snap(yw) : 0; r.rw = rw;
    return r;
}

// Vertex indices of triangle `i` of an instance.  Strip: (i, i+1, i+2) for even i, (i, i+2, i+1) for odd i
// (Vulkan 1.3 primitive topologies); list: (3i, 3i+1, 3i+2).  The reference selects the topology at
// src/primitive.rs:8-13 and never uses an index buffer.
RH_DEVICE void tri_vertices(uint32_t topology, uint32_t i, uint32_t idx[3]) {
    if (topology == 1u) { idx[0] = i; idx[1] = (i & 1u) ? i + 2 : i + 1; idx[2] = (i & 1u) ? i + 1 : i + 2; }
    else { idx[0] = 3 * i; idx[1] = 3 * i + 1; idx[2] = 3 * i + 2; }
}

// Inclusive pixel bounding box of a triangle intersected with the scissor; false when empty.
RH_DEVICE bool tri_pixel_bbox(const rh_draw_params& p, const SnappedVertex& a, const SnappedVertex& b, const SnappedVertex& c,
                              int& x0, int& y0, int& x1, int& y1) {
    if (!(a.ok && b.ok && c.ok)) return false;
    int32_t minX = min(a.X, min(b.X, c.X)), maxX = max(a.X, max(b.X, c.X));
    int32_t minY = min(a.Y, min(b.Y, c.Y)), maxY = max(a.Y, max(b.Y, c.Y));
    const int32_t pad = p.bbox_pad;
    x0 = max(p.sc_x0, (minX - pad + 255) >> 8); x1 = min(p.sc_x1 - 1, (maxX + pad) >> 8);
    y0 = max(p.sc_y0, (minY - pad + 255) >> 8); y1 = min(p.sc_y1 - 1, (maxY + pad) >> 8);
    return x0 <= x1 && y0 <= y1;
}

} // namespace rh
