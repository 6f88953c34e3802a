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

// Clip volume.  A vertex is *inside* when w >= W_MIN, 0 <= z <= w (Vulkan depth clipping: the reference leaves
// `unclipped_depth` unset, src/pipeline.rs:197-207) and its window position lies in the clip band [BAND_LO, BAND_HI]
// pixels on both axes (inside the guard band, so snapping does not move it).  A triangle whose three vertices are inside
// is rasterised as it is; a triangle with a NaN coordinate or with all three vertices outside one plane is dropped; any
// other triangle is clipped against the seven planes first (clip_triangle) and the resulting convex polygon is drawn as a
// fan.  Clipping lives in kernels of its own (rh_bin_count_clip, rh_bin_pairs_clip, rh_setup_clip*): their polygons need
// private memory, which the kernels of the usual case do without; those report a triangle that needs clipping and the
// host runs the pass again with the clipping kernels (and keeps them for the pipeline).
constexpr float W_MIN = 1.0f / 1048576.0f;
constexpr float BAND_LO = -8000.0f, BAND_HI = 24000.0f;

// Which clip planes a vertex is outside of (0: inside the clip volume).  The band bits are only looked at for vertices
// in front of the eye (w >= W_MIN), whose window position exists.
enum : uint32_t { OUT_W = 1u, OUT_NEAR = 2u, OUT_FAR = 4u, OUT_LEFT = 8u, OUT_RIGHT = 16u, OUT_TOP = 32u, OUT_BOTTOM = 64u, OUT_NAN = 128u };

struct SnappedVertex {
    int32_t X, Y;          // 24.8 fixed point, pixel centres on multiples of 256
    float rw;              // 1 / w
    bool ok;               // inside the clip volume: out == 0
    uint32_t out;          // OUT_* bits
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
    const bool nan = (x != x) || (y != y) || (pos.z != pos.z) || (w != w);
    const bool front = !nan && w >= W_MIN;
    float rw = 1.0f;
    if (w != 1.0f && front) { rw = 1.0f / w; x = x * rw; y = y * rw; }
    float xw = __builtin_fmaf(x, p.vp_hw, p.vp_ox);
    float yw = __builtin_fmaf(y, -p.vp_hh, p.vp_oy);
    uint32_t out = nan ? OUT_NAN : (w >= W_MIN ? 0u : OUT_W);
    if (!nan) out |= (pos.z < 0.0f ? OUT_NEAR : 0u) | (pos.z > w ? OUT_FAR : 0u);
    if (front) out |= (xw < BAND_LO ? OUT_LEFT : 0u) | (xw > BAND_HI ? OUT_RIGHT : 0u) | (yw < BAND_LO ? OUT_TOP : 0u) | (yw > BAND_HI ? OUT_BOTTOM : 0u);
    r.out = out; r.ok = out == 0u;
    r.X = r.ok ? snap(xw) : 0; r.Y = r.ok ? snap(yw) : 0; r.rw = rw;
    return r;
}

// What becomes of a triangle: drawn as it is (every vertex inside the clip volume), dropped (a NaN coordinate, or all
// three vertices outside the same plane: nothing of it is inside), or clipped.
enum { TRI_WHOLE = 0, TRI_DROPPED = 1, TRI_CLIPPED = 2 };
RH_DEVICE int classify(const SnappedVertex& a, const SnappedVertex& b, const SnappedVertex& c) {
    const uint32_t any = a.out | b.out | c.out;
    if (any == 0u) return TRI_WHOLE;
    if ((any & OUT_NAN) || (a.out & b.out & c.out)) return TRI_DROPPED;
    return TRI_CLIPPED;
}

// ---- clipping (the rare path: kept out of line, its polygons live in private memory) -------------------------------------
struct ClipVertex { float c[4]; float var[N_VARYINGS]; };       // clip-space x, y, z, w and the varyings
constexpr int CLIP_PLANES = 7, CLIP_MAX_VERTS = 3 + CLIP_PLANES;
struct ClipPolygon { int n; SnappedVertex v[CLIP_MAX_VERTS]; };

// Signed distance of `v` to clip plane `plane` (inside: not negative).  Plane 0 keeps w positive for the others.
RH_DEVICE float clip_distance(const rh_draw_params& p, const ClipVertex& v, int plane) {
    const float x = v.c[0], y = v.c[1], z = v.c[2], w = v.c[3];
    switch (plane) {
        case 0: return w - W_MIN;
        case 1: return z;
        case 2: return w - z;
        case 3: return __builtin_fmaf(x, p.vp_hw, (p.vp_ox - BAND_LO) * w);        // x_w >= BAND_LO
        case 4: return __builtin_fmaf(-x, p.vp_hw, (BAND_HI - p.vp_ox) * w);       // x_w <= BAND_HI
        case 5: return __builtin_fmaf(-y, p.vp_hh, (p.vp_oy - BAND_LO) * w);       // y_w = -y/w * hh + oy >= BAND_LO
        default: return __builtin_fmaf(y, p.vp_hh, (BAND_HI - p.vp_oy) * w);       // y_w <= BAND_HI
    }
}

// Clips triangle (vertex indices idx[0..2] of `instance`) against the clip volume (Sutherland-Hodgman, planes in the
// order of clip_distance) and projects the polygon's vertices.  A new vertex is always computed from the inside end of
// its edge towards the outside end, t = d_in / (d_in - d_out), every coordinate and varying as fma(t, out - in, in): two
// triangles that share a clipped edge get the same point.  Vertices that were inside pass through unchanged and
// project exactly as run_vertex projects them.  out.n = 0: nothing left (or a NaN coordinate: the triangle is dropped).
__device__ __noinline__ void clip_triangle(const rh_draw_params& p, uint32_t instance, uint32_t i0, uint32_t i1, uint32_t i2, ClipPolygon& out) {
    ClipVertex a[CLIP_MAX_VERTS], b[CLIP_MAX_VERTS];
    out.n = 0;
    const uint32_t idx[3] = { i0, i1, i2 };
    for (int k = 0; k < 3; k++) {
        VertexContext c; c.p = &p; c.vertex_index = idx[k]; c.instance_index = instance;
        Varyings var;
        const float4 pos = vertex_main(c, var);
        if (pos.x != pos.x || pos.y != pos.y || pos.z != pos.z || pos.w != pos.w) return;
        a[k].c[0] = pos.x; a[k].c[1] = pos.y; a[k].c[2] = pos.z; a[k].c[3] = pos.w;
        const float* vf = reinterpret_cast<const float*>(&var);
        for (int v = 0; v < N_VARYINGS; v++) a[k].var[v] = vf[v];
    }
    ClipVertex* src = a; ClipVertex* dst = b;
    int n = 3;
    for (int plane = 0; plane < CLIP_PLANES && n >= 3; plane++) {
        int m = 0;
        for (int i = 0; i < n; i++) {
            const ClipVertex& cur = src[i];
            const ClipVertex& nxt = src[i + 1 == n ? 0 : i + 1];
            const float dc = clip_distance(p, cur, plane), dn = clip_distance(p, nxt, plane);
            const bool cin = !(dc < 0.0f), nin = !(dn < 0.0f);
            // (a convex polygon gains at most one vertex per plane; infinite coordinates can make the distances NaN and the
            // inside tests inconsistent: vertices beyond the room of the arrays are left out)
            if (cin && m < CLIP_MAX_VERTS) dst[m++] = cur;
            if (cin != nin && m < CLIP_MAX_VERTS) {
                const ClipVertex& in = cin ? cur : nxt;
                const ClipVertex& ou = cin ? nxt : cur;
                const float di = cin ? dc : dn, dou = cin ? dn : dc;
                const float t = di / (di - dou);
                ClipVertex nv;
                for (int k = 0; k < 4; k++) nv.c[k] = __builtin_fmaf(t, ou.c[k] - in.c[k], in.c[k]);
                for (int v = 0; v < N_VARYINGS; v++) nv.var[v] = __builtin_fmaf(t, ou.var[v] - in.var[v], in.var[v]);
                dst[m++] = nv;
            }
        }
        ClipVertex* t = src; src = dst; dst = t;
        n = m;
    }
    if (n < 3) return;
    for (int i = 0; i < n; i++) {
        const float w = fmaxf(src[i].c[3], W_MIN), rw = 1.0f / w;
        const float x = src[i].c[0] * rw, y = src[i].c[1] * rw;
        SnappedVertex& r = out.v[i];
        r.X = snap(__builtin_fmaf(x, p.vp_hw, p.vp_ox)); r.Y = snap(__builtin_fmaf(y, -p.vp_hh, p.vp_oy)); r.rw = rw; r.ok = true; r.out = 0u;
        float* vf = reinterpret_cast<float*>(&r.var);
        for (int v = 0; v < N_VARYINGS; v++) vf[v] = src[i].var[v];
    }
    out.n = n;
}

// A cluster some of whose triangles are clipped makes one (tile, cluster) pair per tile and *sub-record*: pair value =
// cluster | sub << CLIP_SUB_SHIFT; sub s covers triangles 2*(s mod 4) and 2*(s mod 4) + 1 of the fan of the cluster's
// first (s < 4) or second triangle (a triangle that needs no clipping is a fan of one).  Only the sub-records whose fan
// triangles exist get pairs (rh_tile_range carries their numbers).
constexpr uint32_t CLIP_SUB_SHIFT = 29u;
static_assert((CLIP_MAX_VERTS - 2 + 1) / 2 == 4, "four records of two fan triangles per clipped triangle");
// Bit 28 of a pair's value: the cluster is NOT a lean quad (below): its set-up record is written by the set-up launch and
// read back by the tile kernel.  Clear: the tile kernel sets the quad up itself, in LDS, when it meets the pair, and no
// record of it ever goes through memory.  Cluster indices stay below 2^28.
constexpr uint32_t PAIR_GENERAL = 1u << 28, CLIP_CLUSTER_MASK = PAIR_GENERAL - 1u;

// Edge coefficients below this magnitude (512 px) make f32 edge evaluation exact (edges_inside_f32 in the fragment stage).
constexpr int32_t F32_EDGE_LIMIT = 1 << 17;

// A *lean quad*: the two triangles (v0, v1, v2), (v1, v3, v2) of a 4-vertex strip whose snapped corners form an
// axis-aligned rectangle split along v1-v2, every vertex inside the clip volume with w = 1 (affine varyings), sides below
// 512 px, in a single-sampled pass -- what the reference's quads are.  Its pixel set is the rectangle's exactly and the
// shared diagonal alone says which triangle a pixel belongs to.  `shape` (out): the rectangle test alone.
RH_DEVICE bool lean_quad(const rh_draw_params& p, const SnappedVertex& v0, const SnappedVertex& v1, const SnappedVertex& v2, const SnappedVertex& v3, bool& shape) {
    const bool columns = v0.X == v1.X && v2.X == v3.X && v0.Y == v2.Y && v1.Y == v3.Y;
    const bool rows = v0.Y == v1.Y && v2.Y == v3.Y && v0.X == v2.X && v1.X == v3.X;
    const int32_t rminX = min(v0.X, v3.X), rmaxX = max(v0.X, v3.X), rminY = min(v0.Y, v3.Y), rmaxY = max(v0.Y, v3.Y);
    shape = p.samples == 1u && (columns || rows) && v0.X != v3.X && v0.Y != v3.Y;
    return shape && p.no_lean == 0u && v0.ok && v1.ok && v2.ok && v3.ok && rmaxX - rminX < F32_EDGE_LIMIT && rmaxY - rminY < F32_EDGE_LIMIT &&
           v0.rw == 1.0f && v1.rw == 1.0f && v2.rw == 1.0f && v3.rw == 1.0f;
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
