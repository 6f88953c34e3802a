// Fragment-stage code object: compiled once per (vertex source, fragment source) pair with
//     hipcc --genco --offload-arch=gfx950 -ffp-contract=off -DRH_VERTEX_SOURCE=... -DRH_FRAGMENT_SOURCE=... rh_stage_fragment.hip
// Exports
//   rh_setup (rh_setup_ms2/4/8 for multisampled passes): triangle set-up, one thread per (tile, cluster) pair of the
//     sorted pair array.  A cluster is two consecutive triangles of one instance, i.e. one quad of a 4-vertex strip.
//     The thread runs the vertex function for the cluster's vertices, snaps them, and writes the pair's record relative
//     to the pair's 32x32 tile: clipped bounding box, flags, plane equations of the varyings and -- for clusters that
//     are not axis-aligned rectangles -- the edge functions with the top-left rule folded into the constant.
//   rh_tile_<A> (A = 1..4 colour attachments; rh_tile_w<A> when an attachment is RGBA16F/RGBA32F; rh_tile_ms<S>):
//     ONE WAVEFRONT per 32x32 tile (a workgroup of 64 threads).  Per tile:
//       1. the tile's pixels are brought into LDS (cleared, or loaded with coalesced row reads), one u32 RGBA8 per pixel;
//       2. the tile's records are walked in submission order, RH_ROUND_HITS at a time through a small LDS staging area.
//          For every record the pixels of its box are packed linearly onto the lanes, 64 per pass; a lane finds the
//          triangle that covers its pixel, evaluates the plane equations, runs the fragment function and blends into
//          its LDS pixel (read, blend, re-quantise to UNORM8, write).  A tile belongs to one wave and a wave's LDS
//          operations complete in order, so every pixel sees its fragments in primitive order without any barrier;
//       3. the tile is written back with coalesced row stores (RGBA or BGRA).
//     Nothing in a tile waits for another wave: there is no workgroup barrier in the kernel, every lane of the set-up
//     launch has a pair of its own, and a quad is cut into pieces only where it crosses a tile border.
//     The last workgroups of a launch may each take a band of a tile's rows (a half, a third, a quarter or an
//     eighth of the tile) instead of a whole tile (rh_tile_args::split2_first): the same records in the same order, every box cut
//     to the band, only the band's rows brought in and written back -- a pixel still belongs to exactly one wave.
//
// The kernel body exists in several specialisations (template parameters BLEND and TEX, selected per launch by
// rh_tile_args::variant): with the blend factors and the texture format / filter known at compile time the per-fragment
// branches on pipeline state fold away; variant 0 reads every state at run time.
//
// Cost model this is written against (measured on MI355X, tools/probe_valu4.hip, tools/probe_pass.hip): a wave issues
// one vector instruction per ~4.3 cycles; a SIMD overlaps instructions of two different waves when at least one of them
// is an f32 add/mul/fma, at ~2.4 cycles per instruction overall, while instructions of every other kind (integer,
// logic, conversions, min/max/med3, compares) go one at a time at ~4.2 cycles.  Hence per-fragment work is kept in the
// f32 add/mul/fma class where that is exact, and everything else is counted.
#include "rh_shader.h"
#include RH_VERTEX_SOURCE
#include RH_FRAGMENT_SOURCE
#include "rh_geometry.h"

#ifndef RH_FRAGMENT_OUTPUTS
#error "the fragment source must define RH_FRAGMENT_OUTPUTS (number of colour outputs it writes)"
#endif

namespace rh {

constexpr int NV = N_VARYINGS;
constexpr int TILE = RH_TILE_W, TILE_PX = RH_TILE_W * RH_TILE_H;
constexpr int ROUND = RH_ROUND_HITS;
static_assert(RH_TILE_W == 32 && RH_TILE_H == 32 && RH_TILE_THREADS == 64, "one wavefront per 32x32 tile; 8-bit tile-relative boxes");
static_assert(ROUND == 8 || ROUND == 16 || ROUND == 32, "a round is at most one hit per lane, and two lanes set a pair up");

// ---- record layout (dwords) -------------------------------------------------------------------------------------------
// Set-up record, staged through LDS by the tile wave:
//   [0] header: box x0 | y0 << 5 | x1 << 10 | y1 << 15 (tile-relative, inclusive, 0..31 each; x0 > x1: contributes
//       nothing) | flags << 20 | hint << 26.  hint = near | g << 1 with g = floor(log2(1.41 * largest |d varying / dx| per
//       pixel)) over the cluster's planes clamped to -16..15: how far apart neighbouring fragments sample a texture, in
//       units of its size; near: that distance is within half a per cent of a power of two
//   [1..3] shared diagonal of a rectangle (edge 1 of triangle A): A, B, T
//   then per triangle TRI_WORDS: XR, YR (reference vertex in 24.8 fixed point relative to the tile origin, as f32, exact),
//   v0[NV], dvdx[NV], dvdy[NV] (gradients per 1/256 pixel)
// Extended record (global memory, read only by the clusters that need it), per triangle EXT_TRI_WORDS:
//   A[3], B[3], T[3]: edge e is inside <=> A*lx + B*ly + T >= 0 at tile-relative pixel (lx, ly), as f32 or (FLAG_BIG) i32
//   q0, qx, qy: the 1/w plane of a perspective triangle
// followed in a multisampled pass by RH_MS_EXTRA_WORDS(S) dwords: MsRecord<S>.
constexpr int HDR_WORDS = 4;
constexpr int TRI_WORDS = (2 + 3 * NV + 3) & ~3;
constexpr int REC_WORDS = HDR_WORDS + 2 * TRI_WORDS;
constexpr int EXT_TRI_WORDS = 12;
constexpr int EXT_WORDS = 2 * EXT_TRI_WORDS;
constexpr int OFF_XR = 0, OFF_YR = 1, OFF_V0 = 2, OFF_DVDX = 2 + NV, OFF_DVDY = 2 + 2 * NV;
constexpr int EOFF_A = 0, EOFF_B = 3, EOFF_T = 6, EOFF_Q = 9;
static_assert(REC_WORDS % 4 == 0 && EXT_WORDS % 4 == 0, "records move as 16-byte pieces");

constexpr uint32_t BBOX_EMPTY = 0x000000ffu;    // x0 = 255 > x1 = 0 (the kernels work with boxes as x0 | y0 << 8 | x1 << 16 | y1 << 24)
constexpr uint32_t HDR_EMPTY = 31u;             // header of a record that contributes nothing: x0 = 31 > x1 = 0, no flags
// `gmax`: the largest |d varying / dx| of the cluster, per 1/256 pixel.
RH_DEVICE uint32_t hdr_pack(uint32_t box, uint32_t flags, float gmax) {
    if (box == BBOX_EMPTY) return HDR_EMPTY;
    int g = (int)((__float_as_uint(gmax * (256.0f * 1.41421356f)) >> 23) & 0xffu) - 127;
    g = g < -16 ? -16 : (g > 15 ? 15 : g);
    // (a power of two times 1 +- 0.005: the fraction bits of the distance per pixel are nearly all zeros or nearly all ones)
    const uint32_t frac = __float_as_uint(gmax * 256.0f) & 0x7fffffu;
    const uint32_t near = (frac < 41943u || frac > 8304721u) ? 1u : 0u;
    return (box & 31u) | (((box >> 8) & 31u) << 5) | (((box >> 16) & 31u) << 10) | (((box >> 24) & 31u) << 15) | (flags << 20) | (near << 26) | ((uint32_t)g << 27);
}
RH_DEVICE uint32_t hdr_box(uint32_t h) { return (h & 31u) | (((h >> 5) & 31u) << 8) | (((h >> 10) & 31u) << 16) | (((h >> 15) & 31u) << 24); }
RH_DEVICE uint32_t hdr_flags(uint32_t h) { return (h >> 20) & 63u; }
RH_DEVICE int hdr_hint(uint32_t h) { return (int)h >> 26; }          // the six bits as one signed number (compared, and taken apart by PolyCopy::pick)
constexpr uint32_t FLAG_PERSP_A = 1u, FLAG_PERSP_B = 2u, FLAG_HAS_A = 4u, FLAG_HAS_B = 8u, FLAG_BIG = 16u, FLAG_RECT = 32u;
} // namespace rh

extern "C" __device__ const rh_module_info_t rh_module_info = {
    RH_MODULE_MAGIC, RH_DEVICE_ABI_VERSION, RH_MODULE_STAGE_FRAGMENT, (uint32_t)rh::N_VARYINGS, (uint32_t)RH_FRAGMENT_OUTPUTS,
    rh::fnv1a(RH_VERTEX_TAG), (uint32_t)(rh::REC_WORDS * 4), (uint32_t)(rh::EXT_WORDS * 4), (uint32_t)RH_POLY_ITEM_BYTES
};

// Diagnostic builds (-DRH_PHASE_TIMING, tools/phase_timing.sh): a few sample tiles print how many cycles their wave
// spent [0] loading the tile, [1] bringing records into LDS, [2] shading, [3] storing the tile.
#ifdef RH_PHASE_TIMING
#define RH_PHASE_BEGIN unsigned long long ph_t = __builtin_readcyclecounter(), ph_acc[4] = {0, 0, 0, 0};
#define RH_PHASE_MARK(K) { const unsigned long long now_ = __builtin_readcyclecounter(); ph_acc[K] += now_ - ph_t; ph_t = now_; }
#define RH_PHASE_END if ((blockIdx.x % 1019u) == 7u && lane == 0) \
    printf("tile %u records %u: load %llu, stage %llu, shade %llu, store %llu\n", tile, end - start, ph_acc[0], ph_acc[1], ph_acc[2], ph_acc[3]);
#elif defined(RH_WG_TIMES)
// (-DRH_WG_TIMES, tools/wg_times.sh: every workgroup notes when it started and ended, on the 100 MHz clock all CUs share;
// the last one to end prints every fourth note -- after the fact, so that printing does not disturb what is measured)
__device__ unsigned long long rh_wg_t[3 * 32768];
__device__ unsigned int rh_wg_done;
#define RH_PHASE_BEGIN const unsigned long long wg_t0 = wall_clock64();
#define RH_PHASE_MARK(K)
#define RH_PHASE_END if (lane == 0 && gridDim.x <= 32768u) { \
        rh_wg_t[3 * blockIdx.x] = wg_t0; rh_wg_t[3 * blockIdx.x + 1] = wall_clock64(); rh_wg_t[3 * blockIdx.x + 2] = (unsigned long long)rows | ((unsigned long long)(end - start) << 8); \
        __threadfence(); \
        if (atomicAdd(&rh_wg_done, 1u) == gridDim.x - 1u) { \
            rh_wg_done = 0u; __threadfence(); \
            printf("launch %u workgroups\n", gridDim.x); \
            for (unsigned int i = 0; i < gridDim.x; i += 4u) printf("wg %u rows %u pairs %u t %llu %llu\n", i, (unsigned int)(rh_wg_t[3 * i + 2] & 255u), (unsigned int)(rh_wg_t[3 * i + 2] >> 8), rh_wg_t[3 * i], rh_wg_t[3 * i + 1]); \
        } }
#else
#define RH_PHASE_BEGIN
#define RH_PHASE_MARK(K)
#define RH_PHASE_END
#endif

namespace rh {

RH_DEVICE uint32_t swap_rb(uint32_t v) { return (v & 0xff00ff00u) | ((v & 0xffu) << 16) | ((v >> 16) & 0xffu); }

RH_DEVICE float blend_factor(uint32_t f, float src_alpha) {
    return f == 0u ? 0.0f : (f == 1u ? 1.0f : (f == 2u ? src_alpha : 1.0f - src_alpha));
}

// out = src*SF + dst*DF for every channel, alpha included; evaluated on the 0..255 scale:
//   q = rint(clamp(fma(dst_byte, DF, (src*SF)*255), 0, 255)), NaN -> 0, ties to even
// v_cvt_pk_u8_f32 performs exactly that clamp + round + byte insert.
// BLEND: 0 = factors read at run time, 1 = (One, OneMinusSrcAlpha), 2 = (One, One); the results are identical.
template <int BLEND>
RH_DEVICE uint32_t blend(uint32_t dst, float4 s, uint32_t src_factor, uint32_t dst_factor) {
#if defined(RH_EXP) && RH_EXP == 3          /* diagnostic build: no blend arithmetic */
    return dst ^ __float_as_uint(s.x + s.y + s.z + s.w);
#endif
    float df;
    if (BLEND == 1) df = 1.0f - s.w;
    else if (BLEND == 2) df = 1.0f;
    else {
        df = blend_factor(dst_factor, s.w);
        if (src_factor != 1u) { const float sf = blend_factor(src_factor, s.w); s.x *= sf; s.y *= sf; s.z *= sf; s.w *= sf; }
    }
    const float o0 = __builtin_fmaf((float)(dst & 0xffu), df, s.x * 255.0f);
    const float o1 = __builtin_fmaf((float)((dst >> 8) & 0xffu), df, s.y * 255.0f);
    const float o2 = __builtin_fmaf((float)((dst >> 16) & 0xffu), df, s.z * 255.0f);
    const float o3 = __builtin_fmaf((float)(dst >> 24), df, s.w * 255.0f);
    uint32_t q = 0;
    q = __builtin_amdgcn_cvt_pk_u8_f32(o0, 0, q);
    q = __builtin_amdgcn_cvt_pk_u8_f32(o1, 1, q);
    q = __builtin_amdgcn_cvt_pk_u8_f32(o2, 2, q);
    q = __builtin_amdgcn_cvt_pk_u8_f32(o3, 3, q);
    return q;
}

RH_DEVICE int64_t clamp_i64(int64_t v, int64_t lim) { return v < -lim ? -lim : (v > lim ? lim : v); }

// ---- triangle set-up ---------------------------------------------------------------------------------------------------
struct TriSetup {
    float XR = 0.0f, YR = 0.0f;
    float v0[NV] = {}, dvdx[NV] = {}, dvdy[NV] = {};
    float q0 = 0.0f, qx = 0.0f, qy = 0.0f;
    int32_t ei[9] = {};            // A[3], B[3], T[3] of the edge functions, tile-relative
    int64_t cf[3] = {};            // per edge: the constant in 1/256 px with the top-left rule folded in
};

// Triangle set-up relative to tile origin (ox, oy).  Returns false when the triangle contributes nothing to this tile;
// otherwise fills `t` and grows the tile-relative bbox.  `big` is set when an edge coefficient is too large for the
// f32 coverage path.  cf: inside at sub-pixel position P <=> A*Px + B*Py + cf >= 0; multisampled passes derive
// per-sample constants from it.
// `edges`: bit K set = the constant of edge K is needed (an axis-aligned quad with affine varyings is drawn from its
// shared diagonal alone: the 64-bit work of the other edges is skipped; their coefficients still decide `big`).
RH_DEVICE bool setup_triangle(const rh_draw_params& p, SnappedVertex a, SnappedVertex b, SnappedVertex c, int ox, int oy,
                              TriSetup& t, bool& persp, bool& big, int& bx0, int& by0, int& bx1, int& by1, uint32_t edges = 7u) {
    int x0, y0, x1, y1;
    if (!tri_pixel_bbox(p, a, b, c, x0, y0, x1, y1)) return false;
    x0 = max(x0, ox); y0 = max(y0, oy); x1 = min(x1, ox + TILE - 1); y1 = min(y1, oy + TILE - 1);
    if (x0 > x1 || y0 > y1) return false;
    const int64_t area2 = (int64_t)(b.X - a.X) * (c.Y - a.Y) - (int64_t)(c.X - a.X) * (b.Y - a.Y);
    if (area2 == 0) return false;
    if (area2 < 0) { SnappedVertex s = b; b = c; c = s; }      // no culling: both windings are drawn
    // T is relative to the tile origin and saturated at +-2^30: |A*lx + B*ly| < 2^30 for |A|, |B| < 2^23, lx, ly < 32
    // keeps the sign (and the sum inside 32 bits).
#define RH_EDGE(P, Q, K) { \
        const int32_t A_ = P.Y - Q.Y, B_ = Q.X - P.X; \
        t.ei[K] = A_; t.ei[3 + K] = B_; \
        if (edges & (1u << K)) { \
            const int64_t C_ = -((int64_t)A_ * P.X + (int64_t)B_ * P.Y); \
            const bool tl_ = (A_ > 0) || (A_ == 0 && B_ > 0); \
            t.cf[K] = C_ + (tl_ ? 0 : -1); \
            const int64_t Cq_ = t.cf[K] >> RH_SUBPIXEL_BITS; \
            t.ei[6 + K] = (int32_t)clamp_i64(Cq_ + (int64_t)A_ * ox + (int64_t)B_ * oy, (int64_t)1 << 30); \
        } \
        if (A_ >= F32_EDGE_LIMIT || A_ <= -F32_EDGE_LIMIT || B_ >= F32_EDGE_LIMIT || B_ <= -F32_EDGE_LIMIT) big = true; }
    RH_EDGE(a, b, 0)
    RH_EDGE(b, c, 1)
    RH_EDGE(c, a, 2)
#undef RH_EDGE
    t.XR = (float)(a.X - ox * 256); t.YR = (float)(a.Y - oy * 256);
    const float k = 1.0f / 256.0f;
    const float e1x = (float)(b.X - a.X) * k, e1y = (float)(b.Y - a.Y) * k;
    const float e2x = (float)(c.X - a.X) * k, e2y = (float)(c.Y - a.Y) * k;
    const float area = __builtin_fmaf(e1x, e2y, -(e2x * e1y));
    const float inv = 1.0f / area;
    persp = !(a.rw == 1.0f && b.rw == 1.0f && c.rw == 1.0f);
    const float* fa = reinterpret_cast<const float*>(&a.var);
    const float* fb = reinterpret_cast<const float*>(&b.var);
    const float* fc = reinterpret_cast<const float*>(&c.var);
#define RH_PLANE(F0, F1, F2, O0, OX, OY) { \
        const float f0_ = (F0), d1_ = (F1) - f0_, d2_ = (F2) - f0_; \
        O0 = f0_; \
        OX = (__builtin_fmaf(d1_, e2y, -(d2_ * e1y)) * inv) * k; \
        OY = (__builtin_fmaf(d2_, e1x, -(d1_ * e2x)) * inv) * k; }
#pragma unroll
    for (int v = 0; v < NV; v++) {
        if (persp) RH_PLANE(fa[v] * a.rw, fb[v] * b.rw, fc[v] * c.rw, t.v0[v], t.dvdx[v], t.dvdy[v])
        else RH_PLANE(fa[v], fb[v], fc[v], t.v0[v], t.dvdx[v], t.dvdy[v])
    }
    RH_PLANE(a.rw, b.rw, c.rw, t.q0, t.qx, t.qy)
#undef RH_PLANE
    bx0 = min(bx0, x0 - ox); by0 = min(by0, y0 - oy); bx1 = max(bx1, x1 - ox); by1 = max(by1, y1 - oy);
    return true;
}

// The 9 edge values of a triangle as stored: f32 (T saturated at +-2^24) or, for a BIG cluster, raw i32 bits.
RH_DEVICE uint32_t edge_word(const TriSetup& t, int i, bool big) {
    if (big) return (uint32_t)t.ei[i];
    int32_t v = t.ei[i];
    if (i >= 6) { const int32_t lim = 1 << 24; v = v < -lim ? -lim : (v > lim ? lim : v); }
    return __float_as_uint((float)v);
}

// Standard sample locations (Vulkan 1.3, "Multisampling"), as offsets from the pixel centre in 1/256 px.
template <int S> RH_DEVICE void sample_offset(int s, int& dx, int& dy) {
    constexpr int8_t O2[2][2] = { { 64, 64 }, { -64, -64 } };
    constexpr int8_t O4[4][2] = { { -32, -96 }, { 96, -32 }, { -96, 32 }, { 32, 96 } };
    constexpr int8_t O8[8][2] = { { 16, -48 }, { -16, 48 }, { 80, 16 }, { -48, -80 }, { -80, 80 }, { -112, -16 }, { 48, 112 }, { 112, -112 } };
    if (S == 2) { dx = O2[s & 1][0]; dy = O2[s & 1][1]; }
    else if (S == 4) { dx = O4[s & 3][0]; dy = O4[s & 3][1]; }
    else { dx = O8[s & 7][0]; dy = O8[s & 7][1]; }
}

// Integer edge functions of a cluster's two triangles for a multisampled pass: sample s of tile-relative pixel
// (lx, ly) is inside edge e of triangle t <=> A[t][e]*lx + B[t][e]*ly + T[t][s][e] >= 0.  With the sample at
// P = 256*pixel + offset_s the exact test A*Px + B*Py + cf >= 0 equals A*i + B*j + floor((cf + A*dx_s + B*dy_s) / 256) >= 0.
template <int S>
struct MsRecord {
    int32_t A[2][3], B[2][3];
    int32_t T[2][S][3];
};
template <int S> constexpr bool ms_words_ok() { return sizeof(MsRecord<S>) == 4u * RH_MS_EXTRA_WORDS(S); }
static_assert(ms_words_ok<2>() && ms_words_ok<4>() && ms_words_ok<8>(), "host and device agree on the multisampled record size");

template <int S>
RH_DEVICE void store_ms_edges(MsRecord<S>& m, int t, const TriSetup& ts, int ox, int oy) {
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const int32_t A = ts.ei[e], B = ts.ei[3 + e];
        m.A[t][e] = A; m.B[t][e] = B;
#pragma unroll
        for (int sidx = 0; sidx < S; sidx++) {
            int dx, dy; sample_offset<S>(sidx, dx, dy);
            const int64_t q = (ts.cf[e] + (int64_t)A * dx + (int64_t)B * dy) >> RH_SUBPIXEL_BITS;
            m.T[t][sidx][e] = (int32_t)clamp_i64(q + (int64_t)A * ox + (int64_t)B * oy, (int64_t)1 << 30);
        }
    }
}

// What the set-up of a pair's (at most) two triangles found.
struct PairSetup {
    TriSetup ta, tb;
    bool has_a = false, has_b = false, persp_a = false, persp_b = false, big = false, rect = false;
    int bx0 = 255, by0 = 255, bx1 = -1, by1 = -1;          // tile-relative bounding box of the two
    uint32_t bbox = BBOX_EMPTY;                              // the box the record gets
};

// Writes the pair's set-up record to `w` (REC_WORDS dwords) and, when the fragments will read it, the extended record.
template <int S>
RH_DEVICE void write_records(const PairSetup& s, int ox, int oy, uint32_t* w, uint32_t* ext) {
    const bool has_a = s.has_a, has_b = s.has_b, rect = s.rect, big = s.big;
    uint32_t flags = (has_a ? FLAG_HAS_A : 0u) | (has_b ? FLAG_HAS_B : 0u) | (s.persp_a ? FLAG_PERSP_A : 0u) | (s.persp_b ? FLAG_PERSP_B : 0u) |
                     (big ? FLAG_BIG : 0u) | (rect ? FLAG_RECT : 0u);
    const uint32_t bbox = s.bbox;
    if (bbox == BBOX_EMPTY) flags = 0u;
    // the record (the planes of a missing triangle are zeros: never used, masked by the flags)
#pragma unroll
    for (int i = 0; i < REC_WORDS; i++) w[i] = 0u;
    {   float g = 0.0f;
        const TriSetup& tg = has_a ? s.ta : s.tb;
#pragma unroll
        for (int v = 0; v < NV; v++) g = fmaxf(g, fabsf(tg.dvdx[v]));
        w[0] = hdr_pack(bbox, flags, g);
    }
    w[1] = edge_word(s.ta, 1, false); w[2] = edge_word(s.ta, 4, false); w[3] = edge_word(s.ta, 7, false);      // the shared diagonal (used when RECT: never BIG)
    const TriSetup* const ts[2] = { &s.ta, &s.tb };
#pragma unroll
    for (int t = 0; t < 2; t++) {
        uint32_t* o = w + HDR_WORDS + t * TRI_WORDS;
        o[OFF_XR] = __float_as_uint(ts[t]->XR); o[OFF_YR] = __float_as_uint(ts[t]->YR);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            o[OFF_V0 + v] = __float_as_uint(ts[t]->v0[v]); o[OFF_DVDX + v] = __float_as_uint(ts[t]->dvdx[v]); o[OFF_DVDY + v] = __float_as_uint(ts[t]->dvdy[v]);
        }
    }
    // the extended record: only for clusters whose fragments read it
    if (ext && bbox != BBOX_EMPTY && (!rect || s.persp_a || s.persp_b)) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            uint32_t* o = ext + t * EXT_TRI_WORDS;
#pragma unroll
            for (int i = 0; i < 9; i++) o[i] = edge_word(*ts[t], i, big);
            o[EOFF_Q] = __float_as_uint(ts[t]->q0); o[EOFF_Q + 1] = __float_as_uint(ts[t]->qx); o[EOFF_Q + 2] = __float_as_uint(ts[t]->qy);
        }
    }
    if (S > 1 && ext && bbox != BBOX_EMPTY) {
        MsRecord<S>& m = *reinterpret_cast<MsRecord<S>*>(ext + EXT_WORDS);
        if (has_a) store_ms_edges<S>(m, 0, s.ta, ox, oy);
        if (has_b) store_ms_edges<S>(m, 1, s.tb, ox, oy);
    }
}

// Sub-record `piece` (0..3) of triangle (i0, i1, i2) of a clipped cluster: triangles 2*piece and 2*piece + 1 of the fan
// of the triangle's part inside the clip volume; a triangle that lies inside as it is (`whole`) is piece 0 by itself.
// The rare path: out of line, and it writes the records where they go (`w`: the pair's place in the staging area).
template <int S>
__device__ __noinline__ void setup_clipped(const rh_draw_params& p, uint32_t instance, uint32_t i0, uint32_t i1, uint32_t i2, bool whole,
                                           uint32_t piece, int ox, int oy, uint32_t* w, uint32_t* ext) {
    PairSetup s;
    if (whole) {
        if (piece == 0u) {
            const SnappedVertex v0 = run_vertex(p, i0, instance), v1 = run_vertex(p, i1, instance), v2 = run_vertex(p, i2, instance);
            s.has_a = setup_triangle(p, v0, v1, v2, ox, oy, s.ta, s.persp_a, s.big, s.bx0, s.by0, s.bx1, s.by1);
        }
    } else {
        ClipPolygon poly;
        clip_triangle(p, instance, i0, i1, i2, poly);
        const int f = 2 * (int)piece;
        if (f + 2 < poly.n) s.has_a = setup_triangle(p, poly.v[0], poly.v[f + 1], poly.v[f + 2], ox, oy, s.ta, s.persp_a, s.big, s.bx0, s.by0, s.bx1, s.by1);
        if (f + 3 < poly.n) s.has_b = setup_triangle(p, poly.v[0], poly.v[f + 2], poly.v[f + 3], ox, oy, s.tb, s.persp_b, s.big, s.bx0, s.by0, s.bx1, s.by1);
    }
    if (s.has_a || s.has_b) s.bbox = (uint32_t)s.bx0 | ((uint32_t)s.by0 << 8) | ((uint32_t)s.bx1 << 16) | ((uint32_t)s.by1 << 24);
    write_records<S>(s, ox, oy, w, ext);
}

// The two triangles (v0, v1, v2), (v1, v3, v2) of a strip quad for the tile at (ox, oy).  A quad whose snapped corners
// form an axis-aligned rectangle split along v1-v2 is drawn from its exact pixel box and the shared diagonal alone (edge 1
// of triangle A) when its varyings are affine and its sides short enough for the f32 coverage path (a lean quad,
// rh_geometry.h).  `known_lean`: the caller knows that it is one (the tile kernel: the binning kernels said so with the
// same test on the same vertices), so the code of every other case folds away.
template <int S>
RH_DEVICE void setup_strip_quad(const rh_draw_params& p, const SnappedVertex& v0, const SnappedVertex& v1, const SnappedVertex& v2, const SnappedVertex& v3,
                                int ox, int oy, PairSetup& s, bool known_lean) {
    bool shape = true, lean = true;
    if (!known_lean) { lean = lean_quad(p, v0, v1, v2, v3, shape); if (S != 1) { shape = false; lean = false; } }
    s.has_a = setup_triangle(p, v0, v1, v2, ox, oy, s.ta, s.persp_a, s.big, s.bx0, s.by0, s.bx1, s.by1, lean ? 2u : 7u);
    s.has_b = setup_triangle(p, v1, v3, v2, ox, oy, s.tb, s.persp_b, s.big, s.bx0, s.by0, s.bx1, s.by1, lean ? 0u : 7u);
    s.rect = shape && s.has_a && s.has_b && !s.big;
    if (s.has_a || s.has_b) {
        if (s.rect) {
            // Exact pixel set of the rectangle under the top-left rule: left and top edges take pixel centres on them,
            // right and bottom edges do not.  Inside it every pixel belongs to exactly one of the two triangles, decided
            // by the shared diagonal alone (edge 1 of triangle A).
            const int32_t rminX = min(v0.X, v3.X), rmaxX = max(v0.X, v3.X), rminY = min(v0.Y, v3.Y), rmaxY = max(v0.Y, v3.Y);
            const int ex0 = max(max(p.sc_x0, ox), (rminX + 255) >> 8), ex1 = min(min(p.sc_x1 - 1, ox + TILE - 1), (rmaxX - 1) >> 8);
            const int ey0 = max(max(p.sc_y0, oy), (rminY + 255) >> 8), ey1 = min(min(p.sc_y1 - 1, oy + TILE - 1), (rmaxY - 1) >> 8);
            if (ex0 <= ex1 && ey0 <= ey1) s.bbox = (uint32_t)(ex0 - ox) | ((uint32_t)(ey0 - oy) << 8) | ((uint32_t)(ex1 - ox) << 16) | ((uint32_t)(ey1 - oy) << 24);
        } else s.bbox = (uint32_t)s.bx0 | ((uint32_t)s.by0 << 8) | ((uint32_t)s.bx1 << 16) | ((uint32_t)s.by1 << 24);
    }
}

// The record of a lean quad (pair value without its flags: the cluster), set up by the tile wave itself: lane = pair,
// written straight to the pair's place in the LDS staging area `w`; nothing of it goes through memory.  The arithmetic
// is that of run_vertex / setup_triangle / setup_strip_quad / write_records for such a quad, operation for operation
// (the set-up launch computes the same record when the device option RH_OPTION_SETUP_LAUNCH sends lean quads to it, and
// the tests compare the two), with what a lean quad makes constant left out: every vertex is inside the clip volume
// with w = 1, both triangles have the rectangle's bounding box, nothing is perspective or BIG, and only the shared
// diagonal's constant is needed.  A quad without pixels in the tile leaves an empty header and nothing else.
struct LeanVertex { int32_t X, Y; Varyings var; };
RH_DEVICE LeanVertex run_vertex_lean(const rh_draw_params& p, uint32_t vertex, uint32_t instance) {
    VertexContext c; c.p = &p; c.vertex_index = vertex; c.instance_index = instance;
    LeanVertex r;
    const float4 pos = vertex_main(c, r.var);
    r.X = snap(__builtin_fmaf(pos.x, p.vp_hw, p.vp_ox)); r.Y = snap(__builtin_fmaf(pos.y, -p.vp_hh, p.vp_oy));
    return r;
}
// Triangle (a, b, c) of a lean quad: its TRI_WORDS dwords of the record to `o`; `diag` (triangle A): the shared diagonal
// (edge 1) to dg[0..2]; `g`: the largest |d varying / dx| of the triangle.
RH_DEVICE void lean_triangle(const LeanVertex& a, LeanVertex b, LeanVertex c, int ox, int oy, uint32_t* o, bool diag, uint32_t* dg, float& g) {
    const int64_t area2 = (int64_t)(b.X - a.X) * (c.Y - a.Y) - (int64_t)(c.X - a.X) * (b.Y - a.Y);
    if (area2 < 0) { const LeanVertex t = b; b = c; c = t; }
    if (diag) {
        const int32_t A_ = b.Y - c.Y, B_ = c.X - b.X;
        const int64_t C_ = -((int64_t)A_ * b.X + (int64_t)B_ * b.Y);
        const bool tl_ = (A_ > 0) || (A_ == 0 && B_ > 0);
        const int64_t Cq_ = (C_ + (tl_ ? 0 : -1)) >> RH_SUBPIXEL_BITS;
        int32_t T = (int32_t)clamp_i64(Cq_ + (int64_t)A_ * ox + (int64_t)B_ * oy, (int64_t)1 << 30);
        const int32_t lim = 1 << 24;
        T = T < -lim ? -lim : (T > lim ? lim : T);
        dg[0] = __float_as_uint((float)A_); dg[1] = __float_as_uint((float)B_); dg[2] = __float_as_uint((float)T);
    }
    o[OFF_XR] = __float_as_uint((float)(a.X - ox * 256)); o[OFF_YR] = __float_as_uint((float)(a.Y - oy * 256));
    const float k = 1.0f / 256.0f;
    const float e1x = (float)(b.X - a.X) * k, e1y = (float)(b.Y - a.Y) * k;
    const float e2x = (float)(c.X - a.X) * k, e2y = (float)(c.Y - a.Y) * k;
    const float area = __builtin_fmaf(e1x, e2y, -(e2x * e1y));
    const float inv = 1.0f / area;
    const float* fa = reinterpret_cast<const float*>(&a.var);
    const float* fb = reinterpret_cast<const float*>(&b.var);
    const float* fc = reinterpret_cast<const float*>(&c.var);
    g = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const float f0 = fa[v], d1 = fb[v] - f0, d2 = fc[v] - f0;
        const float dvdx = (__builtin_fmaf(d1, e2y, -(d2 * e1y)) * inv) * k;
        const float dvdy = (__builtin_fmaf(d2, e1x, -(d1 * e2x)) * inv) * k;
        o[OFF_V0 + v] = __float_as_uint(f0); o[OFF_DVDX + v] = __float_as_uint(dvdx); o[OFF_DVDY + v] = __float_as_uint(dvdy);
        g = fmaxf(g, fabsf(dvdx));
    }
#pragma unroll
    for (int i = 2 + 3 * NV; i < TRI_WORDS; i++) o[i] = 0u;
}
// Two lanes set a quad up together, lanes l and l + ROUND of the wave (a round brings ROUND <= 32 pairs): lane l runs
// vertices 0 and 1 and sets up triangle A with the header and the diagonal, lane l + ROUND runs vertices 3 and 2 and sets
// up triangle B; what the other needs of one's vertices changes sides through the wave's lane permute.  `half`: 0 for
// lane l, 1 for l + ROUND; both lanes are active together (they hold the same pair value).
RH_DEVICE uint32_t from_partner(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ ROUND) << 2, (int)v); }
// (r0, rows): the rows of the tile the workgroup works on (a band, or all 32).  The record's box is cut to them here
// already -- the wave cuts every box to its rows anyway -- so a quad that misses the band ends after its first vertex.
RH_DEVICE void setup_lean_pair(const rh_draw_params& p, uint32_t cluster, int ox, int oy, int r0, int rows, uint32_t* w, int lane) {
    const bool half = lane >= ROUND;
#if defined(RH_EXP) && RH_EXP == 6          /* diagnostic build: no set-up at all */
    w[0] = HDR_EMPTY; return;
#endif
    const uint32_t instance = p.cpi == 1u ? cluster : cluster / p.cpi, first = (cluster - instance * p.cpi) * 2u;     // (a quad per instance: no division)
    // the corner this lane runs (vertex 0 or 3) and the partner's
    const LeanVertex corner = run_vertex_lean(p, first + (half ? 3u : 0u), instance);
    const int32_t oX = (int32_t)from_partner((uint32_t)corner.X, lane), oY = (int32_t)from_partner((uint32_t)corner.Y, lane);
    // the bounding box of either triangle is the rectangle's (tri_pixel_bbox, cut to the tile), and the exact pixel set of
    // the rectangle under the top-left rule (setup_strip_quad)
    // (the padded box of tri_pixel_bbox contains the exact one -- the pad is never negative -- so the exact box alone
    // decides whether the record is empty)
    const int32_t minX = min(corner.X, oX), maxX = max(corner.X, oX), minY = min(corner.Y, oY), maxY = max(corner.Y, oY);
    const int ex0 = max(max(p.sc_x0, ox), (minX + 255) >> 8), ex1 = min(min(p.sc_x1 - 1, ox + TILE - 1), (maxX - 1) >> 8);
    const int ey0 = max(max(p.sc_y0, oy + r0), (minY + 255) >> 8), ey1 = min(min(p.sc_y1 - 1, oy + r0 + rows - 1), (maxY - 1) >> 8);
    if (ex0 > ex1 || ey0 > ey1) { if (!half) w[0] = HDR_EMPTY; return; }     // (the same verdict in both lanes)
    const uint32_t bbox = (uint32_t)(ex0 - ox) | ((uint32_t)(ey0 - oy) << 8) | ((uint32_t)(ex1 - ox) << 16) | ((uint32_t)(ey1 - oy) << 24);
    // the vertex on the diagonal this lane runs (vertex 1 or 2) and the partner's
    const LeanVertex mine = run_vertex_lean(p, first + (half ? 2u : 1u), instance);
    LeanVertex other;
    other.X = (int32_t)from_partner((uint32_t)mine.X, lane); other.Y = (int32_t)from_partner((uint32_t)mine.Y, lane);
    {   const float* mf = reinterpret_cast<const float*>(&mine.var); float* of = reinterpret_cast<float*>(&other.var);
#pragma unroll
        for (int v = 0; v < NV; v++) of[v] = __uint_as_float(from_partner(__float_as_uint(mf[v]), lane));
    }
    // triangle A = (v0, v1, v2) = (corner, mine, other); triangle B = (v1, v3, v2) = (other, corner, mine)
    LeanVertex ta, tb, tc;
    ta.X = half ? other.X : corner.X; ta.Y = half ? other.Y : corner.Y;
    tb.X = half ? corner.X : mine.X; tb.Y = half ? corner.Y : mine.Y;
    tc.X = half ? mine.X : other.X; tc.Y = half ? mine.Y : other.Y;
    {   const float* cf = reinterpret_cast<const float*>(&corner.var); const float* mf = reinterpret_cast<const float*>(&mine.var);
        const float* of = reinterpret_cast<const float*>(&other.var);
        float* fa = reinterpret_cast<float*>(&ta.var); float* fb = reinterpret_cast<float*>(&tb.var); float* fc = reinterpret_cast<float*>(&tc.var);
#pragma unroll
        for (int v = 0; v < NV; v++) { fa[v] = half ? of[v] : cf[v]; fb[v] = half ? cf[v] : mf[v]; fc[v] = half ? mf[v] : of[v]; }
    }
    uint4* const q = reinterpret_cast<uint4*>(w);
    uint32_t t[TRI_WORDS], dg[3];
    float g;
    lean_triangle(ta, tb, tc, ox, oy, t, !half, dg, g);
    if (!half) q[0] = make_uint4(hdr_pack(bbox, FLAG_HAS_A | FLAG_HAS_B | FLAG_RECT, g), dg[0], dg[1], dg[2]);
    uint4* const qt = q + HDR_WORDS / 4 + (half ? TRI_WORDS / 4 : 0);
#pragma unroll
    for (int i = 0; i < TRI_WORDS / 4; i++) qt[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
}

// Sets up the triangles of pair value `value` (cluster | sub-record << CLIP_SUB_SHIFT) for the tile at (ox, oy) and
// writes the pair's records (`stage`: the pair's REC_WORDS dwords of the staging area).  S > 1 (multisampled pass) also
// writes the per-sample constants and never takes the exact-box shortcut.  CLIP: the clipping kernels (see
// rh_geometry.h); the others leave out what is not inside the clip volume as it is.
template <int S, bool CLIP>
RH_DEVICE void setup_pair(const rh_draw_params& p, uint32_t value, int ox, int oy, uint4* stage, uint32_t* ext) {
    const uint32_t cluster = value & CLIP_CLUSTER_MASK, sub = value >> CLIP_SUB_SHIFT;
    const uint32_t instance = cluster / p.cpi, local = cluster - instance * p.cpi;
    const uint32_t tri0 = local * 2;
    const bool has_second = tri0 + 1 < p.tpi;
    // Strip: triangle A = (v0, v1, v2), triangle B = (v1, v3, v2) with v_i = vertex tri0 + i (4 vertex runs).
    // List:  triangle A = (v0, v1, v2), triangle B = (v3, v4, v5) with v_i = vertex 3*tri0 + i.
    const bool strip = p.topology == 1u;
    const uint32_t first = strip ? tri0 : 3 * tri0;
    const SnappedVertex v0 = run_vertex(p, first, instance);
    const SnappedVertex v1 = run_vertex(p, first + 1, instance);
    const SnappedVertex v2 = run_vertex(p, first + 2, instance);
    // triangle B's vertices (b0, b1, b2)
    SnappedVertex b0 = v1, b1 = v1, b2 = v2;
    if (has_second) {
        b1 = run_vertex(p, first + 3, instance);
        if (!strip) { b0 = b1; b1 = run_vertex(p, first + 4, instance); b2 = run_vertex(p, first + 5, instance); }
    }
    if (CLIP) {
        const int kind_a = classify(v0, v1, v2), kind_b = has_second ? classify(b0, b1, b2) : TRI_DROPPED;
        if (kind_a == TRI_CLIPPED || kind_b == TRI_CLIPPED) {
            // a clipped cluster: sub-records 0..3 belong to triangle A, 4..7 to triangle B
            uint32_t* const w = reinterpret_cast<uint32_t*>(stage);
            const int kind = sub < 4u ? kind_a : kind_b;
            if (kind == TRI_DROPPED) { PairSetup none; write_records<S>(none, ox, oy, w, ext); }
            else if (sub < 4u) setup_clipped<S>(p, instance, first, first + 1, first + 2, kind == TRI_WHOLE, sub, ox, oy, w, ext);
            else setup_clipped<S>(p, instance, strip ? first + 1 : first + 3, strip ? first + 3 : first + 4, strip ? first + 2 : first + 5, kind == TRI_WHOLE, sub - 4u, ox, oy, w, ext);
            return;
        }
    }
    PairSetup s;
    if (has_second && strip) setup_strip_quad<S>(p, v0, v1, v2, b1, ox, oy, s, false);
    else {
        s.has_a = setup_triangle(p, v0, v1, v2, ox, oy, s.ta, s.persp_a, s.big, s.bx0, s.by0, s.bx1, s.by1);
        if (has_second) s.has_b = setup_triangle(p, b0, b1, b2, ox, oy, s.tb, s.persp_b, s.big, s.bx0, s.by0, s.bx1, s.by1);
        if (s.has_a || s.has_b) s.bbox = (uint32_t)s.bx0 | ((uint32_t)s.by0 << 8) | ((uint32_t)s.bx1 << 16) | ((uint32_t)s.by1 << 24);
    }
    uint32_t w[REC_WORDS];
    write_records<S>(s, ox, oy, w, ext);
#pragma unroll
    for (int q = 0; q < REC_WORDS / 4; q++) stage[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

template <int S, bool CLIP = false>
RH_DEVICE void setup_kernel(const rh_setup_args& a) {
    uint32_t n = a.n_pairs;
    if (a.pair_total) { if (*a.pair_total > n) return; n = (uint32_t)*a.pair_total; }   // (n arrives as the room for the pairs)
    // The records of the workgroup's 256 pairs are gathered in LDS and leave as one contiguous run of 16-byte pieces
    // (a thread's own record is 96 bytes: stored directly, the lanes of a store would be 96 bytes apart).
    __shared__ uint4 stage[256 * (REC_WORDS / 4)];
    const uint32_t first = blockIdx.x * 256u, i = first + threadIdx.x;
    if (first >= n) return;
    uint32_t tile = 0u, value = 0u;
    if (i < n) {
        tile = a.tiles[i]; value = a.clusters[i];
        // the ends of a tile's run of pairs
        if (i == 0u || a.tiles[i - 1u] != tile) a.tile_start[tile] = i;
        if (i == n - 1u || a.tiles[i + 1u] != tile) a.tile_end[tile] = i + 1u;
    }
    // Lean quads (pair values without PAIR_GENERAL) are set up by the tile kernel itself: a workgroup all of whose pairs
    // are lean writes no records at all.  (One with some general pair writes its whole run, the lean ones included:
    // nobody reads those.)
    if (!__syncthreads_or(i < n && (value & PAIR_GENERAL) != 0u)) return;
    if (i < n) {
        const int ox = (int)(tile % a.p.tiles_x) * TILE, oy = (int)(tile / a.p.tiles_x) * TILE;
        setup_pair<S, CLIP>(a.p, value & ~PAIR_GENERAL, ox, oy, &stage[threadIdx.x * (REC_WORDS / 4)], reinterpret_cast<uint32_t*>(a.ext + (size_t)i * a.ext_bytes));
    }
    __syncthreads();
    const uint32_t pieces = min(256u, n - first) * (REC_WORDS / 4);
    uint4* const out = reinterpret_cast<uint4*>(a.rec + (size_t)first * (REC_WORDS * 4));
    for (uint32_t q = threadIdx.x; q < pieces; q += 256u) out[q] = stage[q];
}

// f32 coverage: with |A|, |B| < 2^17 and lx, ly < 32 every product is an exactly representable integer below 2^22,
// T is an exactly representable integer in [-2^24, 2^24], and |A*lx + B*ly| < 2^24.  Whenever the true edge value
// A*lx + B*ly + T is within a few units of zero every partial sum is below 2^24 and therefore exact, so the sign test
// agrees with integer arithmetic; when T is saturated the sign is that of T.
RH_DEVICE bool edges_inside_f32(const float* e, float lxf, float lyf) {
    const float s0 = __builtin_fmaf(e[EOFF_A + 0], lxf, __builtin_fmaf(e[EOFF_B + 0], lyf, e[EOFF_T + 0]));
    const float s1 = __builtin_fmaf(e[EOFF_A + 1], lxf, __builtin_fmaf(e[EOFF_B + 1], lyf, e[EOFF_T + 1]));
    const float s2 = __builtin_fmaf(e[EOFF_A + 2], lxf, __builtin_fmaf(e[EOFF_B + 2], lyf, e[EOFF_T + 2]));
    return ((__float_as_int(s0) | __float_as_int(s1) | __float_as_int(s2)) >= 0);     // no sign bit set (a zero sum is +0)
}

RH_DEVICE bool edges_inside_i32(const float* e, int lx, int ly) {
    const int s0 = __mul24(__float_as_int(e[EOFF_A + 0]), lx) + __mul24(__float_as_int(e[EOFF_B + 0]), ly) + __float_as_int(e[EOFF_T + 0]);
    const int s1 = __mul24(__float_as_int(e[EOFF_A + 1]), lx) + __mul24(__float_as_int(e[EOFF_B + 1]), ly) + __float_as_int(e[EOFF_T + 1]);
    const int s2 = __mul24(__float_as_int(e[EOFF_A + 2]), lx) + __mul24(__float_as_int(e[EOFF_B + 2]), ly) + __float_as_int(e[EOFF_T + 2]);
    return (s0 | s1 | s2) >= 0;
}

constexpr int PACK_MAX = 4;               // boxes that may share one shading pass
template <bool B> struct Flag { static constexpr bool value = B; };
template <int N> struct Int { static constexpr int value = N; };

// ---- attachment pixels --------------------------------------------------------------------------------------------
// LDS slot of one attachment pixel.  Narrow kernels (every attachment 8-bit UNORM): the packed pixel, r in byte 0 (an R8
// attachment loads and stores byte 0 only).  Wide kernels (some attachment is RGBA16F / RGBA32F): 16 bytes; a float
// attachment keeps its four channels as f32 (RGBA16F values always rounded to f16, as the memory would hold them), a
// UNORM attachment its packed pixel in .x.
template <bool WIDE> struct Px { using T = uint32_t; };
template <> struct Px<true> { using T = uint4; };

RH_DEVICE uint4 wide_words(uint32_t v) { return make_uint4(v, 0u, 0u, 0u); }
RH_DEVICE uint4 wide_words(uint4 v) { return v; }
RH_DEVICE uint32_t wide_pixel(uint4 v, uint32_t*) { return v.x; }
RH_DEVICE uint4 wide_pixel(uint4 v, uint4*) { return v; }

RH_DEVICE bool is_float_format(uint32_t f) { return f == FORMAT_RGBA_F16 || f == FORMAT_RGBA_F32; }
RH_DEVICE float round_to_f16(float v) { return (float)(_Float16)v; }

RH_DEVICE uint32_t load_unorm(const rh_attach_dev& at, size_t pix) {
    if (at.format == FORMAT_R_U8) return at.ptr[pix];
    const uint32_t v = reinterpret_cast<const uint32_t*>(at.ptr)[pix];
    return at.format == FORMAT_BGRA_U8 ? swap_rb(v) : v;
}
RH_DEVICE void store_unorm(const rh_attach_dev& at, size_t pix, uint32_t v) {
    if (at.format == FORMAT_R_U8) { at.ptr[pix] = (uint8_t)v; return; }
    reinterpret_cast<uint32_t*>(at.ptr)[pix] = at.format == FORMAT_BGRA_U8 ? swap_rb(v) : v;
}

RH_DEVICE uint32_t load_pixel(const rh_attach_dev& at, size_t pix, uint32_t*) { return load_unorm(at, pix); }
RH_DEVICE uint4 load_pixel(const rh_attach_dev& at, size_t pix, uint4*) {
    if (at.format == FORMAT_RGBA_F32) return reinterpret_cast<const uint4*>(at.ptr)[pix];
    if (at.format == FORMAT_RGBA_F16) {
        const _Float16* h = reinterpret_cast<const _Float16*>(at.ptr) + pix * 4;
        return make_uint4(__float_as_uint((float)h[0]), __float_as_uint((float)h[1]), __float_as_uint((float)h[2]), __float_as_uint((float)h[3]));
    }
    return make_uint4(load_unorm(at, pix), 0u, 0u, 0u);
}
RH_DEVICE uint32_t clear_pixel(const rh_attach_dev& at, uint32_t*) { return at.clear_rgba8; }
RH_DEVICE uint4 clear_pixel(const rh_attach_dev& at, uint4*) {
    if (!is_float_format(at.format)) return make_uint4(at.clear_rgba8, 0u, 0u, 0u);
    const bool h = at.format == FORMAT_RGBA_F16;
    return make_uint4(__float_as_uint(h ? round_to_f16(at.clear_f[0]) : at.clear_f[0]), __float_as_uint(h ? round_to_f16(at.clear_f[1]) : at.clear_f[1]),
                      __float_as_uint(h ? round_to_f16(at.clear_f[2]) : at.clear_f[2]), __float_as_uint(h ? round_to_f16(at.clear_f[3]) : at.clear_f[3]));
}
RH_DEVICE void store_pixel(const rh_attach_dev& at, size_t pix, uint32_t v) { store_unorm(at, pix, v); }
RH_DEVICE void store_pixel(const rh_attach_dev& at, size_t pix, uint4 v) {
    if (at.format == FORMAT_RGBA_F32) { reinterpret_cast<uint4*>(at.ptr)[pix] = v; return; }
    if (at.format == FORMAT_RGBA_F16) {
        _Float16* h = reinterpret_cast<_Float16*>(at.ptr) + pix * 4;
        h[0] = (_Float16)__uint_as_float(v.x); h[1] = (_Float16)__uint_as_float(v.y); h[2] = (_Float16)__uint_as_float(v.z); h[3] = (_Float16)__uint_as_float(v.w);
        return;
    }
    store_unorm(at, pix, v.x);
}
// Float attachments blend in f32 without clamping: out = fma(dst, DF, src*SF) per channel; an RGBA16F result is rounded
// to f16 (to nearest even) after every fragment, like the 8-bit re-quantisation.
template <int BLEND>
RH_DEVICE void blend_pixel(uint32_t* px, const rh_attach_dev&, float4 s, uint32_t src_factor, uint32_t dst_factor) {
    *px = blend<BLEND>(*px, s, src_factor, dst_factor);
}
template <int BLEND>
RH_DEVICE void blend_pixel(uint4* px, const rh_attach_dev& at, float4 s, uint32_t src_factor, uint32_t dst_factor) {
    if (!is_float_format(at.format)) { px->x = blend<BLEND>(px->x, s, src_factor, dst_factor); return; }
    const float sf = blend_factor(src_factor, s.w), df = blend_factor(dst_factor, s.w);
    const uint4 d = *px;
    float o0 = __builtin_fmaf(__uint_as_float(d.x), df, s.x * sf), o1 = __builtin_fmaf(__uint_as_float(d.y), df, s.y * sf);
    float o2 = __builtin_fmaf(__uint_as_float(d.z), df, s.z * sf), o3 = __builtin_fmaf(__uint_as_float(d.w), df, s.w * sf);
    if (at.format == FORMAT_RGBA_F16) { o0 = round_to_f16(o0); o1 = round_to_f16(o1); o2 = round_to_f16(o2); o3 = round_to_f16(o3); }
    *px = make_uint4(__float_as_uint(o0), __float_as_uint(o1), __float_as_uint(o2), __float_as_uint(o3));
}

// The tile's part of a captured region (rh_attach_dev::cap_ptr): the texels as the attachment's memory holds them, and,
// from the tile that holds the region's last column, the zeroed bytes behind each row.
RH_DEVICE void capture_tile(const rh_attach_dev& at, const uint32_t* px, int ox, int oy, int lane, int r0 = 0, int rows = TILE) {
    for (int i = r0 * TILE + lane; i < (r0 + rows) * TILE; i += 64) {
        const uint32_t cx = (uint32_t)(ox + (i & (TILE - 1))) - at.cap_x0, cy = (uint32_t)(oy + i / TILE) - at.cap_y0;
        if (cx < at.cap_w && cy < at.cap_h) {
            const uint32_t v = px[i];
            *reinterpret_cast<uint32_t*>(at.cap_ptr + (size_t)cy * at.cap_pitch + (size_t)cx * 4u) = at.format == FORMAT_BGRA_U8 ? swap_rb(v) : v;
        }
    }
    const uint32_t last = at.cap_x0 + at.cap_w - 1u, pad = (at.cap_pitch - at.cap_w * 4u) / 4u;
    if (last - (uint32_t)ox < (uint32_t)TILE && pad != 0u)
        for (int r = r0; r < r0 + rows; r++) {
            const uint32_t cy = (uint32_t)(oy + r) - at.cap_y0;
            if (cy < at.cap_h)
                for (uint32_t k = (uint32_t)lane; k < pad; k += 64u) *reinterpret_cast<uint32_t*>(at.cap_ptr + (size_t)cy * at.cap_pitch + (size_t)(at.cap_w + k) * 4u) = 0u;
        }
}
RH_DEVICE void capture_tile(const rh_attach_dev&, const uint4*, int, int, int, int, int) {}        // (wide kernels: never asked to)

// Which tile a workgroup takes, and which band of its rows (rh_tile_args::split2_first): wave-uniform.
// Workgroups go to the eight XCDs in turn, each with an L2 of its own, and the bands of a tile read the same records:
// within the tiles cut into F bands, eight consecutive tiles form a group whose 8 * F workgroups are numbered band by
// band, so that the bands of a tile are eight workgroups apart -- on the same XCD, one after the other.
template <uint32_t F>
RH_DEVICE uint32_t band_of_workgroup(uint32_t job, uint32_t n_tiles, int& r0, int& rows) {
    const uint32_t group = job / (8u * F), r = job - group * (8u * F);
    const uint32_t left = n_tiles - group * 8u, m = left < 8u ? left : 8u;       // tiles of this group (the last may hold fewer)
    const uint32_t band = r / m;
    if (F == 3u) { r0 = (int)band * 11; rows = band < 2u ? 11 : 10; }       // thirds: 11 + 11 + 10 rows
    else { rows = TILE / (int)F; r0 = (int)band * rows; }
    return group * 8u + (r - band * m);
}
RH_DEVICE uint32_t tile_of_workgroup(const rh_tile_args& a, int& r0, int& rows) {
    uint32_t job = blockIdx.x;
    r0 = 0; rows = TILE;
    if (job < a.split2_first) return job;
    job -= a.split2_first;
    const uint32_t halved = a.split3_first - a.split2_first;
    if (job < halved * 2u) return a.split2_first + band_of_workgroup<2>(job, halved, r0, rows);
    job -= halved * 2u;
    const uint32_t in_thirds = a.split4_first - a.split3_first;
    if (job < in_thirds * 3u) return a.split3_first + band_of_workgroup<3>(job, in_thirds, r0, rows);
    job -= in_thirds * 3u;
    const uint32_t quartered = a.split8_first - a.split4_first;
    if (job < quartered * 4u) return a.split4_first + band_of_workgroup<4>(job, quartered, r0, rows);
    job -= quartered * 4u;
    const uint32_t in_eighths = (gridDim.x - (blockIdx.x - job)) >> 3;          // the rest of the grid is theirs
    return a.split8_first + band_of_workgroup<8>(job, in_eighths, r0, rows);
}

// The LDS of a tile wave.
template <int NA, bool WIDE = false>
struct TileLds {
    alignas(16) uint32_t rec[ROUND * REC_WORDS];     // first: record fields are reached with small immediate offsets
    typename Px<WIDE>::T px[NA][TILE_PX];
#ifdef RH_LDS_PAD                                    /* diagnostic build: fewer workgroups per CU */
    uint32_t pad[RH_LDS_PAD];
#endif
};

// Brings records [first, first + n) into the staging area (16-byte pieces, consecutive lanes consecutive pieces).
RH_DEVICE void stage_records(uint32_t* lds_rec, const uint8_t* rec, uint32_t first, int n, int lane) {
    const uint4* const src = reinterpret_cast<const uint4*>(rec + (size_t)first * (REC_WORDS * 4));
    uint4* const dst = reinterpret_cast<uint4*>(lds_rec);
    for (int i = lane; i < n * (REC_WORDS / 4); i += 64) dst[i] = src[i];
    // one wave: its LDS operations complete in issue order, so the reads that follow see these writes
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <int NA, int BLEND, int TEX, bool WIDE = false>
RH_DEVICE void tile_kernel(const rh_tile_args& a, TileLds<NA, WIDE>& lds) {
    using PxT = typename Px<WIDE>::T;
    if (a.pair_total && *a.pair_total > a.pair_capacity) return;     // the pair arrays were too small: the host runs the pass again
    const rh_draw_params& p = a.p;
    const int lane = (int)threadIdx.x;
    RH_PHASE_BEGIN
    int r0, rows;
    const uint32_t tile = tile_of_workgroup(a, r0, rows) + p.tile_first;
    const int ox = (int)(tile % p.tiles_x) * TILE, oy = (int)(tile / p.tiles_x) * TILE;
    const uint32_t start = a.tile_start[tile], end = a.tile_end[tile];
    const int px_first = r0 * TILE, px_end = (r0 + rows) * TILE;

    // 1. bring the tile into LDS
#pragma unroll
    for (int t = 0; t < NA; t++) {
        const rh_attach_dev& at = a.att[t];
        if (at.load_op == 0u && start == end && !at.cap_ptr) continue;          // untouched tile of a Load attachment
        for (int i = px_first + lane; i < px_end; i += 64) {
            const uint32_t gx = (uint32_t)(ox + (i & (TILE - 1))), gy = (uint32_t)(oy + i / TILE);
            PxT v{};
            if (gx < p.width && gy < p.height)
                v = at.load_op != 0u ? clear_pixel(at, (PxT*)nullptr) : load_pixel(at, (size_t)gy * at.pitch_px + gx, (PxT*)nullptr);
            lds.px[t][i] = v;
        }
    }
    RH_PHASE_MARK(0)

    FragmentContext fctx; fctx.p = &p;
    fctx.static_rgba8 = TEX != 0 ? 1u : 0u;
    fctx.static_filter = (TEX == 1 || TEX == 3) ? 1u + FILTER_LINEAR : (TEX == 2 ? 1u + FILTER_NEAREST : 0u);
    fctx.static_poly = TEX == 3 ? 1u : 0u;
    PxT* const tile_px = &lds.px[0][0];
    const float lane_f = (float)lane;
    const float* const recf = reinterpret_cast<const float*>(lds.rec);

    // 2. rounds of records
    int picked_grad = -1000;                            // the gradient hint the footprint copies were last picked for
    // Which form of footprint items the primitive at hand reads (PolyCopy::wide_for): one choice for all its texture
    // slots, going by the width of the first of them (the host says which); the pass loops exist once for either form.
    bool wide_items = true;
    const uint32_t poly_log2w = a.poly_log2w; const bool poly_pow2 = a.poly_pow2 != 0u;      // (the width of the pass's first texture with copies)
    // (the values of a round's pairs are fetched a round ahead: the set-up below starts with them)
    static_assert(2 * ROUND <= 64, "two lanes per pair of a round");
    const int pl = lane & (ROUND - 1);                  // the pair of the round this lane works on while the round is set up ...
    const bool sets_up = lane < 2 * ROUND;              // ... if it does
    uint32_t value_next = (sets_up && start < end && pl < (int)min((uint32_t)ROUND, end - start)) ? a.vals[start + (uint32_t)pl] : PAIR_GENERAL;
    for (uint32_t base = start; base < end; base += ROUND) {
        const int n = (int)min((uint32_t)ROUND, end - base);
        if (a.prio_quantum) {
            // Issue priority by the work the wave has left (pairs x rows, in quanta the host derives from the pass's
            // average tile): the SIMD picks the wave of highest priority, the oldest among equals, so without this the
            // waves of a SIMD end one after the other and the pass drains through SIMDs that are left with one starved
            // wave; with it the waves with most left go first and those of a SIMD end closer together.
            const uint32_t work = (end - base) * (uint32_t)rows, q = a.prio_quantum;
            if (work >= 3u * q) __builtin_amdgcn_s_setprio(3); else if (work >= 2u * q) __builtin_amdgcn_s_setprio(2);
            else if (work >= q) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        {
            // lane = pair of this round.  General pairs bring the record the set-up launch wrote; a lean quad is set up
            // here and now, straight into the staging area.
            // (lanes l and l + ROUND hold the same pair: they set a lean quad up together, see setup_lean_pair)
            const uint32_t value = value_next;
            const uint32_t nb = base + ROUND;
            value_next = (sets_up && nb < end && pl < (int)min((uint32_t)ROUND, end - nb)) ? a.vals[nb + (uint32_t)pl] : PAIR_GENERAL;
            const bool lean = (value & PAIR_GENERAL) == 0u;
            if (!__any(lean)) stage_records(lds.rec, a.rec, base, n, lane);
            else {
                uint32_t* const w = &lds.rec[pl * REC_WORDS];
                if (lean) setup_lean_pair(p, value & CLIP_CLUSTER_MASK, ox, oy, r0, rows, w, lane);
                else if (lane < n) {
                    const uint4* const src = reinterpret_cast<const uint4*>(a.rec + (size_t)(base + (uint32_t)lane) * (REC_WORDS * 4));
#pragma unroll
                    for (int i = 0; i < REC_WORDS / 4; i++) reinterpret_cast<uint4*>(w)[i] = src[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        RH_PHASE_MARK(1)
        // lane = record of this round (all wave-parallel)
        const uint32_t hdr = lane < n ? lds.rec[lane * REC_WORDS] : HDR_EMPTY;
        const uint32_t bb0 = hdr_box(hdr), fl = hdr_flags(hdr);
        // the box, cut to the rows of this workgroup's band
        const int hx0 = (int)(bb0 & 0xffu), hy0 = max((int)((bb0 >> 8) & 0xffu), r0), hx1 = (int)((bb0 >> 16) & 0xffu), hy1 = min((int)(bb0 >> 24), r0 + rows - 1);
        const uint32_t bb = (bb0 & 0x00ff00ffu) | ((uint32_t)hy0 << 8) | ((uint32_t)hy1 << 24);
        const bool hit = hx0 <= hx1 && hy0 <= hy1;
        const int npx = (hx1 - hx0 + 1) * (hy1 - hy0 + 1);
        const unsigned long long mask = __ballot(hit);
        const int nh = __popcll(mask);
        if (nh == 0) continue;
#if defined(RH_EXP) && RH_EXP == 5          /* diagnostic build: records staged, no hit is shaded */
        if (nh < 1000) continue;
#endif
        // Small axis-aligned quads may share one pass (see below); anything else runs alone, 64 pixels per pass.
        const bool lean_hit = (fl & (FLAG_RECT | FLAG_PERSP_A | FLAG_PERSP_B)) == FLAG_RECT;
        const bool small = lean_hit && npx <= 64;
        // Compact the hits, in submission order, into the first nh lanes (a lane permutation, no memory):
        //   box = the box, info = record of the round | flags << 6 | gradient hint (6 bits) << 12 | pixels << 18 |
        //   lean << 29 | (gradient hint differs from the hit's before: added below) << 30 | small << 31
        constexpr uint32_t INFO_LEAN = 1u << 29, INFO_REPICK = 1u << 30, INFO_SMALL = 1u << 31;
        uint32_t box_v, info_v;
        {
            const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const int dest = hit ? below : nh + lane - below;
            // (the hint, with what makes no difference to the copies taken out: beyond the last copy and below the first
            // one every distance reads the same copy, in the 32-byte form -- so a scene of tiny quads of many sizes, or
            // of magnified ones, does not pick anew hit after hit)
            uint32_t hint6 = hdr >> 26;
            if (TEX == 3) {
                const int k = ((int)hdr >> 27) + (int)poly_log2w;
                if (k >= RH_POLY_COPIES - 1) hint6 = (((uint32_t)(RH_POLY_COPIES - 1 - (int)poly_log2w) << 1) | 1u) & 63u;
                else if (k <= 0) hint6 = ((uint32_t)(-(int)poly_log2w) << 1) & 63u;
            }
            const uint32_t info = (uint32_t)lane | (fl << 6) | (hint6 << 12) | ((uint32_t)npx << 18) | (lean_hit ? INFO_LEAN : 0u) | (small ? INFO_SMALL : 0u);
            box_v = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)bb);
            info_v = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)(hit ? info : 0u));
        }
        // What the passes of a hit need as wave-uniform numbers is worked out here for all the hits at once, each in the
        // lane of its hit, and fetched by the scalar side hit by hit (the SIMD issues one instruction of ANY kind per ~2.3
        // cycles, scalar ones included: what is done once per round for 32 hits is nearly free, what is done per hit is not):
        //   boxw = x0 | y0 << 8 | width << 16 | record of the round << 24, invw = 1 / width, last = pixels - 1
        uint32_t boxw_v; float invw_v, last_v;
        {
            const uint32_t bw = ((box_v >> 16) & 0xffu) - (box_v & 0xffu) + 1u;
            boxw_v = (box_v & 0xffffu) | (bw << 16) | ((info_v & 0x3fu) << 24);
            invw_v = __builtin_amdgcn_rcpf((float)bw);
            last_v = (float)((info_v >> 18) & 0x7ffu) - 1.0f;
            if (TEX == 3) {
                // (the footprint copies are picked again only where the gradient hint changes from one hit to the next)
                const uint32_t before = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane - 1) & 63) << 2, (int)info_v);
                const int mine = (int)(info_v << 14) >> 26, prev = lane == 0 ? picked_grad : (int)(before << 14) >> 26;
                if (mine != prev) info_v |= INFO_REPICK;
                picked_grad = (int)(__builtin_amdgcn_readlane(info_v, nh - 1) << 14) >> 26;        // what the round leaves picked
            }
        }
        // A pack = up to PACK_MAX consecutive small hits whose boxes are pairwise disjoint and whose pixels fit
        // the 64 lanes of ONE pass (disjoint boxes cannot touch the same pixel, so their relative order is
        // immaterial).  Every lane works out the pack that would start at its hit:
        //   run = pixels | lane at which member 1, 2, 3 starts << 8, 16, 24 (127 = no such member)
        // (the footprint copies a pack reads are the ones picked for its first member or before: a hint, any copy holds the same texels)
        uint32_t run_v;
        {
            const uint32_t n0 = (info_v >> 18) & 0x7ffu;
            uint32_t len = 1u, total = n0, c1 = 127u, c2 = 127u, c3 = 127u;
            const unsigned long long sm = __ballot((info_v >> 31) != 0u);
            if (sm & (sm >> 1)) {
                static_assert(PACK_MAX == 4, "written for four members");
                // the three hits after this one (lanes past the end wrap around: excluded by the bound checks)
                auto after = [&](int d) { const int src = ((lane + d) & 63) << 2;
                                          return make_uint2((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)box_v), (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)info_v)); };
                const uint2 m1 = after(1), m2 = after(2), m3 = after(3);
                auto apart = [](uint32_t p, uint32_t q) {
                    return (p & 0xffu) > ((q >> 16) & 0xffu) || ((p >> 16) & 0xffu) < (q & 0xffu) ||
                           ((p >> 8) & 0xffu) > (q >> 24) || (p >> 24) < ((q >> 8) & 0xffu);
                };
                auto joins = [](uint32_t inf) { return (inf & INFO_SMALL) != 0u; };
                const uint32_t n1 = (m1.y >> 18) & 0x7ffu, n2 = (m2.y >> 18) & 0x7ffu, n3 = (m3.y >> 18) & 0x7ffu;
                const bool ok1 = (info_v >> 31) != 0u && lane + 1 < nh && joins(m1.y) && n0 + n1 <= 64u && apart(box_v, m1.x);
                const bool ok2 = ok1 && lane + 2 < nh && joins(m2.y) && n0 + n1 + n2 <= 64u && apart(box_v, m2.x) && apart(m1.x, m2.x);
                const bool ok3 = ok2 && lane + 3 < nh && joins(m3.y) && n0 + n1 + n2 + n3 <= 64u &&
                                 apart(box_v, m3.x) && apart(m1.x, m3.x) && apart(m2.x, m3.x);
                len = 1u + (ok1 ? 1u : 0u) + (ok2 ? 1u : 0u) + (ok3 ? 1u : 0u);
                c1 = ok1 ? n0 : 127u; c2 = ok2 ? n0 + n1 : 127u; c3 = ok3 ? n0 + n1 + n2 : 127u;
                total = n0 + (ok1 ? n1 : 0u) + (ok2 ? n2 : 0u) + (ok3 ? n3 : 0u);
            }
            run_v = total | (c1 << 8) | (c2 << 16) | (c3 << 24);
            // (a small hit that nobody joins runs alone like any other lean quad: fewer instructions than a pack of one;
            // from here on the bit says "a pack of two or more starts at this hit")
            if (len == 1u) info_v &= ~INFO_SMALL;
        }
        // The passes of an axis-aligned rectangle with affine varyings (what the reference's quads are): every pixel of
        // the box is covered and the shared diagonal alone says by which triangle.  The loop is kept free of per-pass
        // tests and of lane masks: a lane past the last pixel shades the last pixel once more -- same inputs, same result,
        // the same bytes written to the same place.  The arguments are wave-uniform for a hit that runs alone, per lane
        // for a pack.
        auto lean_passes = [&](auto wide_tag, const float hx0f, const float hy0f, const float wf, const float invw, float idx_f, const float last_f, const int npass, const float* const rf) {
            // (this copy of the loop reads one form of items: known to the compiler through the context's static_wide)
            FragmentContext fc = fctx;
            if (TEX == 3) fc.static_wide = decltype(wide_tag)::value ? 1u : 2u;
            const float half_invw = 0.5f * invw;
            const float4 dh = *reinterpret_cast<const float4*>(rf);                     // header, diagonal A, B, T
            const float3 dg = make_float3(dh.y, dh.z, dh.w);
            uint8_t* const px_bytes = reinterpret_cast<uint8_t*>(tile_px);
            // The planes of the triangle a pixel falls into are read through a per-lane LDS address: triangle B's sit
            // TRI_WORDS behind triangle A's, and B it is where the diagonal's edge function is negative.  (Sign bit times
            // the distance plus the base, in two instructions; the compiler's own sequence takes three.)
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            typedef const f32x4 __attribute__((address_space(3))) lds_f32x4_t;
            uint32_t planes_a = (uint32_t)(uintptr_t)(lds_f32x4_t*)(rf + HDR_WORDS);
            asm("" : "+v"(planes_a));               // (in a vector register once per hit, not once per pass)
            // One pass; LAST: the hit's last pass, whose lanes past the last pixel shade the last pixel again.
            auto one_pass = [&](auto last_tag, const float idx_f) {
                // (both are small non-negative integers: the minimum of their bit patterns)
                const float idc = decltype(last_tag)::value ? __uint_as_float(min(__float_as_uint(idx_f), __float_as_uint(last_f))) : idx_f;
                // floor((idx + 0.5) / w): the half keeps the quotient at least 1/128 away from an integer
                const float row_f = floorf(__builtin_fmaf(idc, invw, half_invw));
                const float col_f = __builtin_fmaf(-row_f, wf, idc);
                const float lxf = hx0f + col_f, lyf = hy0f + row_f;          // tile-relative pixel, exact small integers
                const uint32_t pb = (uint32_t)__builtin_fmaf(lyf, (float)(TILE * 4), lxf * 4.0f);
                const float e = __builtin_fmaf(dg.x, lxf, __builtin_fmaf(dg.y, lyf, dg.z));
                uint32_t at;
                if (TRI_WORDS * 4 <= 64) asm("v_lshrrev_b32 %0, 31, %1\n\tv_mad_u32_u24 %0, %0, %3, %2" : "=&v"(at) : "v"(e), "v"(planes_a), "n"(TRI_WORDS * 4));
                else at = planes_a + (__float_as_uint(e) >> 31) * (uint32_t)(TRI_WORDS * 4);
                float tp[TRI_WORDS];                  // (16-byte reads; what is not used is not read)
#pragma unroll
                for (int i = 0; i < TRI_WORDS / 4; i++) {
                    const f32x4 q = ((lds_f32x4_t*)(uintptr_t)at)[i];
                    tp[4 * i] = q.x; tp[4 * i + 1] = q.y; tp[4 * i + 2] = q.z; tp[4 * i + 3] = q.w;
                }
                const float dx = __builtin_fmaf(lxf, 256.0f, -tp[OFF_XR]), dy = __builtin_fmaf(lyf, 256.0f, -tp[OFF_YR]);   // exact
                Varyings var; float* vf = reinterpret_cast<float*>(&var);
#pragma unroll
                for (int v = 0; v < NV; v++) vf[v] = __builtin_fmaf(tp[OFF_DVDY + v], dy, __builtin_fmaf(tp[OFF_DVDX + v], dx, tp[OFF_V0 + v]));
                Outputs out;
                fragment_main(fc, var, out);
#pragma unroll
                for (int t = 0; t < NA; t++) {
                    blend_pixel<BLEND>(reinterpret_cast<PxT*>(px_bytes + (size_t)t * (TILE_PX * sizeof(PxT)) + (size_t)pb * (sizeof(PxT) / 4)), a.att[t],
                                       out.color[t < RH_FRAGMENT_OUTPUTS ? t : RH_FRAGMENT_OUTPUTS - 1], a.src_factor, a.dst_factor);
                }
            };
#if defined(RH_EXP) && RH_EXP == 4          /* diagnostic build: the hits are walked, no pass runs */
            if (dg.x == 12345.0f)
#endif
            {
                for (int pass = npass; pass > 1; pass--, idx_f += 64.0f) one_pass(Flag<false>{}, idx_f);      // (a hit has at least one pass)
                one_pass(Flag<true>{}, idx_f);
            }
        };
        // Hits in order; k and everything read with readlane is wave-uniform and lives on the scalar unit.
        // (nh >= 1; the three bodies are the arms of one if / else chain, so that every hit crosses few branches)
        // The loop exists twice: a round none of whose hits changes the footprint copies -- nearly every round -- runs a
        // loop that does not know of them, so that what was picked is not carried around the loop hit by hit.
        // the footprint copy of every texture slot for the primitive of a hit (uniform; unused slots fold away)
        auto repick = [&](const uint32_t info) {
            const int hint = (int)(info << 14) >> 26;
            wide_items = PolyCopy::wide_for(hint >> 1, (hint & 1) != 0, poly_log2w, poly_pow2);
#pragma unroll
            for (int i = 0; i < RH_DEV_MAX_SLOTS; i++) fctx.poly[i].pick(p.slots[i], hint >> 1, wide_items);
        };
        // ... and, where nothing is picked anew, once for either form of footprint items, so that the loop and the passes
        // it runs know the form at compile time (ITEMS: 1 the 32-byte items, 2 the 16-byte ones, 0 looked up as they go).
        auto hit_loop = [&](auto repicks_tag, auto wide_tag) {
        constexpr bool REPICKS = decltype(repicks_tag)::value;
        constexpr int ITEMS = decltype(wide_tag)::value;
        auto run_lean = [&](const float x0, const float y0, const float w, const float invw, const float idx, const float last, const int np, const float* const rfp) {
            if (ITEMS == 1) lean_passes(Flag<true>{}, x0, y0, w, invw, idx, last, np, rfp);
            else if (ITEMS == 2) lean_passes(Flag<false>{}, x0, y0, w, invw, idx, last, np, rfp);
            else if (wide_items) lean_passes(Flag<true>{}, x0, y0, w, invw, idx, last, np, rfp);
            else lean_passes(Flag<false>{}, x0, y0, w, invw, idx, last, np, rfp);
        };
        FragmentContext fg = fctx;                      // the general body's: the form known where the loop knows it
        if (TEX == 3 && ITEMS != 0) fg.static_wide = (uint32_t)ITEMS;
        FragmentContext& fgen = ITEMS != 0 ? fg : fctx;  // (the loop that picks anew reads what it picked)
        int k = 0;
        do {
            const uint32_t info = __builtin_amdgcn_readlane(info_v, k);
            if (__builtin_expect((info & (INFO_LEAN | (REPICKS ? INFO_REPICK : 0u) | INFO_SMALL)) == INFO_LEAN, 1)) {
                // the usual hit -- a lean quad alone, footprint copies as they are -- on a path of its own: one test
                const uint32_t boxw = __builtin_amdgcn_readlane(boxw_v, k);
                const float invw = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(invw_v), k));
                const float last_f = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(last_v), k));
                k++;
                const float bx = (float)(boxw & 0xffu), by = (float)((boxw >> 8) & 0xffu), bwf = (float)((boxw >> 16) & 0xffu);
                const int np = (int)((((info >> 18) & 0x7ffu) + 63u) >> 6);
                const float* const rfk = recf + (int)(info & 0x3fu) * REC_WORDS;
                run_lean(bx, by, bwf, invw, lane_f, last_f, np, rfk);
                continue;
            }
            if (REPICKS && (info & INFO_REPICK)) repick(info);
            if (info & INFO_SMALL) {
                // a pack: every lane takes the box and the record of the member its pixel belongs to.  Where the members
                // start is wave-uniform, so "lane >= start" is a mask the scalar unit makes with a shift, and the member
                // of a lane (times eight: where its start sits in `run`) takes three selects by those masks.
                const uint32_t run = __builtin_amdgcn_readlane(run_v, k);
                const uint32_t c1 = (run >> 8) & 127u, c2 = (run >> 16) & 127u, c3 = run >> 24;
                const unsigned long long m1 = c1 < 64u ? ~0ull << c1 : 0ull, m2 = c2 < 64u ? ~0ull << c2 : 0ull, m3 = c3 < 64u ? ~0ull << c3 : 0ull;
                uint32_t slot8;
                asm("v_cndmask_b32_e64 %0, 0, 8, %1" : "=v"(slot8) : "s"(m1));
                asm("v_cndmask_b32_e64 %0, %1, 16, %2" : "=v"(slot8) : "v"(slot8), "s"(m2));
                asm("v_cndmask_b32_e64 %0, %1, 24, %2" : "=v"(slot8) : "v"(slot8), "s"(m3));
                const uint32_t st = __builtin_amdgcn_ubfe(run & 0xffffff00u, slot8, 7u);      // (member 0 starts at lane 0)
                const int src = (k << 2) + (int)(slot8 >> 1);
                const uint32_t mb = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)boxw_v);
                const float minvw = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(invw_v)));
                // (the lanes past the pack's pixels shade its last pixel again)
                const float mx = (float)(mb & 0xffu), my = (float)((mb >> 8) & 0xffu), mw = (float)((mb >> 16) & 0xffu);
                const float midx = (float)(int)((uint32_t)lane - st), mlast = (float)(int)((run & 127u) - 1u - st);
                const float* const mrf = recf + (int)(mb >> 24) * REC_WORDS;
                run_lean(mx, my, mw, minvw, midx, mlast, 1, mrf);
                k += 1 + (int)(((~c1 >> 6) & 1u) + ((~c2 >> 6) & 1u) + ((~c3 >> 6) & 1u));         // (a start is below 64, or 127)
            } else {
            const uint32_t rec_index = info & 0x3fu;
            const float* const rf = recf + (int)rec_index * REC_WORDS;
            const uint32_t boxw = __builtin_amdgcn_readlane(boxw_v, k);
            const float invw = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(invw_v), k));
            const float last_f = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(last_v), k));
            const float hx0f = (float)(boxw & 0xffu), hy0f = (float)((boxw >> 8) & 0xffu), wf = (float)((boxw >> 16) & 0xffu);
            const int npass = (int)((((info >> 18) & 0x7ffu) + 63u) >> 6);
            k++;
            if (info & INFO_LEAN) run_lean(hx0f, hy0f, wf, invw, lane_f, last_f, npass, rf);
            else {
            const uint32_t flags = (info >> 6) & 0x3fu;
            const float half_invw = 0.5f * invw;
            float idx_f = lane_f;
            // (only hits that run alone come here: the extended record of this one)
            const float* const ef = reinterpret_cast<const float*>(a.ext + (size_t)(base + rec_index) * a.ext_bytes);
            const bool has_a = (flags & FLAG_HAS_A) != 0u, has_b = (flags & FLAG_HAS_B) != 0u;
            const bool any_persp = (flags & (FLAG_PERSP_A | FLAG_PERSP_B)) != 0u;
            for (int pass = 0; pass < npass; pass++, idx_f += 64.0f) {
                const bool live = idx_f <= last_f;
                // floor((idx + 0.5) / w): the half keeps the quotient at least 1/128 away from an integer
                const float row_f = floorf(__builtin_fmaf(idx_f, invw, half_invw));
                const float col_f = __builtin_fmaf(-row_f, wf, idx_f);
                const float lxf = hx0f + col_f, lyf = hy0f + row_f;          // tile-relative pixel, exact small integers
                const int pidx = (int)__builtin_fmaf(lyf, (float)TILE, lxf);
                bool cov_a, cov_b, overlap = false;
                if (flags & FLAG_RECT) {
                    // every pixel of the box is covered; the shared diagonal (edge 1 of A) says by which triangle
                    const bool in_a = __float_as_int(__builtin_fmaf(rf[1], lxf, __builtin_fmaf(rf[2], lyf, rf[3]))) >= 0;
                    cov_a = live && in_a; cov_b = live && !in_a;
                } else {
                    // (a missing triangle's record holds zeros: masked by has_*)
                    bool in_a, in_b;
                    if (flags & FLAG_BIG) {
                        const int lx = (int)lxf, ly = (int)lyf;
                        in_a = edges_inside_i32(ef, lx, ly); in_b = edges_inside_i32(ef + EXT_TRI_WORDS, lx, ly);
                    } else {
                        in_a = edges_inside_f32(ef, lxf, lyf); in_b = edges_inside_f32(ef + EXT_TRI_WORDS, lxf, lyf);
                    }
                    cov_a = live && has_a && in_a; cov_b = live && has_b && in_b;
                    overlap = __any(cov_a && cov_b);
                }
                // One shading pass when no pixel is covered by both triangles (the usual case): every lane takes
                // the plane equations of the triangle that covers it.  Otherwise A's fragments, then B's.
                auto shade = [&](bool cov, bool use_b) {
                    // per-lane choice of the triangle = per-lane LDS address (one copy of the shading code)
                    const float* const tp = rf + HDR_WORDS + (use_b ? TRI_WORDS : 0);
                    const float dx = __builtin_fmaf(lxf, 256.0f, -tp[OFF_XR]), dy = __builtin_fmaf(lyf, 256.0f, -tp[OFF_YR]);   // exact
                    Varyings var; float* vf = reinterpret_cast<float*>(&var);
#pragma unroll
                    for (int v = 0; v < NV; v++) vf[v] = __builtin_fmaf(tp[OFF_DVDY + v], dy, __builtin_fmaf(tp[OFF_DVDX + v], dx, tp[OFF_V0 + v]));
                    if (any_persp) {
                        const bool persp = use_b ? (flags & FLAG_PERSP_B) != 0u : (flags & FLAG_PERSP_A) != 0u;
                        const float* const eq = ef + (use_b ? EXT_TRI_WORDS : 0) + EOFF_Q;
                        const float q = __builtin_fmaf(eq[2], dy, __builtin_fmaf(eq[1], dx, eq[0]));
#pragma unroll
                        for (int v = 0; v < NV; v++) vf[v] = persp ? vf[v] / q : vf[v];
                    }
                    if (cov) {
                        Outputs out;
                        fragment_main(fgen, var, out);
#pragma unroll
                        for (int t = 0; t < NA; t++) {
                            blend_pixel<BLEND>(tile_px + t * TILE_PX + pidx, a.att[t],
                                               out.color[t < RH_FRAGMENT_OUTPUTS ? t : RH_FRAGMENT_OUTPUTS - 1], a.src_factor, a.dst_factor);
                        }
                    }
                };
                if (__builtin_expect(!overlap, 1)) shade(cov_a || cov_b, cov_b);
                else {
                    // overlapping triangles inside the cluster (bow-tie, folded quads): strictly A before B
#pragma nounroll
                    for (int phase = 0; phase < 2; phase++) shade(phase ? cov_b : cov_a, phase != 0);
                }
            }       // passes of the hit
            }       // general body
            }       // a hit alone
        } while (k < nh);
        };
        // (a change at the round's first hit only -- the first round of a tile, as a rule -- is made here, ahead of the loop)
        unsigned long long changes = TEX == 3 ? __ballot((info_v & INFO_REPICK) != 0u && lane < nh) : 0ull;
        if (changes & 1ull) { repick(__builtin_amdgcn_readlane(info_v, 0)); changes &= ~1ull; if (lane == 0) info_v &= ~INFO_REPICK; }
        if (changes) hit_loop(Flag<true>{}, Int<0>{});
        else if (TEX != 3) hit_loop(Flag<false>{}, Int<0>{});
        else if (wide_items) hit_loop(Flag<false>{}, Int<1>{});
        else hit_loop(Flag<false>{}, Int<2>{});
        RH_PHASE_MARK(2)
    }

    // 3. write the tile back
#pragma unroll
    for (int t = 0; t < NA; t++) {
        const rh_attach_dev& at = a.att[t];
        const bool untouched = at.load_op == 0u && start == end;
        if (untouched && !at.cap_ptr) continue;
        if (!untouched)
            for (int i = px_first + lane; i < px_end; i += 64) {
                const uint32_t gx = (uint32_t)(ox + (i & (TILE - 1))), gy = (uint32_t)(oy + i / TILE);
                if (gx < p.width && gy < p.height) store_pixel(at, (size_t)gy * at.pitch_px + gx, lds.px[t][i]);
            }
        if (!WIDE && at.cap_ptr) capture_tile(at, &lds.px[t][0], ox, oy, lane, r0, rows);
    }
    RH_PHASE_MARK(3)
    RH_PHASE_END
}

// ---- multisampled passes -----------------------------------------------------------------------------------------
// Same structure as tile_kernel (one wave per tile, LDS-resident tile, records in order), with S colour samples per
// pixel: coverage is evaluated per sample at the standard sample locations with the same integer edge functions and
// top-left rule, the fragment function runs once per pixel and triangle with its inputs interpolated at the pixel
// centre, and its colour is blended into every covered sample.  The tile's samples go back to the multisampled texture
// and their mean (round to nearest, ties to even; the sum is exact) to the resolve target.
// One attachment of any format (WIDE: RGBA16F / RGBA32F, 16-byte LDS pixels; the multisampled texture is shaped after
// the target, reference src/pipeline.rs:164-176); every state is read at run time.  A float resolve adds the samples in
// f32 in sample order and multiplies by 1/S; an RGBA16F mean is rounded to f16.
template <int S, bool WIDE>
struct TileLdsMs {
    alignas(16) uint32_t rec[ROUND * REC_WORDS];
    typename Px<WIDE>::T px[S][TILE_PX];   // one plane per sample index
};

template <int S, bool WIDE = false>
RH_DEVICE void tile_kernel_ms(const rh_tile_args& a) {
    using PxT = typename Px<WIDE>::T;
    __shared__ TileLdsMs<S, WIDE> lds;
    if (a.pair_total && *a.pair_total > a.pair_capacity) return;     // the pair arrays were too small: the host runs the pass again
    const rh_draw_params& p = a.p;
    const rh_attach_dev& at = a.att[0];
    const int lane = (int)threadIdx.x;
    int r0, rows;
    const uint32_t tile = tile_of_workgroup(a, r0, rows) + p.tile_first;
    const int ox = (int)(tile % p.tiles_x) * TILE, oy = (int)(tile / p.tiles_x) * TILE;
    const uint32_t start = a.tile_start[tile], end = a.tile_end[tile];
    const int px_first = r0 * TILE, px_end = (r0 + rows) * TILE;

    // 1. the tile's samples into LDS (every tile: the resolve below rewrites the whole target)
    for (int i = px_first + lane; i < px_end; i += 64) {
        const uint32_t gx = (uint32_t)(ox + (i & (TILE - 1))), gy = (uint32_t)(oy + i / TILE);
        const bool inside = gx < p.width && gy < p.height;
        const size_t first = ((size_t)gy * at.pitch_px + gx) * S;
#pragma unroll
        for (int sidx = 0; sidx < S; sidx++) {
            PxT v{};
            if (inside) v = at.load_op != 0u ? clear_pixel(at, (PxT*)nullptr) : load_pixel(at, first + (size_t)sidx, (PxT*)nullptr);
            lds.px[sidx][i] = v;
        }
    }

    FragmentContext fctx; fctx.p = &p; fctx.static_rgba8 = 0u; fctx.static_filter = 0u;
    const float lane_f = (float)lane;
    const float* const recf = reinterpret_cast<const float*>(lds.rec);

    // 2. rounds of records
    for (uint32_t base = start; base < end; base += ROUND) {
        const int n = (int)min((uint32_t)ROUND, end - base);
        stage_records(lds.rec, a.rec, base, n, lane);
        const uint32_t bb = hdr_box(lane < n ? lds.rec[lane * REC_WORDS] : HDR_EMPTY);
        unsigned long long mask = __ballot((bb & 0xffu) <= ((bb >> 16) & 0xffu));
        while (mask) {
            const int ci = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float* const rf = recf + ci * REC_WORDS;
            const uint32_t hdr = __builtin_amdgcn_readfirstlane(lds.rec[ci * REC_WORDS]);
            const uint32_t flags = hdr_flags(hdr), hb = hdr_box(hdr);
            const float* const ef = reinterpret_cast<const float*>(a.ext + (size_t)(base + (uint32_t)ci) * a.ext_bytes);
            const MsRecord<S>& m = *reinterpret_cast<const MsRecord<S>*>(ef + EXT_WORDS);
            const bool has_a = (flags & FLAG_HAS_A) != 0u, has_b = (flags & FLAG_HAS_B) != 0u;
            const bool any_persp = (flags & (FLAG_PERSP_A | FLAG_PERSP_B)) != 0u;
            // (the box cut to the rows of this workgroup's band: nothing left of it = no passes)
            const int hx0 = (int)(hb & 0xffu), hy0 = max((int)((hb >> 8) & 0xffu), r0), hx1 = (int)((hb >> 16) & 0xffu), hy1 = min((int)(hb >> 24), r0 + rows - 1);
            const int w = hx1 - hx0 + 1, npx = w * (hy1 - hy0 + 1);
            const float wf = (float)w, invw = __builtin_amdgcn_rcpf(wf);
            for (int first = 0; first < npx; first += 64) {
                const float idx_f = lane_f + (float)first;
                const float row_f = floorf(__builtin_fmaf(idx_f, invw, 0.5f * invw));
                const float col_f = __builtin_fmaf(-row_f, wf, idx_f);
                const float lxf = (float)hx0 + col_f, lyf = (float)hy0 + row_f;
                const bool live = idx_f < (float)npx;
                const int lx = (int)lxf, ly = (int)lyf;
                const int pidx = live ? ly * TILE + lx : 0;
                // sample masks of the two triangles (a missing triangle's constants are unset: masked by has_*)
                uint32_t mask_a = 0u, mask_b = 0u;
#pragma unroll
                for (int sidx = 0; sidx < S; sidx++) {
                    int ia = 0, ib = 0;
#pragma unroll
                    for (int e = 0; e < 3; e++) {
                        ia |= __mul24(m.A[0][e], lx) + __mul24(m.B[0][e], ly) + m.T[0][sidx][e];
                        ib |= __mul24(m.A[1][e], lx) + __mul24(m.B[1][e], ly) + m.T[1][sidx][e];
                    }
                    mask_a |= ia >= 0 ? 1u << sidx : 0u;
                    mask_b |= ib >= 0 ? 1u << sidx : 0u;
                }
                if (!(live && has_a)) mask_a = 0u;
                if (!(live && has_b)) mask_b = 0u;
                auto shade = [&](uint32_t cov, bool use_b) {
                    const float* const tp = rf + HDR_WORDS + (use_b ? TRI_WORDS : 0);
                    const float dx = __builtin_fmaf(lxf, 256.0f, -tp[OFF_XR]), dy = __builtin_fmaf(lyf, 256.0f, -tp[OFF_YR]);
                    Varyings var; float* vf = reinterpret_cast<float*>(&var);
#pragma unroll
                    for (int v = 0; v < NV; v++) vf[v] = __builtin_fmaf(tp[OFF_DVDY + v], dy, __builtin_fmaf(tp[OFF_DVDX + v], dx, tp[OFF_V0 + v]));
                    if (any_persp) {
                        const bool persp = use_b ? (flags & FLAG_PERSP_B) != 0u : (flags & FLAG_PERSP_A) != 0u;
                        const float* const eq = ef + (use_b ? EXT_TRI_WORDS : 0) + EOFF_Q;
                        const float q = __builtin_fmaf(eq[2], dy, __builtin_fmaf(eq[1], dx, eq[0]));
#pragma unroll
                        for (int v = 0; v < NV; v++) vf[v] = persp ? vf[v] / q : vf[v];
                    }
                    if (cov) {
                        Outputs out;
                        fragment_main(fctx, var, out);
#pragma unroll
                        for (int sidx = 0; sidx < S; sidx++)
                            if (cov & (1u << sidx)) blend_pixel<0>(&lds.px[sidx][pidx], at, out.color[0], a.src_factor, a.dst_factor);
                    }
                };
                // triangle A's fragment first; a pixel with samples in both triangles gets a second invocation for B
                const bool first_is_b = mask_a == 0u;
                shade(first_is_b ? mask_b : mask_a, first_is_b);
                const uint32_t second = mask_a != 0u ? mask_b : 0u;
                if (__any(second != 0u)) shade(second, true);
            }
        }
    }

    // 3. samples back to the multisampled texture, their mean to the resolve target
    const float inv_s = 1.0f / (float)S;
    rh_attach_dev rt = at;                   // the resolve target seen as an attachment
    rt.ptr = at.resolve_ptr; rt.pitch_px = at.resolve_pitch_px; rt.format = at.resolve_format;
    for (int i = px_first + lane; i < px_end; i += 64) {
        const uint32_t gx = (uint32_t)(ox + (i & (TILE - 1))), gy = (uint32_t)(oy + i / TILE);
        if (!(gx < p.width && gy < p.height)) continue;
        const size_t first = ((size_t)gy * at.pitch_px + gx) * S;
        const bool write_samples = at.load_op != 0u || start != end;
        if (WIDE) {
            float sum[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int sidx = 0; sidx < S; sidx++) {
                const PxT v = lds.px[sidx][i];
                const uint4 w = wide_words(v);
                if (sidx == 0) { sum[0] = __uint_as_float(w.x); sum[1] = __uint_as_float(w.y); sum[2] = __uint_as_float(w.z); sum[3] = __uint_as_float(w.w); }
                else { sum[0] = sum[0] + __uint_as_float(w.x); sum[1] = sum[1] + __uint_as_float(w.y); sum[2] = sum[2] + __uint_as_float(w.z); sum[3] = sum[3] + __uint_as_float(w.w); }
                if (write_samples) store_pixel(at, first + (size_t)sidx, v);
            }
            if (at.resolve_ptr)
                store_pixel(rt, (size_t)gy * rt.pitch_px + gx, wide_pixel(make_uint4(__float_as_uint(sum[0] * inv_s), __float_as_uint(sum[1] * inv_s),
                                                                                 __float_as_uint(sum[2] * inv_s), __float_as_uint(sum[3] * inv_s)), (PxT*)nullptr));
        } else {
            uint32_t sum[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
            for (int sidx = 0; sidx < S; sidx++) {
                const PxT pv = lds.px[sidx][i];
                const uint32_t v = wide_words(pv).x;
                sum[0] += v & 0xffu; sum[1] += (v >> 8) & 0xffu; sum[2] += (v >> 16) & 0xffu; sum[3] += v >> 24;
                if (write_samples) store_pixel(at, first + (size_t)sidx, pv);
            }
            if (at.resolve_ptr) {
                uint32_t q = 0u;
                q = __builtin_amdgcn_cvt_pk_u8_f32((float)sum[0] * inv_s, 0, q);
                q = __builtin_amdgcn_cvt_pk_u8_f32((float)sum[1] * inv_s, 1, q);
                q = __builtin_amdgcn_cvt_pk_u8_f32((float)sum[2] * inv_s, 2, q);
                q = __builtin_amdgcn_cvt_pk_u8_f32((float)sum[3] * inv_s, 3, q);
                store_pixel(rt, (size_t)gy * rt.pitch_px + gx, wide_pixel(make_uint4(q, 0u, 0u, 0u), (PxT*)nullptr));
            }
        }
    }
}

// Runs the specialisation rh_tile_args::variant names; one and two attachments (plain and recorded rendering) have the
// specialised bodies, more attachments use the general one.
template <int NA>
RH_DEVICE void tile_dispatch(const rh_tile_args& a) {
    __shared__ TileLds<NA> lds;
    if (NA <= 2) {
        switch (a.variant) {
            case 1: tile_kernel<NA, 1, 1>(a, lds); return;
            case 2: tile_kernel<NA, 1, 2>(a, lds); return;
            case 3: tile_kernel<NA, 1, 3>(a, lds); return;
            case 4: tile_kernel<NA, 2, 1>(a, lds); return;
            case 5: tile_kernel<NA, 2, 2>(a, lds); return;
            case 6: tile_kernel<NA, 2, 3>(a, lds); return;
            default: break;
        }
    }
    tile_kernel<NA, 0, 0>(a, lds);
}

// Passes with an RGBA16F / RGBA32F attachment: 16-byte LDS pixels, every state read at run time.
template <int NA>
RH_DEVICE void tile_dispatch_wide(const rh_tile_args& a) {
    __shared__ TileLds<NA, true> lds;
    tile_kernel<NA, 0, 0, true>(a, lds);
}

} // namespace rh

// Waves per SIMD the register allocation aims at: what the LDS of a tile workgroup admits anyway (one attachment: 6.5 KB,
// 24 workgroups per CU; two: 10.5 KB, 15; ...).
#ifndef RH_TILE_MIN_WAVES
#define RH_TILE_MIN_WAVES 6        // 80 registers per lane
#endif
extern "C" __global__ __launch_bounds__(256) void rh_setup(rh_setup_args a) { rh::setup_kernel<1>(a); }
extern "C" __global__ __launch_bounds__(256) void rh_setup_ms2(rh_setup_args a) { rh::setup_kernel<2>(a); }
extern "C" __global__ __launch_bounds__(256) void rh_setup_ms4(rh_setup_args a) { rh::setup_kernel<4>(a); }
extern "C" __global__ __launch_bounds__(256) void rh_setup_ms8(rh_setup_args a) { rh::setup_kernel<8>(a); }
extern "C" __global__ __launch_bounds__(256) void rh_setup_clip(rh_setup_args a) { rh::setup_kernel<1, true>(a); }
extern "C" __global__ __launch_bounds__(256) void rh_setup_clip_ms2(rh_setup_args a) { rh::setup_kernel<2, true>(a); }
extern "C" __global__ __launch_bounds__(256) void rh_setup_clip_ms4(rh_setup_args a) { rh::setup_kernel<4, true>(a); }
extern "C" __global__ __launch_bounds__(256) void rh_setup_clip_ms8(rh_setup_args a) { rh::setup_kernel<8, true>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS, RH_TILE_MIN_WAVES) void rh_tile_1(rh_tile_args a) { rh::tile_dispatch<1>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS, 4) void rh_tile_2(rh_tile_args a) { rh::tile_dispatch<2>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS, 3) void rh_tile_3(rh_tile_args a) { rh::tile_dispatch<3>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS, 2) void rh_tile_4(rh_tile_args a) { rh::tile_dispatch<4>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_ms2(rh_tile_args a) { rh::tile_kernel_ms<2>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_ms4(rh_tile_args a) { rh::tile_kernel_ms<4>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_ms8(rh_tile_args a) { rh::tile_kernel_ms<8>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_msw2(rh_tile_args a) { rh::tile_kernel_ms<2, true>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_msw4(rh_tile_args a) { rh::tile_kernel_ms<4, true>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_msw8(rh_tile_args a) { rh::tile_kernel_ms<8, true>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_w1(rh_tile_args a) { rh::tile_dispatch_wide<1>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_w2(rh_tile_args a) { rh::tile_dispatch_wide<2>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_w3(rh_tile_args a) { rh::tile_dispatch_wide<3>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_w4(rh_tile_args a) { rh::tile_dispatch_wide<4>(a); }
