// Fragment-stage code object: compiled once per (vertex source, fragment source) pair with
//     hipcc --genco --offload-arch=gfx950 -ffp-contract=off -DRH_VERTEX_SOURCE=... -DRH_FRAGMENT_SOURCE=... rh_stage_fragment.hip
// Exports rh_tile_<A> (A = 1..4 colour attachments): one workgroup per 32x32 screen tile.
//
// Per tile:
//   1. the tile's pixels are brought into LDS (cleared, or loaded with coalesced row reads), one u32 RGBA8 per pixel,
//      stored block-linear (8x8 blocks, padded so that both block-wise and row-wise accesses are conflict-free);
//   2. the tile's ordered cluster list is walked in batches of RH_BATCH_TRIS triangles: threads set triangles up
//      (vertex function, snap, integer edge equations with the top-left rule folded into the constant, varying
//      plane equations) into LDS records; then every wave walks its four 8x8 blocks: lanes test one triangle each
//      against the block (ballot), the set bits are visited in ascending order -- which is primitive order -- and
//      every lane (= pixel) evaluates coverage, shades, and blends into its register-held pixel, re-quantising to
//      UNORM8 after every fragment;
//   3. the tile is written back with coalesced row stores (RGBA or BGRA).
#include "rh_shader.h"
#include RH_VERTEX_SOURCE
#include RH_FRAGMENT_SOURCE
#include "rh_geometry.h"

#ifndef RH_FRAGMENT_OUTPUTS
#error "the fragment source must define RH_FRAGMENT_OUTPUTS (number of colour outputs it writes)"
#endif

extern "C" __device__ const rh_module_info_t rh_module_info = {
    RH_MODULE_MAGIC, RH_DEVICE_ABI_VERSION, RH_MODULE_STAGE_FRAGMENT, (uint32_t)rh::N_VARYINGS, (uint32_t)RH_FRAGMENT_OUTPUTS,
    rh::fnv1a(RH_VERTEX_TAG), { 0u, 0u }
};

namespace rh {

constexpr int NV = N_VARYINGS;
constexpr int BLOCKS_PER_TILE = (RH_TILE / RH_BLOCK) * (RH_TILE / RH_BLOCK);   // 16
constexpr int BLOCK_STRIDE = 72;          // 64 pixels + 8 dwords of padding
constexpr uint32_t BBOX_EMPTY = 0x000000ffu;   // x0 = 255 > x1 = 0
constexpr uint32_t FLAG_PERSP = 1u;

struct alignas(16) TriRecord {
    int32_t A0, B0, T0, A1;
    int32_t B1, T1, A2, B2;
    int32_t T2, X0t, Y0t; uint32_t flags;
    float q0, dqdx, dqdy, pad0;
    float v0[NV], dvdx[NV], dvdy[NV];
};

RH_DEVICE int lds_pixel_index(int lx, int ly) {
    return ((ly >> 3) * (RH_TILE / RH_BLOCK) + (lx >> 3)) * BLOCK_STRIDE + (ly & 7) * 8 + (lx & 7);
}

RH_DEVICE uint32_t swap_rb(uint32_t v) { return (v & 0xff00ff00u) | ((v & 0xffu) << 16) | ((v >> 16) & 0xffu); }

RH_DEVICE float blend_factor(uint32_t f, float src_alpha) {
    return f == 0u ? 0.0f : (f == 1u ? 1.0f : (f == 2u ? src_alpha : 1.0f - src_alpha));
}

RH_DEVICE uint32_t to_unorm8(float o) {
    o = fminf(fmaxf(o, 0.0f), 1.0f);
    return (uint32_t)rintf(o * 255.0f);
}

// out = src*SF + dst*DF for every channel, alpha included; the destination is the stored UNORM8 value and the result
// is re-quantised (round half to even) after every fragment.
RH_DEVICE uint32_t blend(uint32_t dst, float4 s, uint32_t src_factor, uint32_t dst_factor) {
    const float sf = blend_factor(src_factor, s.w), df = blend_factor(dst_factor, s.w);
    float d0 = (float)(dst & 0xffu) * INV_255, d1 = (float)((dst >> 8) & 0xffu) * INV_255;
    float d2 = (float)((dst >> 16) & 0xffu) * INV_255, d3 = (float)(dst >> 24) * INV_255;
    uint32_t q0 = to_unorm8(__builtin_fmaf(d0, df, s.x * sf));
    uint32_t q1 = to_unorm8(__builtin_fmaf(d1, df, s.y * sf));
    uint32_t q2 = to_unorm8(__builtin_fmaf(d2, df, s.z * sf));
    uint32_t q3 = to_unorm8(__builtin_fmaf(d3, df, s.w * sf));
    return q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
}

RH_DEVICE int32_t clamp_i64_to_i31(int64_t v) {
    const int64_t lim = (int64_t)1 << 30;
    return (int32_t)(v < -lim ? -lim : (v > lim - 1 ? lim - 1 : v));
}

// Triangle setup for tile (ox, oy).  Returns the packed tile-relative bbox, or BBOX_EMPTY when the triangle
// contributes nothing to this tile.
RH_DEVICE uint32_t setup_triangle(const rh_draw_params& p, uint32_t cluster, uint32_t which, int ox, int oy, TriRecord& r) {
    const uint32_t instance = cluster / p.cpi, local = cluster - instance * p.cpi;
    const uint32_t tri = local * 2 + which;
    if (tri >= p.tpi) return BBOX_EMPTY;
    uint32_t idx[3]; tri_vertices(p.topology, tri, idx);
    SnappedVertex a = run_vertex(p, idx[0], instance);
    SnappedVertex b = run_vertex(p, idx[1], instance);
    SnappedVertex c = run_vertex(p, idx[2], instance);
    int x0, y0, x1, y1;
    if (!tri_pixel_bbox(p, a, b, c, x0, y0, x1, y1)) return BBOX_EMPTY;
    x0 = max(x0, ox); y0 = max(y0, oy); x1 = min(x1, ox + RH_TILE - 1); y1 = min(y1, oy + RH_TILE - 1);
    if (x0 > x1 || y0 > y1) return BBOX_EMPTY;
    const int64_t area2 = (int64_t)(b.X - a.X) * (c.Y - a.Y) - (int64_t)(c.X - a.X) * (b.Y - a.Y);
    if (area2 == 0) return BBOX_EMPTY;
    if (area2 < 0) { SnappedVertex t = b; b = c; c = t; }      // no culling: both windings are drawn
    // Edge e from P to Q: inside <=> A*i + B*j + Cq >= 0 at pixel (i, j), with the top-left rule folded into Cq.
    // Stored relative to the tile origin and saturated: |A*lx + B*ly| < 2^29 for lx, ly < 32, so the sign survives.
#define RH_EDGE(P, Q, AA, BB, TT) { \
        const int32_t A_ = P.Y - Q.Y, B_ = Q.X - P.X; \
        const int64_t C_ = -((int64_t)A_ * P.X + (int64_t)B_ * P.Y); \
        const bool tl_ = (A_ > 0) || (A_ == 0 && B_ > 0); \
        const int64_t Cq_ = (C_ + (tl_ ? 0 : -1)) >> RH_SUBPIXEL_BITS; \
        AA = A_; BB = B_; TT = clamp_i64_to_i31(Cq_ + (int64_t)A_ * ox + (int64_t)B_ * oy); }
    RH_EDGE(a, b, r.A0, r.B0, r.T0)
    RH_EDGE(b, c, r.A1, r.B1, r.T1)
    RH_EDGE(c, a, r.A2, r.B2, r.T2)
#undef RH_EDGE
    r.X0t = a.X - ox * 256; r.Y0t = a.Y - oy * 256;
    const float k = 1.0f / 256.0f;
    const float e1x = (float)(b.X - a.X) * k, e1y = (float)(b.Y - a.Y) * k;
    const float e2x = (float)(c.X - a.X) * k, e2y = (float)(c.Y - a.Y) * k;
    const float area = __builtin_fmaf(e1x, e2y, -(e2x * e1y));
    const float inv = 1.0f / area;
    const bool persp = !(a.rw == 1.0f && b.rw == 1.0f && c.rw == 1.0f);
    r.flags = persp ? FLAG_PERSP : 0u;
    const float* fa = reinterpret_cast<const float*>(&a.var);
    const float* fb = reinterpret_cast<const float*>(&b.var);
    const float* fc = reinterpret_cast<const float*>(&c.var);
#define RH_PLANE(F0, F1, F2, O0, OX, OY) { \
        const float f0_ = (F0), d1_ = (F1) - f0_, d2_ = (F2) - f0_; \
        O0 = f0_; \
        OX = __builtin_fmaf(d1_, e2y, -(d2_ * e1y)) * inv; \
        OY = __builtin_fmaf(d2_, e1x, -(d1_ * e2x)) * inv; }
#pragma unroll
    for (int v = 0; v < NV; v++) {
        if (persp) RH_PLANE(fa[v] * a.rw, fb[v] * b.rw, fc[v] * c.rw, r.v0[v], r.dvdx[v], r.dvdy[v])
        else RH_PLANE(fa[v], fb[v], fc[v], r.v0[v], r.dvdx[v], r.dvdy[v])
    }
    RH_PLANE(a.rw, b.rw, c.rw, r.q0, r.dqdx, r.dqdy)
#undef RH_PLANE
    r.pad0 = 0.0f;
    return (uint32_t)(x0 - ox) | ((uint32_t)(y0 - oy) << 8) | ((uint32_t)(x1 - ox) << 16) | ((uint32_t)(y1 - oy) << 24);
}

template <int NA>
RH_DEVICE void tile_kernel(const rh_tile_args& a) {
    __shared__ TriRecord rec[RH_BATCH_TRIS];
    __shared__ uint32_t rec_bbox[RH_BATCH_TRIS];
    __shared__ uint32_t tile_px[NA][BLOCKS_PER_TILE * BLOCK_STRIDE];

    const rh_draw_params& p = a.p;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t tile = blockIdx.x;
    const int ox = (int)(tile % p.tiles_x) * RH_TILE, oy = (int)(tile / p.tiles_x) * RH_TILE;
    const uint32_t start = a.tile_start[tile], end = a.tile_end[tile];

    // 1. bring the tile into LDS
#pragma unroll
    for (int t = 0; t < NA; t++) {
        const rh_attach_dev& at = a.att[t];
        if (at.load_op == 0u && start == end) continue;          // untouched tile of a Load attachment
        for (int i = tid; i < RH_TILE * RH_TILE; i += RH_TILE_THREADS) {
            const int lx = i & (RH_TILE - 1), ly = i / RH_TILE;
            const uint32_t gx = (uint32_t)(ox + lx), gy = (uint32_t)(oy + ly);
            uint32_t v = 0;
            if (gx < p.width && gy < p.height) {
                if (at.load_op != 0u) v = at.clear_rgba8;
                else {
                    v = reinterpret_cast<const uint32_t*>(at.ptr)[(size_t)gy * at.pitch_px + gx];
                    if (at.format == FORMAT_BGRA_U8) v = swap_rb(v);
                }
            }
            tile_px[t][lds_pixel_index(lx, ly)] = v;
        }
    }

    FragmentContext fctx; fctx.p = &p;

    // 2. batches of RH_BATCH_TRIS triangles = RH_BATCH_TRIS / 2 clusters
    for (uint32_t base = start; base < end; base += RH_BATCH_TRIS / 2) {
        __syncthreads();                                          // tile loaded / previous batch fully consumed
        if (tid < RH_BATCH_TRIS) {
            const uint32_t ci = base + (uint32_t)(tid >> 1);
            uint32_t bb = BBOX_EMPTY;
            if (ci < end) bb = setup_triangle(p, a.list[ci], (uint32_t)(tid & 1), ox, oy, rec[tid]);
            rec_bbox[tid] = bb;
        }
        __syncthreads();

        for (int b = 0; b < 4; b++) {
            const int bx = (wave & 1) * 2 + (b & 1), by = (wave >> 1) * 2 + (b >> 1);
            const int lx = bx * RH_BLOCK + (lane & 7), ly = by * RH_BLOCK + (lane >> 3);
            const int gx = ox + lx, gy = oy + ly;
            const bool in_scissor = gx >= p.sc_x0 && gx < p.sc_x1 && gy >= p.sc_y0 && gy < p.sc_y1;
            const int pidx = (by * (RH_TILE / RH_BLOCK) + bx) * BLOCK_STRIDE + lane;
            uint32_t pix[NA];
#pragma unroll
            for (int t = 0; t < NA; t++) pix[t] = tile_px[t][pidx];
            const uint32_t bx0 = (uint32_t)bx * RH_BLOCK, by0 = (uint32_t)by * RH_BLOCK;
            bool touched = false;
            for (int round = 0; round < RH_BATCH_TRIS / 64; round++) {
                const uint32_t bb = rec_bbox[round * 64 + lane];
                const uint32_t tx0 = bb & 0xffu, ty0 = (bb >> 8) & 0xffu, tx1 = (bb >> 16) & 0xffu, ty1 = bb >> 24;
                const bool hit = tx0 <= bx0 + (RH_BLOCK - 1) && tx1 >= bx0 && ty0 <= by0 + (RH_BLOCK - 1) && ty1 >= by0;
                unsigned long long mask = __ballot(hit);
                while (mask) {
                    const int tl = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const TriRecord& r = rec[round * 64 + tl];
                    const int s0 = __mul24(r.A0, lx) + __mul24(r.B0, ly) + r.T0;
                    const int s1 = __mul24(r.A1, lx) + __mul24(r.B1, ly) + r.T1;
                    const int s2 = __mul24(r.A2, lx) + __mul24(r.B2, ly) + r.T2;
                    if (((s0 | s1 | s2) >= 0) && in_scissor) {
                        const float dx = (float)(lx * 256 - r.X0t) * (1.0f / 256.0f);
                        const float dy = (float)(ly * 256 - r.Y0t) * (1.0f / 256.0f);
                        Varyings var; float* vf = reinterpret_cast<float*>(&var);
#pragma unroll
                        for (int v = 0; v < NV; v++) vf[v] = __builtin_fmaf(r.dvdy[v], dy, __builtin_fmaf(r.dvdx[v], dx, r.v0[v]));
                        if (r.flags & FLAG_PERSP) {
                            const float q = __builtin_fmaf(r.dqdy, dy, __builtin_fmaf(r.dqdx, dx, r.q0));
#pragma unroll
                            for (int v = 0; v < NV; v++) vf[v] = vf[v] / q;
                        }
                        Outputs out;
                        fragment_main(fctx, var, out);
#pragma unroll
                        for (int t = 0; t < NA; t++)
                            pix[t] = blend(pix[t], out.color[t < RH_FRAGMENT_OUTPUTS ? t : RH_FRAGMENT_OUTPUTS - 1], a.src_factor, a.dst_factor);
                        touched = true;
                    }
                }
            }
            if (touched) {
#pragma unroll
                for (int t = 0; t < NA; t++) tile_px[t][pidx] = pix[t];
            }
        }
    }
    __syncthreads();

    // 3. write the tile back
#pragma unroll
    for (int t = 0; t < NA; t++) {
        const rh_attach_dev& at = a.att[t];
        if (at.load_op == 0u && start == end) continue;
        for (int i = tid; i < RH_TILE * RH_TILE; i += RH_TILE_THREADS) {
            const int lx = i & (RH_TILE - 1), ly = i / RH_TILE;
            const uint32_t gx = (uint32_t)(ox + lx), gy = (uint32_t)(oy + ly);
            if (gx < p.width && gy < p.height) {
                uint32_t v = tile_px[t][lds_pixel_index(lx, ly)];
                if (at.format == FORMAT_BGRA_U8) v = swap_rb(v);
                reinterpret_cast<uint32_t*>(at.ptr)[(size_t)gy * at.pitch_px + gx] = v;
            }
        }
    }
}

} // namespace rh

extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_1(rh_tile_args a) { rh::tile_kernel<1>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_2(rh_tile_args a) { rh::tile_kernel<2>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_3(rh_tile_args a) { rh::tile_kernel<3>(a); }
extern "C" __global__ __launch_bounds__(RH_TILE_THREADS) void rh_tile_4(rh_tile_args a) { rh::tile_kernel<4>(a); }
