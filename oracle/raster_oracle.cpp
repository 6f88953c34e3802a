// TEST INFRASTRUCTURE ONLY -- CPU restatement of the draw path, used as the parity checker.
//
// Nothing in the shipped product may include, link or call this file; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// PARITY UNPINNED: the reference (/root/reference, a Rust crate over wgpu) contains no
// rasterisation arithmetic of its own and no golden images; all per-vertex / per-triangle /
// per-fragment arithmetic lives in wgpu -> Vulkan -> Mesa lavapipe, none of which is present
// in /root/reference or buildable here.  This file restates:
//   * the STATE the reference selects (what is bound where, formats, blend factors, sampler
//     modes, topology, viewport, clear/load) -- each function cites the reference file:line;
//   * the Vulkan 1.3 rasterisation rules that state selects (top-left fill rule on a
//     1/256-pixel snapped grid, barycentric plane-equation varyings, clamp-to-edge
//     bilinear/nearest unnormalised-coordinate sampling, fixed-function blend with
//     round-to-nearest-even UNORM8 conversion after every fragment).
// Every float operation below is written as an explicit single IEEE-754 binary32 operation
// (fmaf = one fused rounding) so that another implementation can reproduce it bit for bit.
// Build with -ffp-contract=off.
//
// C ABI (see orc_draw below) so that Python can drive it through ctypes.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

extern "C" {

// Format discriminants follow the declaration order of reference src/format.rs:3-9.
enum { ORC_FMT_R_U8 = 0, ORC_FMT_BGRA_U8 = 1, ORC_FMT_RGBA_U8 = 2, ORC_FMT_RGBA_F16 = 3, ORC_FMT_RGBA_F32 = 4 };
// reference src/filter_mode.rs:2-5
enum { ORC_FILTER_LINEAR = 0, ORC_FILTER_NEAREST = 1 };
// reference src/primitive.rs:2-5
enum { ORC_TOPO_TRIANGLE_LIST = 0, ORC_TOPO_TRIANGLE_STRIP = 1 };
// blend factors used by reference src/blend_mode.rs:8-14 (+ Zero/SrcAlpha for completeness)
enum { ORC_BF_ZERO = 0, ORC_BF_ONE = 1, ORC_BF_SRC_ALPHA = 2, ORC_BF_ONE_MINUS_SRC_ALPHA = 3 };
// built-in shader programs (restatements of GLSL sources)
enum {
    ORC_SHADER_HELLO = 0,        // reference examples/quads/hello.vert:5-15 + hello.frag:7-17
    ORC_SHADER_SIZED = 1,        // hello with a vec4 {x, y, half_w, half_h} instance record
    ORC_SHADER_HELLO_NEAREST_LAYER = 2, // hello sampling layer `uniform[0]` of a 2D-array texture
    ORC_SHADER_TILTED = 3,       // projective quads: per-vertex w, a third varying, a uniform read by both stages
    ORC_SHADER_SPACE = 4,        // quads in space: a 4x4 matrix from a uniform, z and w from the vertex stage
    ORC_SHADER_CROWD = 5,        // hello's vertex stage; the fragment stage adds up six textures, each times a vec4 uniform, times a gain
};

struct orc_texture {
    const uint8_t* data;   // tightly packed rows, layer-major (reference src/texture.rs:45-56)
    uint32_t width, height, layers;
    uint32_t format;       // ORC_FMT_*
    uint32_t filter;       // ORC_FILTER_* (mag = min, mip nearest, lod 0; reference src/texture.rs:141-158)
};

struct orc_target {
    uint8_t* data;         // width*height texels of the format's size (1, 4, 8 or 16 bytes), row 0 = top
    uint32_t width, height;
    uint32_t format;       // any ORC_FMT_* (reference src/target.rs:8-13: the screen is BGRA8, a texture has its own format)
    int32_t  clear;        // 1 = LoadOp::Clear(clear_rgba), 0 = LoadOp::Load (reference src/render_pass.rs:71-74)
    double   clear_rgba[4];// reference src/clear_color.rs:7-16 (f32 -> f64)
};

struct orc_draw {
    int32_t  shader;                 // ORC_SHADER_*
    int32_t  topology;               // ORC_TOPO_*
    uint32_t vertex_count;           // draw(0..vertex_count, 0..instance_count): reference src/render_pass.rs:43
    uint32_t instance_count;
    const float* attr[8];            // one tightly packed f32 buffer per attribute location (reference src/attribute.rs:10-29)
    uint32_t attr_size[8];           // 1..4 floats
    const float* storage[4];         // read-only storage buffers, in slot order (reference src/instanced.rs:9-38)
    const float* uniform[8];         // uniform buffers (reference src/uniform.rs:9-38)
    orc_texture  tex[8];             // textures in slot order
    int32_t  has_viewport;           // reference src/render_pass.rs:39-41
    float    viewport[4];            // margin_x, margin_y, width, height
    int32_t  clip_to_viewport;       // 1: fragments outside the viewport rectangle are clipped (strict clip volume)
    uint32_t src_factor, dst_factor; // same factors for colour and alpha, op Add (reference src/blend_mode.rs:23-40)
    uint32_t n_targets;              // colour attachments incl. the recording attachment (reference src/render_pass.rs:60-69)
    orc_target targets[4];
    int32_t  threads;                // <=0: hardware concurrency
    uint64_t fragments_out;          // out: number of blended fragments per attachment
    uint32_t samples;                // 0 or 1: single-sampled.  2, 4, 8: the pass renders into `msaa` and resolves into
                                     // targets[0] (reference src/render_pass.rs:76-81, src/pipeline.rs:164-176, 209-211)
    uint8_t* msaa;                   // width*height*samples texels of targets[0].format (reference src/pipeline.rs:164-176), the
                                     // samples of a pixel adjacent; persists between passes (LoadOp::Load reads it back)
};

int orc_render(orc_draw* d);

} // extern "C"

namespace {

constexpr int SUBPIXEL_BITS = 8;
constexpr float SUBPIXEL_SCALE = 256.0f;
constexpr float INV_SUBPIXEL = 1.0f / 256.0f;
// Guard band: snapped coordinates are clamped to [-8192 px, 24576 px) so that every edge delta fits 24 signed bits.
constexpr float GB_LO = -8192.0f * 256.0f;
constexpr float GB_HI = 24576.0f * 256.0f - 1.0f;
constexpr float INV_255 = 1.0f / 255.0f;
constexpr int MAX_VARYINGS = 8;

struct VsOut { float pos[4]; float var[MAX_VARYINGS]; };

struct Shader {
    int n_varyings;
    int n_outputs;
    void (*vs)(const orc_draw&, uint32_t vertex, uint32_t instance, VsOut&);
    void (*fs)(const orc_draw&, const float* var, float out[][4]);
};

// ---- sampler: Vulkan unnormalised-coordinate rules for the sampler the reference creates --------------------
// ClampToEdge on every axis, mag = min = filter, one mip level (reference src/texture.rs:141-158).
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// IEEE binary16 -> binary32, exact (subnormals, infinities and NaNs included)
inline float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { int e = -1; uint32_t m = man; do { e++; m <<= 1; } while (!(m & 0x400u)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13); }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f; std::memcpy(&f, &bits, 4); return f;
}

// f32 -> f16, round to nearest even, overflow to infinity, denormals kept (IEEE 754 binary16)
inline uint16_t float_to_half(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u | ((x >> 13) & 0x3ffu) : 0u));   // inf / NaN
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                  // rounds to >= 2^16: infinity
    if (x < 0x33000001u) return (uint16_t)sign;                               // <= 2^-25: rounds to zero
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;                                 // 24-bit significand
    int shift = e < -14 ? 13 + (-14 - e) : 13;                                // bits dropped (denormal results drop more)
    uint32_t kept = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (kept & 1u))) kept++;
    // kept carries the implicit bit for normal results: adding the exponent field absorbs a mantissa carry correctly
    uint32_t h = e < -14 ? kept : (((uint32_t)(e + 15) << 10) + (kept - 0x400u));
    return (uint16_t)(sign | h);
}

inline void fetch_texel(const orc_texture& t, int x, int y, uint32_t layer, float out[4]) {
    size_t idx = ((size_t)layer * t.height + (size_t)y) * t.width + (size_t)x;
    switch (t.format) {
    case ORC_FMT_R_U8: out[0] = (float)t.data[idx]; out[1] = 0.0f; out[2] = 0.0f; out[3] = 255.0f; break;
    case ORC_FMT_BGRA_U8: { const uint8_t* p = t.data + idx * 4; out[0] = p[2]; out[1] = p[1]; out[2] = p[0]; out[3] = p[3]; break; }
    case ORC_FMT_RGBA_F32: { float f[4]; std::memcpy(f, t.data + idx * 16, 16); for (int k = 0; k < 4; k++) out[k] = f[k]; break; }
    case ORC_FMT_RGBA_F16: { uint16_t h[4]; std::memcpy(h, t.data + idx * 8, 8); for (int k = 0; k < 4; k++) out[k] = half_to_float(h[k]); break; }
    default: { const uint8_t* p = t.data + idx * 4; out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3]; break; }
    }
}

// Result in [0,1].  Texel values are kept on the 0..255 scale through the lerps and scaled once at the end.
//
// Linear filtering, clamp-to-edge, written in the weight-clamp form: with u = s*W - 0.5 the two taps along x are
// texels ib and min(ib + 1, W - 1) where ib = clamp(floor(u), 0, max(W - 2, 0)), and the weight is
// a = clamp(u - ib, 0, 1).  Inside the texture this is the textbook (floor(u), frac(u)); beyond the edge texel centres
// the weight saturates, so the result equals the edge texel exactly -- the same value the Vulkan formulation
// (both tap indices clamped to the edge) produces.  Rows are combined with lerp(t, b, w) = fma(w, b, fma(-w, t, t)),
// which returns t at w = 0 and b at w = 1 exactly.
void sample_2d(const orc_texture& t, float s, float tc, uint32_t layer, float out[4]) {
    const int W = (int)t.width, H = (int)t.height;
    if (layer >= t.layers) layer = t.layers - 1;
    // 8-bit texels are filtered on the 0..255 scale and normalised once at the end; float texels are not scaled
    const float scale = (t.format == ORC_FMT_RGBA_F16 || t.format == ORC_FMT_RGBA_F32) ? 1.0f : INV_255;
    if (t.filter == ORC_FILTER_NEAREST) {
        float u = floorf(s * (float)W), v = floorf(tc * (float)H);
        u = fminf(fmaxf(u, -1.0f), (float)W); v = fminf(fmaxf(v, -1.0f), (float)H);
        int i = clampi((int)u, 0, W - 1), j = clampi((int)v, 0, H - 1);
        float c[4]; fetch_texel(t, i, j, layer, c);
        for (int k = 0; k < 4; k++) out[k] = c[k] * scale;
        return;
    }
    const float u = fmaf(s, (float)W, -0.5f), v = fmaf(tc, (float)H, -0.5f);
    const float ibf = floorf(fminf(fmaxf(u, 0.0f), (float)(W >= 2 ? W - 2 : 0)));
    const float jbf = floorf(fminf(fmaxf(v, 0.0f), (float)(H >= 2 ? H - 2 : 0)));
    const float a = fminf(fmaxf(u - ibf, 0.0f), 1.0f), b = fminf(fmaxf(v - jbf, 0.0f), 1.0f);
    const int ib = (int)ibf, jb = (int)jbf;
    const int i1 = ib + 1 < W ? ib + 1 : W - 1, j1 = jb + 1 < H ? jb + 1 : H - 1;
    float t00[4], t10[4], t01[4], t11[4];
    fetch_texel(t, ib, jb, layer, t00); fetch_texel(t, i1, jb, layer, t10);
    fetch_texel(t, ib, j1, layer, t01); fetch_texel(t, i1, j1, layer, t11);
    for (int k = 0; k < 4; k++) {
        float top = fmaf(a, t10[k] - t00[k], t00[k]);
        float bot = fmaf(a, t11[k] - t01[k], t01[k]);
        float c = fmaf(b, bot, fmaf(-b, top, top));
        out[k] = c * scale;
    }
}

// ---- shaders ---------------------------------------------------------------------------------------------------
// hello.vert:12-15: v_tex_coord = a_tex_coord; gl_Position = vec4(a_position + i_offset[gl_InstanceIndex], 0, 1)
void hello_vs(const orc_draw& d, uint32_t v, uint32_t inst, VsOut& o) {
    const float* pos = d.attr[0] + (size_t)v * d.attr_size[0];
    const float* uv = d.attr[1] + (size_t)v * d.attr_size[1];
    const float* off = d.storage[0] + (size_t)inst * 2;   // ArrayStride 8 (std430 vec2[])
    o.pos[0] = pos[0] + off[0]; o.pos[1] = pos[1] + off[1]; o.pos[2] = 0.0f; o.pos[3] = 1.0f;
    o.var[0] = uv[0]; o.var[1] = uv[1];
}
// hello.frag:14-17: f_color = texture(sampler2D(t_texture, s_sampler), v_tex_coord); f_recording = f_color
void hello_fs(const orc_draw& d, const float* var, float out[][4]) {
    float c[4]; sample_2d(d.tex[0], var[0], var[1], 0, c);
    for (uint32_t a = 0; a < 2; a++) for (int k = 0; k < 4; k++) out[a][k] = c[k];
}
// hello with per-instance size: storage record vec4 {x, y, half_w, half_h}; pos = a_position * half + xy
void sized_vs(const orc_draw& d, uint32_t v, uint32_t inst, VsOut& o) {
    const float* pos = d.attr[0] + (size_t)v * d.attr_size[0];
    const float* uv = d.attr[1] + (size_t)v * d.attr_size[1];
    const float* r = d.storage[0] + (size_t)inst * 4;
    o.pos[0] = fmaf(pos[0], r[2], r[0]); o.pos[1] = fmaf(pos[1], r[3], r[1]); o.pos[2] = 0.0f; o.pos[3] = 1.0f;
    o.var[0] = uv[0]; o.var[1] = uv[1];
}
// hello sampling layer uniform[0][0] (as float -> int) of a 2D array
void layer_fs(const orc_draw& d, const float* var, float out[][4]) {
    float lf = d.uniform[0] ? d.uniform[0][0] : 0.0f;
    uint32_t layer = lf > 0.0f ? (uint32_t)lf : 0u;
    float c[4]; sample_2d(d.tex[0], var[0], var[1], layer, c);
    for (uint32_t a = 0; a < 2; a++) for (int k = 0; k < 4; k++) out[a][k] = c[k];
}

// Projective quads.  Storage record vec4 {cx, cy, half, tilt}; uniform floats {tint r, g, b, a, scale_x, scale_y}:
//   w = a_position.x * tilt + 1;  gl_Position = vec4((a_position.x*scale_x*half + cx) * w, (a_position.y*scale_y*half + cy) * w, 0, w)
//   varyings: tex_coord, shade = a_position.y * 0.25 + 0.75;  colour = (texture * tint) * shade
void tilted_vs(const orc_draw& d, uint32_t v, uint32_t inst, VsOut& o) {
    const float* pos = d.attr[0] + (size_t)v * d.attr_size[0];
    const float* uv = d.attr[1] + (size_t)v * d.attr_size[1];
    const float* r = d.storage[0] + (size_t)inst * 4;
    const float* u = d.uniform[0];
    const float w = fmaf(pos[0], r[3], 1.0f);
    const float x = fmaf(pos[0] * u[4], r[2], r[0]), y = fmaf(pos[1] * u[5], r[2], r[1]);
    o.pos[0] = x * w; o.pos[1] = y * w; o.pos[2] = 0.0f; o.pos[3] = w;
    o.var[0] = uv[0]; o.var[1] = uv[1]; o.var[2] = fmaf(pos[1], 0.25f, 0.75f);
}
void tilted_fs(const orc_draw& d, const float* var, float out[][4]) {
    const float* u = d.uniform[0];
    float c[4]; sample_2d(d.tex[0], var[0], var[1], 0, c);
    for (uint32_t a = 0; a < 2; a++) for (int k = 0; k < 4; k++) out[a][k] = (c[k] * u[k]) * var[2];
}

// Quads in space.  Storage record vec4 {cx, cy, half, depth}; uniform floats {tint r, g, b, a, m[16] (column-major)}:
//   local = (a_position.x * half + cx, a_position.y * half + cy, depth, 1);  gl_Position = M * local
//   varyings: tex_coord, shade = 1;  fragment stage as for the projective quads
void space_vs(const orc_draw& d, uint32_t v, uint32_t inst, VsOut& o) {
    const float* pos = d.attr[0] + (size_t)v * d.attr_size[0];
    const float* uv = d.attr[1] + (size_t)v * d.attr_size[1];
    const float* r = d.storage[0] + (size_t)inst * 4;
    const float* m = d.uniform[0] + 4;
    const float lx = fmaf(pos[0], r[2], r[0]), ly = fmaf(pos[1], r[2], r[1]), lz = r[3];
    for (int i = 0; i < 4; i++) o.pos[i] = fmaf(m[i], lx, fmaf(m[4 + i], ly, fmaf(m[8 + i], lz, m[12 + i])));
    o.var[0] = uv[0]; o.var[1] = uv[1]; o.var[2] = 1.0f;
}

// Twenty binding slots (reference src/pipeline.rs:25-27: four to a group): uniform[0] = gain, uniform[1..6] = one vec4 weight
// per texture; colour = gain * sum_i texture_i(uv) * w_i, accumulated in order with one fma per channel; one colour output.
void crowd_fs(const orc_draw& d, const float* var, float out[][4]) {
    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (int i = 0; i < 6; i++) {
        float c[4]; sample_2d(d.tex[i], var[0], var[1], 0, c);
        const float* w = d.uniform[1 + i];
        for (int k = 0; k < 4; k++) acc[k] = fmaf(c[k], w[k], acc[k]);
    }
    const float gain = d.uniform[0][0];
    for (int k = 0; k < 4; k++) out[0][k] = acc[k] * gain;
}

const Shader SHADERS[] = {
    { 2, 2, hello_vs, hello_fs },
    { 2, 2, sized_vs, hello_fs },
    { 2, 2, hello_vs, layer_fs },
    { 3, 2, tilted_vs, tilted_fs },
    { 3, 2, space_vs, tilted_fs },
    { 2, 1, hello_vs, crowd_fs },
};

// ---- triangle setup ----------------------------------------------------------------------------------------------
// `out`: the clip planes the vertex is outside of (OUT_*; 0 = inside the clip volume = `ok`); `clip`: what the clipper starts from.
struct Vert { int32_t X, Y; float rw; bool ok; float var[MAX_VARYINGS]; float clip[4]; uint32_t out; };

struct Tri {
    int32_t A[3], B[3]; int64_t Cq[3];   // inside <=> A*i + B*j + Cq >= 0 for all three edges (i, j = pixel indices)
    int64_t Cf[3];                       // inside at a sub-pixel position P (1/256 px) <=> A*Px + B*Py + Cf >= 0
    int32_t X0, Y0;
    int x0, y0, x1, y1;                  // inclusive pixel bbox, already intersected with the scissor
    float v0[MAX_VARYINGS], dvdx[MAX_VARYINGS], dvdy[MAX_VARYINGS];
    float q0, dqdx, dqdy;                // 1/w plane
    bool persp;
};

struct Scissor { int x0, y0, x1, y1; };  // [x0,x1) x [y0,y1)

struct ViewXf { float hw, hh, ox, oy; };

inline int32_t snap(float w) {
    float t = (w - 0.5f) * SUBPIXEL_SCALE;        // pixel centres land on multiples of 256
    t = fminf(fmaxf(t, GB_LO), GB_HI);
    return (int32_t)rintf(t);                      // round half to even
}

// ---- clip volume -------------------------------------------------------------------------------------------------------
// The pipeline state the reference selects clips primitives to the view volume (reference src/pipeline.rs:197-207:
// `unclipped_depth` unset).  A vertex is inside when w >= W_MIN, 0 <= z <= w (Vulkan 1.3 "Primitive Clipping" with the
// 0..1 depth range) and its window position lies within [BAND_LO, BAND_HI] pixels on both axes (a band inside the
// snapping guard band; fragments outside the viewport are removed by the scissor, which equals clipping at the
// viewport's edges).  A triangle with all three vertices inside is drawn as it is; one with a NaN coordinate or with all
// three vertices outside the same plane is dropped; any other is clipped against the seven planes (Sutherland-Hodgman,
// in the order of clip_distance) and the convex polygon that is left is drawn as a fan (0, i+1, i+2), in that order.
constexpr float W_MIN = 1.0f / 1048576.0f;
constexpr float BAND_LO = -8000.0f, BAND_HI = 24000.0f;
constexpr int CLIP_PLANES = 7, CLIP_MAX_VERTS = 3 + CLIP_PLANES;

enum : uint32_t { OUT_W = 1u, OUT_NEAR = 2u, OUT_FAR = 4u, OUT_LEFT = 8u, OUT_RIGHT = 16u, OUT_TOP = 32u, OUT_BOTTOM = 64u, OUT_NAN = 128u };

Vert run_vertex(const orc_draw& d, const Shader& sh, const ViewXf& xf, uint32_t v, uint32_t inst) {
    VsOut o; std::memset(&o, 0, sizeof o);
    sh.vs(d, v, inst, o);
    Vert r; r.ok = false; r.X = r.Y = 0;
    float x = o.pos[0], y = o.pos[1], z = o.pos[2], w = o.pos[3];
    for (int k = 0; k < 4; k++) r.clip[k] = o.pos[k];
    for (int k = 0; k < sh.n_varyings; k++) r.var[k] = o.var[k];
    const bool nan = std::isnan(x) || std::isnan(y) || std::isnan(z) || std::isnan(w);
    const bool front = !nan && w >= W_MIN;      // only then does the vertex have a window position
    float rw = 1.0f;
    if (w != 1.0f && front) { rw = 1.0f / w; x = x * rw; y = y * rw; }
    // Vulkan viewport transform with wgpu's y-up NDC: x_w = x*w/2 + (vx + w/2), y_w = -y*h/2 + (vy + h/2)
    float xw = fmaf(x, xf.hw, xf.ox);
    float yw = fmaf(y, -xf.hh, xf.oy);
    r.rw = rw;
    uint32_t out = nan ? OUT_NAN : (w >= W_MIN ? 0u : OUT_W);
    if (!nan) out |= (z < 0.0f ? OUT_NEAR : 0u) | (z > w ? OUT_FAR : 0u);
    if (front) out |= (xw < BAND_LO ? OUT_LEFT : 0u) | (xw > BAND_HI ? OUT_RIGHT : 0u) | (yw < BAND_LO ? OUT_TOP : 0u) | (yw > BAND_HI ? OUT_BOTTOM : 0u);
    r.out = out; r.ok = out == 0u;
    if (r.ok) { r.X = snap(xw); r.Y = snap(yw); }
    return r;
}

struct ClipVert { float c[4]; float var[MAX_VARYINGS]; };

// Signed distance to clip plane `plane`; inside: not negative.  Plane 0 comes first so that w is positive for the rest.
inline float clip_distance(const ViewXf& xf, const ClipVert& v, int plane) {
    const float x = v.c[0], y = v.c[1], z = v.c[2], w = v.c[3];
    switch (plane) {
    case 0: return w - W_MIN;
    case 1: return z;
    case 2: return w - z;
    case 3: return fmaf(x, xf.hw, (xf.ox - BAND_LO) * w);      // x_w >= BAND_LO
    case 4: return fmaf(-x, xf.hw, (BAND_HI - xf.ox) * w);     // x_w <= BAND_HI
    case 5: return fmaf(-y, xf.hh, (xf.oy - BAND_LO) * w);     // y_w >= BAND_LO
    default: return fmaf(y, xf.hh, (BAND_HI - xf.oy) * w);     // y_w <= BAND_HI
    }
}

// The part of triangle (a, b, c) inside the clip volume as projected, snapped vertices; returns their number (0: nothing).
// A new vertex is computed from the inside end of its edge towards the outside end: t = d_in / (d_in - d_out), every
// coordinate and varying fma(t, out - in, in) -- neighbours that share the clipped edge get the same point.
int clip_triangle(const Shader& sh, const ViewXf& xf, const Vert& a, const Vert& b, const Vert& c, Vert out[CLIP_MAX_VERTS]) {
    // a NaN coordinate, or all three vertices outside one plane: nothing to draw
    if (((a.out | b.out | c.out) & OUT_NAN) || (a.out & b.out & c.out)) return 0;
    ClipVert bufs[2][CLIP_MAX_VERTS];
    const Vert* in3[3] = { &a, &b, &c };
    for (int k = 0; k < 3; k++) {
        for (int j = 0; j < 4; j++) bufs[0][k].c[j] = in3[k]->clip[j];
        for (int j = 0; j < sh.n_varyings; j++) bufs[0][k].var[j] = in3[k]->var[j];
    }
    int n = 3, cur_buf = 0;
    for (int plane = 0; plane < CLIP_PLANES && n >= 3; plane++) {
        const ClipVert* src = bufs[cur_buf]; ClipVert* dst = bufs[cur_buf ^ 1];
        int m = 0;
        for (int i = 0; i < n; i++) {
            const ClipVert& cur = src[i]; const ClipVert& nxt = src[(i + 1) % n];
            const float dc = clip_distance(xf, cur, plane), dn = clip_distance(xf, nxt, plane);
            const bool cin = !(dc < 0.0f), nin = !(dn < 0.0f);
            // (a convex polygon gains at most one vertex per plane; with infinite coordinates the distances can be NaN and the
            // inside tests inconsistent: vertices beyond the room of the arrays are left out)
            if (cin && m < CLIP_MAX_VERTS) dst[m++] = cur;
            if (cin != nin && m < CLIP_MAX_VERTS) {
                const ClipVert& vi = cin ? cur : nxt; const ClipVert& vo = cin ? nxt : cur;
                const float di = cin ? dc : dn, dout = cin ? dn : dc;
                const float t = di / (di - dout);
                ClipVert nv;
                for (int j = 0; j < 4; j++) nv.c[j] = fmaf(t, vo.c[j] - vi.c[j], vi.c[j]);
                for (int j = 0; j < sh.n_varyings; j++) nv.var[j] = fmaf(t, vo.var[j] - vi.var[j], vi.var[j]);
                dst[m++] = nv;
            }
        }
        cur_buf ^= 1; n = m;
    }
    if (n < 3) return 0;
    for (int i = 0; i < n; i++) {
        const ClipVert& v = bufs[cur_buf][i];
        const float w = fmaxf(v.c[3], W_MIN), rw = 1.0f / w;
        const float x = v.c[0] * rw, y = v.c[1] * rw;
        Vert& r = out[i];
        r.X = snap(fmaf(x, xf.hw, xf.ox)); r.Y = snap(fmaf(y, -xf.hh, xf.oy)); r.rw = rw; r.ok = true; r.out = 0u;
        for (int j = 0; j < sh.n_varyings; j++) r.var[j] = v.var[j];
    }
    return n;
}

inline int64_t asr8(int64_t v) { return v >> SUBPIXEL_BITS; }   // floor division by 256

// Standard sample locations (Vulkan 1.3 section 25.3, "Multisampling"), as offsets from the pixel centre in 1/256 px.
const int8_t SAMPLE_OFFSETS_2[2][2] = { { 64, 64 }, { -64, -64 } };
const int8_t SAMPLE_OFFSETS_4[4][2] = { { -32, -96 }, { 96, -32 }, { -96, 32 }, { 32, 96 } };
const int8_t SAMPLE_OFFSETS_8[8][2] = { { 16, -48 }, { -16, 48 }, { 80, 16 }, { -48, -80 }, { -80, 80 }, { -112, -16 }, { 48, 112 }, { 112, -112 } };
inline const int8_t (*sample_offsets(uint32_t samples))[2] {
    return samples == 2 ? SAMPLE_OFFSETS_2 : (samples == 4 ? SAMPLE_OFFSETS_4 : SAMPLE_OFFSETS_8);
}

// `pad`: 0 for single-sampled passes; 128 (half a pixel) when samples away from the centre can be covered.
bool setup_tri(const Shader& sh, Vert a, Vert b, Vert c, const Scissor& sc, int pad, Tri& t) {
    if (!a.ok || !b.ok || !c.ok) return false;
    int64_t area2 = (int64_t)(b.X - a.X) * (c.Y - a.Y) - (int64_t)(c.X - a.X) * (b.Y - a.Y);
    if (area2 == 0) return false;
    if (area2 < 0) std::swap(b, c);     // no culling (reference src/pipeline.rs:197-207): both windings drawn
    const Vert* v[3] = { &a, &b, &c };
    for (int e = 0; e < 3; e++) {
        const Vert& p = *v[e]; const Vert& q = *v[(e + 1) % 3];
        int32_t A = p.Y - q.Y, B = q.X - p.X;
        int64_t C = -((int64_t)A * p.X + (int64_t)B * p.Y);
        bool top_left = (A > 0) || (A == 0 && B > 0);
        t.A[e] = A; t.B[e] = B; t.Cf[e] = C + (top_left ? 0 : -1); t.Cq[e] = asr8(t.Cf[e]);
    }
    int32_t minX = std::min(a.X, std::min(b.X, c.X)), maxX = std::max(a.X, std::max(b.X, c.X));
    int32_t minY = std::min(a.Y, std::min(b.Y, c.Y)), maxY = std::max(a.Y, std::max(b.Y, c.Y));
    t.x0 = std::max(sc.x0, (int)((minX - pad + 255) >> 8)); t.x1 = std::min(sc.x1 - 1, (int)((maxX + pad) >> 8));
    t.y0 = std::max(sc.y0, (int)((minY - pad + 255) >> 8)); t.y1 = std::min(sc.y1 - 1, (int)((maxY + pad) >> 8));
    if (t.x0 > t.x1 || t.y0 > t.y1) return false;
    t.X0 = a.X; t.Y0 = a.Y;
    float e1x = (float)(b.X - a.X) * INV_SUBPIXEL, e1y = (float)(b.Y - a.Y) * INV_SUBPIXEL;
    float e2x = (float)(c.X - a.X) * INV_SUBPIXEL, e2y = (float)(c.Y - a.Y) * INV_SUBPIXEL;
    float area = fmaf(e1x, e2y, -(e2x * e1y));
    float inv = 1.0f / area;
    t.persp = !(a.rw == 1.0f && b.rw == 1.0f && c.rw == 1.0f);
    auto plane = [&](float f0, float f1, float f2, float& o0, float& ox, float& oy) {
        float d1 = f1 - f0, d2 = f2 - f0;
        o0 = f0;
        // gradients are stored per 1/256 pixel so that the fragment stage multiplies them by exact fixed-point deltas
        ox = (fmaf(d1, e2y, -(d2 * e1y)) * inv) * INV_SUBPIXEL;
        oy = (fmaf(d2, e1x, -(d1 * e2x)) * inv) * INV_SUBPIXEL;
    };
    for (int k = 0; k < sh.n_varyings; k++) {
        if (t.persp) plane(a.var[k] * a.rw, b.var[k] * b.rw, c.var[k] * c.rw, t.v0[k], t.dvdx[k], t.dvdy[k]);
        else plane(a.var[k], b.var[k], c.var[k], t.v0[k], t.dvdx[k], t.dvdy[k]);
    }
    plane(a.rw, b.rw, c.rw, t.q0, t.dqdx, t.dqdy);
    return true;
}

// ---- ROP ---------------------------------------------------------------------------------------------------------------
inline float factor(uint32_t f, float src_alpha) {
    switch (f) {
    case ORC_BF_ZERO: return 0.0f;
    case ORC_BF_ONE: return 1.0f;
    case ORC_BF_SRC_ALPHA: return src_alpha;
    default: return 1.0f - src_alpha;
    }
}

inline uint8_t to_unorm8(float o) {
    o = fminf(fmaxf(o, 0.0f), 1.0f);              // NaN -> 0
    return (uint8_t)rintf(o * 255.0f);             // round half to even
}

// out = src*SF + dst*DF per channel, alpha included (reference src/blend_mode.rs:31-40), destination read back as
// UNORM8 and the result re-quantised after every fragment.  Evaluated on the 0..255 scale:
//   q = rint(clamp(fma(dst_byte, DF, (src*SF)*255), 0, 255)),  NaN -> 0, ties to even.
// Float targets blend in f32 without clamping, out = fma(dst, DF, src*SF); an RGBA16F result is rounded to f16 (nearest
// even) after every fragment.  An R8 target stores the red channel only.
inline void blend_store(const orc_draw& d, const orc_target& tg, size_t pix, const float src[4]) {
    const float sf = factor(d.src_factor, src[3]), df = factor(d.dst_factor, src[3]);
    if (tg.format == ORC_FMT_RGBA_F32 || tg.format == ORC_FMT_RGBA_F16) {
        for (int k = 0; k < 4; k++) {
            float dst;
            if (tg.format == ORC_FMT_RGBA_F32) std::memcpy(&dst, tg.data + (pix * 4 + k) * 4, 4);
            else { uint16_t h; std::memcpy(&h, tg.data + (pix * 4 + k) * 2, 2); dst = half_to_float(h); }
            const float o = fmaf(dst, df, src[k] * sf);
            if (tg.format == ORC_FMT_RGBA_F32) std::memcpy(tg.data + (pix * 4 + k) * 4, &o, 4);
            else { const uint16_t h = float_to_half(o); std::memcpy(tg.data + (pix * 4 + k) * 2, &h, 2); }
        }
        return;
    }
    if (tg.format == ORC_FMT_R_U8) {
        float o = fmaf((float)tg.data[pix], df, (src[0] * sf) * 255.0f);
        o = fminf(fmaxf(o, 0.0f), 255.0f);
        tg.data[pix] = (uint8_t)rintf(o);
        return;
    }
    uint8_t* p = tg.data + pix * 4;
    const bool bgra = tg.format == ORC_FMT_BGRA_U8;
    uint8_t q[4];
    for (int k = 0; k < 4; k++) {
        int byte = bgra ? (k == 0 ? 2 : (k == 2 ? 0 : k)) : k;
        float s255 = (src[k] * sf) * 255.0f;
        float o = fmaf((float)p[byte], df, s255);
        o = fminf(fmaxf(o, 0.0f), 255.0f);         // NaN -> 0
        q[k] = (uint8_t)rintf(o);
    }
    if (bgra) { p[0] = q[2]; p[1] = q[1]; p[2] = q[0]; p[3] = q[3]; }
    else { p[0] = q[0]; p[1] = q[1]; p[2] = q[2]; p[3] = q[3]; }
}

inline uint32_t texel_bytes(uint32_t format) {
    return format == ORC_FMT_R_U8 ? 1u : (format == ORC_FMT_RGBA_F16 ? 8u : (format == ORC_FMT_RGBA_F32 ? 16u : 4u));
}

// The clear colour as one texel of the target's format (reference src/clear_color.rs:7-16, src/render_pass.rs:71-74).
inline void clear_texel(const orc_target& tg, uint8_t out[16]) {
    if (tg.format == ORC_FMT_R_U8) { out[0] = to_unorm8((float)tg.clear_rgba[0]); return; }
    if (tg.format == ORC_FMT_RGBA_F32 || tg.format == ORC_FMT_RGBA_F16) {
        for (int k = 0; k < 4; k++) {
            const float c = (float)tg.clear_rgba[k];
            if (tg.format == ORC_FMT_RGBA_F32) std::memcpy(out + k * 4, &c, 4);
            else { const uint16_t h = float_to_half(c); std::memcpy(out + k * 2, &h, 2); }
        }
        return;
    }
    uint8_t q[4];
    for (int k = 0; k < 4; k++) q[k] = to_unorm8((float)tg.clear_rgba[k]);
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
    if (tg.format == ORC_FMT_BGRA_U8) { out[0] = q[2]; out[2] = q[0]; }
}

// Rows [y0, y1) of the target -- or, for a multisampled pass, of the multisampled texture (every sample).
void clear_rows(const orc_target& tg, int y0, int y1, uint8_t* msaa = nullptr, uint32_t samples = 1) {
    if (!tg.clear) return;
    uint8_t px[16]; clear_texel(tg, px);
    const uint32_t bpt = texel_bytes(tg.format);
    uint8_t* const base = msaa ? msaa : tg.data;
    const size_t per_row = (size_t)tg.width * (msaa ? samples : 1u);
    for (size_t k = (size_t)y0 * per_row; k < (size_t)y1 * per_row; k++) std::memcpy(base + k * bpt, px, bpt);
}

// Multisample resolve: per channel the mean of the samples.  8-bit formats: the sum is exact, the mean rounded to
// nearest, ties to even.  Float formats: the samples are added in f32 in sample order and multiplied by 1/S (exact);
// an RGBA16F result is rounded to f16 (nearest even).
void resolve_rows(const orc_target& tg, const uint8_t* msaa, uint32_t samples, int y0, int y1) {
    const float inv = 1.0f / (float)samples;
    const uint32_t bpt = texel_bytes(tg.format);
    for (size_t pix = (size_t)y0 * tg.width; pix < (size_t)y1 * tg.width; pix++) {
        if (tg.format == ORC_FMT_RGBA_F32 || tg.format == ORC_FMT_RGBA_F16) {
            for (int k = 0; k < 4; k++) {
                float sum = 0.0f;
                for (uint32_t sidx = 0; sidx < samples; sidx++) {
                    float v;
                    if (tg.format == ORC_FMT_RGBA_F32) std::memcpy(&v, msaa + ((pix * samples + sidx) * 4 + k) * 4, 4);
                    else { uint16_t h; std::memcpy(&h, msaa + ((pix * samples + sidx) * 4 + k) * 2, 2); v = half_to_float(h); }
                    sum = sidx == 0 ? v : sum + v;
                }
                const float o = sum * inv;
                if (tg.format == ORC_FMT_RGBA_F32) std::memcpy(tg.data + (pix * 4 + k) * 4, &o, 4);
                else { const uint16_t h = float_to_half(o); std::memcpy(tg.data + (pix * 4 + k) * 2, &h, 2); }
            }
            continue;
        }
        for (uint32_t k = 0; k < bpt; k++) {
            uint32_t sum = 0;
            for (uint32_t sidx = 0; sidx < samples; sidx++) sum += msaa[(pix * samples + sidx) * bpt + k];
            tg.data[pix * bpt + k] = (uint8_t)rintf((float)sum * inv);
        }
    }
}

// Triangles of instance `inst`, in primitive order.  Strip triangle i uses vertices (i, i+1, i+2) for even i and
// (i, i+2, i+1) for odd i (Vulkan 1.3 section 21.1.7); list triangle i uses (3i, 3i+1, 3i+2).
inline uint32_t tris_per_instance(const orc_draw& d) {
    if (d.topology == ORC_TOPO_TRIANGLE_STRIP) return d.vertex_count >= 3 ? d.vertex_count - 2 : 0;
    return d.vertex_count / 3;
}
inline void tri_vertices(const orc_draw& d, uint32_t i, uint32_t idx[3]) {
    if (d.topology == ORC_TOPO_TRIANGLE_STRIP) {
        idx[0] = i; idx[1] = (i & 1) ? i + 2 : i + 1; idx[2] = (i & 1) ? i + 1 : i + 2;
    } else { idx[0] = 3 * i; idx[1] = 3 * i + 1; idx[2] = 3 * i + 2; }
}

void shade_tri(const orc_draw& d, const Shader& sh, const Tri& t, int band_y0, int band_y1, uint64_t& frags) {
    int y0 = std::max(t.y0, band_y0), y1 = std::min(t.y1, band_y1 - 1);
    const uint32_t S = d.samples > 1 ? d.samples : 1;
    const int8_t (*offs)[2] = sample_offsets(S);
    for (int j = y0; j <= y1; j++) {
        for (int i = t.x0; i <= t.x1; i++) {
            uint32_t covered = 0;           // sample mask
            if (S == 1) {
                bool in = true;
                for (int e = 0; e < 3; e++)
                    in = in && ((int64_t)t.A[e] * i + (int64_t)t.B[e] * j + t.Cq[e] >= 0);
                covered = in ? 1u : 0u;
            } else {
                // the same edge functions and top-left rule, evaluated at every sample position
                for (uint32_t sidx = 0; sidx < S; sidx++) {
                    const int64_t px = (int64_t)i * 256 + offs[sidx][0], py = (int64_t)j * 256 + offs[sidx][1];
                    bool in = true;
                    for (int e = 0; e < 3; e++) in = in && (t.A[e] * px + t.B[e] * py + t.Cf[e] >= 0);
                    if (in) covered |= 1u << sidx;
                }
            }
            if (!covered) continue;
            // one fragment shader invocation per pixel, inputs interpolated at the pixel centre (no centroid qualifier
            // in the reference's shaders), its colour blended into every covered sample
            float dx = (float)(i * 256 - t.X0), dy = (float)(j * 256 - t.Y0);     // exact: |delta| < 2^24
            float var[MAX_VARYINGS];
            for (int k = 0; k < sh.n_varyings; k++) var[k] = fmaf(t.dvdy[k], dy, fmaf(t.dvdx[k], dx, t.v0[k]));
            if (t.persp) {
                float q = fmaf(t.dqdy, dy, fmaf(t.dqdx, dx, t.q0));
                for (int k = 0; k < sh.n_varyings; k++) var[k] = var[k] / q;
            }
            float out[4][4];
            sh.fs(d, var, out);
            if (S > 1) {
                orc_target ms = d.targets[0]; ms.data = d.msaa;
                for (uint32_t sidx = 0; sidx < S; sidx++)
                    if (covered & (1u << sidx)) blend_store(d, ms, ((size_t)j * ms.width + (size_t)i) * S + sidx, out[0]);
            } else {
                for (uint32_t a = 0; a < d.n_targets; a++) {
                    const orc_target& tg = d.targets[a];
                    blend_store(d, tg, (size_t)j * tg.width + (size_t)i, out[a < (uint32_t)sh.n_outputs ? a : sh.n_outputs - 1]);
                }
            }
            frags++;
        }
    }
}

} // namespace

extern "C" int orc_render(orc_draw* d) {
    if (d->shader < 0 || d->shader >= (int)(sizeof SHADERS / sizeof SHADERS[0])) return -1;
    if (d->n_targets == 0) return 0;            // reference src/render_pass.rs:26: no targets, empty command buffer
    if (d->n_targets > 4) return -2;
    const Shader& sh = SHADERS[d->shader];
    const uint32_t S = d->samples > 1 ? d->samples : 1;
    if (S > 1 && ((S != 2 && S != 4 && S != 8) || d->n_targets != 1 || !d->msaa)) return -4;
    const int W = (int)d->targets[0].width, H = (int)d->targets[0].height;
    for (uint32_t a = 1; a < d->n_targets; a++)
        if ((int)d->targets[a].width != W || (int)d->targets[a].height != H) return -3;

    float vp[4] = { 0.0f, 0.0f, (float)W, (float)H };
    if (d->has_viewport) std::memcpy(vp, d->viewport, sizeof vp);
    ViewXf xf; xf.hw = vp[2] * 0.5f; xf.hh = vp[3] * 0.5f; xf.ox = vp[0] + xf.hw; xf.oy = vp[1] + xf.hh;
    Scissor sc = { 0, 0, W, H };
    if (d->clip_to_viewport) {
        // pixel centre (i + 0.5) inside [vx, vx + vw)
        sc.x0 = std::max(0, (int)std::ceil((double)vp[0] - 0.5));
        sc.y0 = std::max(0, (int)std::ceil((double)vp[1] - 0.5));
        sc.x1 = std::min(W, (int)std::ceil((double)vp[0] + (double)vp[2] - 0.5));
        sc.y1 = std::min(H, (int)std::ceil((double)vp[1] + (double)vp[3] - 0.5));
    }

    const uint32_t tpi = tris_per_instance(*d);
    int nthreads = d->threads > 0 ? d->threads : (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    nthreads = std::min(nthreads, std::max(1, H / 8));

    // Set up every triangle once (shared by all bands), in primitive order.
    std::vector<Tri> tris;
    const bool empty_scissor = sc.x0 >= sc.x1 || sc.y0 >= sc.y1;
    if (!empty_scissor && tpi > 0) {
        tris.reserve((size_t)d->instance_count * tpi);
        std::vector<Vert> verts(d->vertex_count);
        for (uint32_t inst = 0; inst < d->instance_count; inst++) {
            for (uint32_t v = 0; v < d->vertex_count; v++) verts[v] = run_vertex(*d, sh, xf, v, inst);
            for (uint32_t i = 0; i < tpi; i++) {
                uint32_t idx[3]; tri_vertices(*d, i, idx);
                Tri t;
                const Vert& a = verts[idx[0]]; const Vert& b = verts[idx[1]]; const Vert& c = verts[idx[2]];
                if (a.ok && b.ok && c.ok) {
                    if (setup_tri(sh, a, b, c, sc, S > 1 ? 128 : 0, t)) tris.push_back(t);
                } else {
                    Vert poly[CLIP_MAX_VERTS];
                    const int n = clip_triangle(sh, xf, a, b, c, poly);
                    for (int f = 2; f < n; f++)
                        if (setup_tri(sh, poly[0], poly[f - 1], poly[f], sc, S > 1 ? 128 : 0, t)) tris.push_back(t);
                }
            }
        }
    }

    // Row bands: each thread owns a band of rows and walks every triangle in order, so per-pixel blend order is
    // primitive order regardless of the thread count.
    std::vector<uint64_t> frags(nthreads, 0);
    auto work = [&](int tid) {
        int by0 = (int)((int64_t)H * tid / nthreads), by1 = (int)((int64_t)H * (tid + 1) / nthreads);
        if (S > 1) clear_rows(d->targets[0], by0, by1, d->msaa, S);
        else for (uint32_t a = 0; a < d->n_targets; a++) clear_rows(d->targets[a], by0, by1);
        uint64_t n = 0;
        for (const Tri& t : tris) {
            if (t.y1 < by0 || t.y0 >= by1) continue;
            shade_tri(*d, sh, t, by0, by1, n);
        }
        if (S > 1) resolve_rows(d->targets[0], d->msaa, S, by0, by1);      // the resolve overwrites the whole target
        frags[tid] = n;
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; t++) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    uint64_t total = 0; for (uint64_t n : frags) total += n;
    d->fragments_out = total;
    return 0;
}

extern "C" int orc_abi_version(void) { return 4; }

// test hook: the f32 -> f16 conversion used for RGBA16F targets
extern "C" uint16_t orc_float_to_half(float f) { return float_to_half(f); }
