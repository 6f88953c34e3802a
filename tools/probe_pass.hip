// Probe: the shading pass of the tile kernel on its own (plane equations from LDS, bilinear sample of an RGBA8 texture,
// blend into an LDS pixel), every SIMD loaded with `waves` waves that all have work all the time.  Tells what the pass
// costs when nothing but the pass runs: no binning lists, no barriers, no uneven regions.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I renderer_amd/csrc/shader tools/probe_pass.hip -o tools/probe_pass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "rh_shader.h"

using namespace rh;

template <int BLEND>
__device__ __forceinline__ uint32_t blend8(uint32_t dst, float4 s) {
    const float df = BLEND == 1 ? 1.0f - s.w : 1.0f;
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

// MODE 6: the whole pass with the re-balanced sample (see sample_rebalanced); 7: the same without fetch and LDS pixel
// MODE 0: the whole pass; 1: no texture fetch (texels from registers); 2: no LDS pixel (blend into a register);
// 3: neither; 4: sample only (no blend); 5: blend only (constant colour)
__device__ __forceinline__ float ub0(uint32_t v) { float r; asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ float ub1(uint32_t v) { float r; asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ float ub2(uint32_t v) { float r; asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ float ub3(uint32_t v) { float r; asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(v)); return r; }

// Same values as Texture2D::sample for an RGBA8 texture of at least 2x2 texels and under 16 MiB, with fewer
// instructions outside the f32 add/mul/fma class: one conversion per texel byte and f32 differences, med3 clamps,
// the texel offset computed in f32 (exact below 2^24).
template <bool FETCH>
__device__ __forceinline__ float4 sample_rebalanced(const uint8_t* data, float Wf, float Hf, float pitch_f, uint32_t pitch, float2 uv, uint32_t fake) {
    const float u = __builtin_fmaf(uv.x, Wf, -0.5f), v = __builtin_fmaf(uv.y, Hf, -0.5f);
    const float ibf = floorf(__builtin_amdgcn_fmed3f(u, 0.0f, Wf - 2.0f)), jbf = floorf(__builtin_amdgcn_fmed3f(v, 0.0f, Hf - 2.0f));
    const float a = __builtin_amdgcn_fmed3f(u - ibf, 0.0f, 1.0f), b = __builtin_amdgcn_fmed3f(v - jbf, 0.0f, 1.0f);
    const uint32_t off = (uint32_t)__builtin_fmaf(jbf, pitch_f, ibf * 4.0f);
    uint32_t r0x, r0y, r1x, r1y;
    if (FETCH) {
        const u32x2_a4 r0 = *reinterpret_cast<const u32x2_a4*>(data + off);
        const u32x2_a4 r1 = *reinterpret_cast<const u32x2_a4*>(data + pitch + off);
        r0x = r0.x; r0y = r0.y; r1x = r1.x; r1y = r1.y;
    } else { r0x = off * 0x01010101u + fake; r0y = r0x ^ 0x55aa55aau; r1x = off * 0x00010101u; r1y = r1x + 0x10203040u; }
    float4 c;
#define LERPB(ch, UB) { const float p00 = UB(r0x), p10 = UB(r0y), p01 = UB(r1x), p11 = UB(r1y); \
                        const float top = __builtin_fmaf(a, p10 - p00, p00); const float bot = __builtin_fmaf(a, p11 - p01, p01); \
                        c.ch = __builtin_fmaf(b, bot, __builtin_fmaf(-b, top, top)) * INV_255; }
    LERPB(x, ub0) LERPB(y, ub1) LERPB(z, ub2) LERPB(w, ub3)
#undef LERPB
    return c;
}

template <int BLEND>
__device__ __forceinline__ uint32_t blend8b(uint32_t dst, float4 s) {
    const float df = BLEND == 1 ? 1.0f - s.w : 1.0f;
    const float o0 = __builtin_fmaf(ub0(dst), df, s.x * 255.0f), o1 = __builtin_fmaf(ub1(dst), df, s.y * 255.0f);
    const float o2 = __builtin_fmaf(ub2(dst), df, s.z * 255.0f), o3 = __builtin_fmaf(ub3(dst), df, s.w * 255.0f);
    uint32_t q = 0;
    q = __builtin_amdgcn_cvt_pk_u8_f32(o0, 0, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o1, 1, q);
    q = __builtin_amdgcn_cvt_pk_u8_f32(o2, 2, q); q = __builtin_amdgcn_cvt_pk_u8_f32(o3, 3, q);
    return q;
}

template <int MODE>
__global__ __launch_bounds__(256, 8) void k(const uint8_t* tex, uint32_t W, uint32_t H, int passes, float* out, float jitter) {
    __shared__ float rec[64][12];
    __shared__ uint32_t px[4][512 + 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) {
        rec[tid][0] = 0.0f; rec[tid][1] = 0.0f;                  // XR, YR
        rec[tid][2] = 0.001f * tid; rec[tid][3] = 0.002f * tid;   // u0, v0
        rec[tid][4] = 1.0f / 4096.0f; rec[tid][5] = 0.0f; rec[tid][6] = 0.0f; rec[tid][7] = 1.0f / 4096.0f;
    }
    for (int i = tid; i < 4 * 528; i += 256) (&px[0][0])[i] = 0xff000000u;
    __syncthreads();
    Texture2D t{ tex, W, H, 1u, FORMAT_RGBA_U8, 1u };
    Sampler smp{ FILTER_LINEAR, 1u + FILTER_LINEAR };
    const float lane_f = (float)lane;
    uint32_t acc = 0xff000000u;
    float4 keep = make_float4(0, 0, 0, 0);
    for (int p = 0; p < passes; p++) {
        const int r = (p * 7 + wave) & 63;                        // wave-uniform record index
        const float wf = 16.0f, invw = 1.0f / 16.0f;
        const float idx_f = lane_f + (float)((p & 3) * 64);
        const float row_f = floorf(__builtin_fmaf(idx_f, invw, 0.5f * invw));
        const float col_f = __builtin_fmaf(-row_f, wf, idx_f);
        const float lxf = col_f + (float)(p & 15), lyf = row_f;
        const int pidx = (int)__builtin_fmaf(lyf, 32.0f, lxf);
        const float* tp = rec[r];
        const float dx = __builtin_fmaf(lxf, 256.0f, -tp[0]), dy = __builtin_fmaf(lyf, 256.0f, -tp[1]);
        float2 uv;
        uv.x = __builtin_fmaf(tp[6], dy, __builtin_fmaf(tp[4], dx, tp[2])) + jitter;
        uv.y = __builtin_fmaf(tp[7], dy, __builtin_fmaf(tp[5], dx, tp[3]));
        float4 c;
        if (MODE == 1 || MODE == 3) {
            // same arithmetic, the four texels made up from registers
            const float u = __builtin_fmaf(uv.x, (float)W, -0.5f), v = __builtin_fmaf(uv.y, (float)H, -0.5f);
            const float ibf = floorf(fminf(fmaxf(u, 0.0f), (float)(W - 2))), jbf = floorf(fminf(fmaxf(v, 0.0f), (float)(H - 2)));
            const float a = fminf(fmaxf(u - ibf, 0.0f), 1.0f), b = fminf(fmaxf(v - jbf, 0.0f), 1.0f);
            const uint32_t r0x = (uint32_t)(int)ibf * 0x01010101u + acc, r0y = r0x ^ 0x55aa55aau, r1x = (uint32_t)(int)jbf * 0x01010101u, r1y = r1x + 0x10203040u;
#define UNP(t) make_float4((float)((t) & 0xffu), (float)(((t) >> 8) & 0xffu), (float)(((t) >> 16) & 0xffu), (float)((t) >> 24))
            const float4 t00 = UNP(r0x), t10 = UNP(r0y), t01 = UNP(r1x), t11 = UNP(r1y);
#define LERP2(ch) { const float top = __builtin_fmaf(a, t10.ch - t00.ch, t00.ch); const float bot = __builtin_fmaf(a, t11.ch - t01.ch, t01.ch); \
                    c.ch = __builtin_fmaf(b, bot, __builtin_fmaf(-b, top, top)) * INV_255; }
            LERP2(x) LERP2(y) LERP2(z) LERP2(w)
        } else if (MODE == 6 || MODE == 7) {
            c = sample_rebalanced<MODE == 6>(tex, (float)W, (float)H, (float)(W * 4), W * 4, uv, acc);
        } else if (MODE == 5) {
            c = make_float4(uv.x, uv.y, 0.25f, 0.8f);
        } else {
            c = t.sample(smp, uv);
        }
        if (MODE == 4) { keep.x += c.x; keep.y += c.y; keep.z += c.z; keep.w += c.w; }
        else if (MODE == 2 || MODE == 3) acc = blend8<1>(acc, c);
        else if (MODE == 7) acc = blend8b<1>(acc, c);
        else if (MODE == 6) { const int pb = (int)__builtin_fmaf(lyf, 128.0f, lxf * 4.0f) & 2047; uint32_t* d = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(&px[wave][0]) + pb); *d = blend8b<1>(*d, c); }
        else { uint32_t* d = &px[wave][pidx & 511]; *d = blend8<1>(*d, c); }
    }
    out[blockIdx.x * 256 + tid] = __uint_as_float(acc) + px[wave][lane] + keep.x + keep.y + keep.z + keep.w;
}

template <int MODE> void run(const char* name, const uint8_t* tex, uint32_t W, float* out, int grid) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int passes = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, tex, W, W, 10, out, 0.0f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, tex, W, W, passes, out, 0.0f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double passes_per_simd = (double)grid / 256.0 * passes;
    printf("%-46s tex %3u  %d waves/SIMD %8.3f ms  %7.1f cycles/pass/SIMD (if 2.4 GHz)  %6.1f ns/pass/SIMD\n", name, W, grid / 256, ms,
           ms * 1e-3 * 2.4e9 / passes_per_simd, ms * 1e6 / passes_per_simd);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (uint32_t W : {256u, 16u}) {
        std::vector<uint8_t> h((size_t)W * W * 4);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 37 + (i >> 9));
        uint8_t* tex; (void)hipMalloc(&tex, h.size()); (void)hipMemcpy(tex, h.data(), h.size(), hipMemcpyHostToDevice);
        for (int waves : {8, 4, 2, 1}) {
            const int g = 256 * waves;
            run<0>("whole pass", tex, W, out, g);
            run<1>("texels from registers", tex, W, out, g);
            run<2>("blend into a register", tex, W, out, g);
            run<3>("no fetch, no LDS pixel", tex, W, out, g);
            run<4>("sample only", tex, W, out, g);
            run<5>("blend only", tex, W, out, g);
            run<6>("whole pass, re-balanced", tex, W, out, g);
            run<7>("no fetch, no LDS pixel, re-balanced", tex, W, out, g);
        }
        (void)hipFree(tex);
    }
    return 0;
}
