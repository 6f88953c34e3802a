// Hardware probe: issue cost of v_fma_f32, v_pk_fma_f32, v_cvt_f32_ubyte, v_mad_i32_i24, v_mul_lo_u32, v_cvt_pk_u8_f32
// with every SIMD loaded (grid = 256 CUs x 8 blocks of 256 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 + 77, u2 = u0 ^ 0x1234, u3 = u0 * 3;
    int i0 = u0 & 0xffff, i1 = u1 & 0xffff, i2 = u2 & 0xffff, i3 = u3 & 0xffff;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (MODE == 0) { a0 = __builtin_fmaf(a0, s, a1); a1 = __builtin_fmaf(a1, s, a2); a2 = __builtin_fmaf(a2, s, a3); a3 = __builtin_fmaf(a3, s, a0);
                             a4 = __builtin_fmaf(a4, s, a5); a5 = __builtin_fmaf(a5, s, a6); a6 = __builtin_fmaf(a6, s, a7); a7 = __builtin_fmaf(a7, s, a4); }
            if (MODE == 1) { v2f vs = {s, s}; p0 = __builtin_elementwise_fma(p0, vs, p1); p1 = __builtin_elementwise_fma(p1, vs, p2);
                             p2 = __builtin_elementwise_fma(p2, vs, p3); p3 = __builtin_elementwise_fma(p3, vs, p0);
                             p0 = __builtin_elementwise_fma(p0, vs, p2); p1 = __builtin_elementwise_fma(p1, vs, p3);
                             p2 = __builtin_elementwise_fma(p2, vs, p0); p3 = __builtin_elementwise_fma(p3, vs, p1); }
            if (MODE == 2) { a0 += (float)(u0 & 0xff); a1 += (float)((u1 >> 8) & 0xff); a2 += (float)((u2 >> 16) & 0xff); a3 += (float)(u3 >> 24);
                             u0 += 0x01010101u; u1 += 0x01010101u; u2 += 0x01010101u; u3 += 0x01010101u; }   // 4 cvt + 4 add f32 + 4 add u32
            if (MODE == 3) { i0 = __mul24(i0, i1) + i2; i1 = __mul24(i1, i2) + i3; i2 = __mul24(i2, i3) + i0; i3 = __mul24(i3, i0) + i1;
                             i0 &= 0xffff; i1 &= 0xffff; i2 &= 0xffff; i3 &= 0xffff; }                        // 4 mad24 + 4 and
            if (MODE == 4) { u0 = u0 * u1 + 1; u1 = u1 * u2 + 1; u2 = u2 * u3 + 1; u3 = u3 * u0 + 1; }       // 4 mul_lo(+add)
            if (MODE == 5) { u0 = __builtin_amdgcn_cvt_pk_u8_f32(a0, 0, u0); u1 = __builtin_amdgcn_cvt_pk_u8_f32(a1, 1, u1);
                             u2 = __builtin_amdgcn_cvt_pk_u8_f32(a2, 2, u2); u3 = __builtin_amdgcn_cvt_pk_u8_f32(a3, 3, u3);
                             a0 += 1.0f; a1 += 1.0f; a2 += 1.0f; a3 += 1.0f; }                               // 4 cvt_pk + 4 add
            if (MODE == 6) { a0 += 1.0f; a1 += 1.0f; a2 += 1.0f; a3 += 1.0f; a4 += 1.0f; a5 += 1.0f; a6 += 1.0f; a7 += 1.0f; }   // 8 add
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 + u1 + u2 + u3) + (float)(i0 + i1 + i2 + i3);
}
template <int MODE> void run(const char* name, int ops_per_rep, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200, grid = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0001f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)grid * 4 * iters * REP * ops_per_rep;
    // cycles per wave-instruction per SIMD at 2.4 GHz: time * 2.4e9 * 1024 SIMDs / wave_instr
    printf("%-28s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32 x8", 8, out);
    run<1>("v_pk_fma_f32 x8", 8, out);
    run<2>("cvt_ubyte+add+iadd x12", 12, out);
    run<3>("mad_i24+and x8", 8, out);
    run<4>("mul_lo_u32(+add) x4(8)", 8, out);
    run<5>("cvt_pk_u8+add x8", 8, out);
    run<6>("v_add_f32 x8", 8, out);
    return 0;
}
