// Hardware probe with inline asm: issue cost per wave-instruction per SIMD, all SIMDs loaded with 8 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BODY(INS) asm volatile(REP8(REP8(INS "\n")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q) : "v"(e), "v"(f), "s"(s));
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = 1.0001f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p = {a, b}, q = {c, d};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) BODY("v_fma_f32 %0, %6, %7, %0\n v_fma_f32 %1, %6, %7, %1\n v_fma_f32 %2, %6, %7, %2\n v_fma_f32 %3, %6, %7, %3")
        if (MODE == 1) BODY("v_pk_fma_f32 %4, %4, %4, %5\n v_pk_fma_f32 %5, %5, %5, %4\n v_pk_fma_f32 %4, %4, %5, %5\n v_pk_fma_f32 %5, %5, %4, %4")
        if (MODE == 2) BODY("v_cvt_f32_ubyte0 %0, %6\n v_cvt_f32_ubyte1 %1, %6\n v_cvt_f32_ubyte2 %2, %6\n v_cvt_f32_ubyte3 %3, %6")
        if (MODE == 3) BODY("v_mad_i32_i24 %0, %6, %7, %0\n v_mad_i32_i24 %1, %6, %7, %1\n v_mad_i32_i24 %2, %6, %7, %2\n v_mad_i32_i24 %3, %6, %7, %3")
        if (MODE == 4) BODY("v_mul_lo_u32 %0, %6, %7\n v_mul_lo_u32 %1, %6, %7\n v_mul_lo_u32 %2, %6, %7\n v_mul_lo_u32 %3, %6, %7")
        if (MODE == 5) BODY("v_cvt_pk_u8_f32 %0, %6, 0, %0\n v_cvt_pk_u8_f32 %1, %6, 1, %1\n v_cvt_pk_u8_f32 %2, %6, 2, %2\n v_cvt_pk_u8_f32 %3, %6, 3, %3")
        if (MODE == 6) BODY("v_add_f32 %0, %6, %0\n v_add_f32 %1, %6, %1\n v_add_f32 %2, %6, %2\n v_add_f32 %3, %6, %3")
        if (MODE == 7) BODY("v_cndmask_b32 %0, %6, %7, vcc\n v_cndmask_b32 %1, %6, %7, vcc\n v_cndmask_b32 %2, %6, %7, vcc\n v_cndmask_b32 %3, %6, %7, vcc")
        if (MODE == 8) BODY("v_cvt_i32_f32 %0, %6\n v_floor_f32 %1, %6\n v_cvt_f32_i32 %2, %6\n v_med3_f32 %3, %6, %7, %3")
        if (MODE == 9) BODY("v_pk_add_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %4\n v_pk_add_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %4")
        if (MODE == 10) BODY("v_bfe_u32 %0, %6, 8, 8\n v_and_b32 %1, 255, %6\n v_lshrrev_b32 %2, 24, %6\n v_or_b32 %3, %6, %7")
        if (MODE == 11) BODY("v_readlane_b32 s20, %6, 3\n v_readlane_b32 s21, %7, 5\n v_readlane_b32 s22, %6, 7\n v_readlane_b32 s23, %7, 9")
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + p.x + p.y + q.x + q.y;
}
template <int MODE> void run(const char* name, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400, grid = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0001f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)grid * 4 * iters * 64 * 4;
    printf("%-44s %8.3f ms  %6.2f cycles/wave-instr/SIMD (if 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<6>("v_add_f32", out); run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<9>("v_pk_add/mul_f32", out);
    run<2>("v_cvt_f32_ubyteN", out); run<3>("v_mad_i32_i24", out); run<4>("v_mul_lo_u32", out); run<5>("v_cvt_pk_u8_f32", out);
    run<7>("v_cndmask_b32", out); run<8>("cvt_i32_f32/floor/cvt_f32_i32/med3", out); run<10>("bfe/and/lshr/or", out);
    run<11>("v_readlane_b32", out);
    return 0;
}
