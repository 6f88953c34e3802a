// Hardware probe with inline asm: issue cost per wave-instruction per SIMD for the instructions a cheaper fragment
// formulation could use (mixed-precision fma, med3, permutes, byte dot products, DPP moves), all SIMDs loaded with 8 waves.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_valu3.hip -o tools/probe_valu3 && tools/probe_valu3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BODY(INS) asm volatile(REP8(REP8(INS "\n")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(s) : "vcc");
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = 1.0001f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) BODY("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3")
        if (MODE == 1) BODY("v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %5, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %4, %5, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
        if (MODE == 2) BODY("v_med3_f32 %0, %4, %5, %0\n v_med3_f32 %1, %4, %5, %1\n v_med3_f32 %2, %4, %5, %2\n v_med3_f32 %3, %4, %5, %3")
        if (MODE == 3) BODY("v_max_f32 %0, %4, %0\n v_min_f32 %1, %4, %1\n v_max_f32 %2, %4, %2\n v_min_f32 %3, %4, %3")
        if (MODE == 4) BODY("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_sub_f32 %2, %4, %2\n v_sub_f32 %3, %4, %3")
        if (MODE == 5) BODY("v_floor_f32 %0, %4\n v_floor_f32 %1, %5\n v_rndne_f32 %2, %4\n v_rndne_f32 %3, %5")
        if (MODE == 6) BODY("v_cvt_i32_f32 %0, %4\n v_cvt_u32_f32 %1, %5\n v_cvt_f32_u32 %2, %4\n v_cvt_f32_i32 %3, %5")
        if (MODE == 7) BODY("v_perm_b32 %0, %4, %5, %0\n v_perm_b32 %1, %4, %5, %1\n v_perm_b32 %2, %4, %5, %2\n v_perm_b32 %3, %4, %5, %3")
        if (MODE == 8) BODY("v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3")
        if (MODE == 9) BODY("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_lshl_add_u32 %2, %4, 2, %2\n v_lshl_add_u32 %3, %4, 2, %3")
        if (MODE == 10) BODY("v_cmp_gt_f32 vcc, %4, %0\n v_cmp_gt_f32 vcc, %4, %1\n v_cmp_lt_i32 vcc, %4, %2\n v_cmp_lt_i32 vcc, %4, %3")
        if (MODE == 11) BODY("v_mov_b32_dpp %0, %4 row_shr:1\n v_mov_b32_dpp %1, %5 row_shr:2\n v_mov_b32_dpp %2, %4 row_shr:3\n v_mov_b32_dpp %3, %5 row_shr:4")
        if (MODE == 12) BODY("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %5\n v_cvt_f32_f16 %2, %4\n v_cvt_f32_f16 %3, %5")
        if (MODE == 13) BODY("v_pk_fma_f16 %0, %4, %5, %0\n v_pk_fma_f16 %1, %4, %5, %1\n v_pk_fma_f16 %2, %4, %5, %2\n v_pk_fma_f16 %3, %4, %5, %3")
        if (MODE == 14) BODY("v_fma_f32 %0, %4, %5, %0\n v_cvt_f32_ubyte0 %1, %4\n v_fma_f32 %2, %4, %5, %2\n v_cvt_f32_ubyte1 %3, %5")
        if (MODE == 15) BODY("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4")
        if (MODE == 16) BODY("ds_bpermute_b32 %0, %4, %5\n ds_bpermute_b32 %1, %4, %5\n ds_bpermute_b32 %2, %4, %5\n ds_bpermute_b32 %3, %4, %5\n s_waitcnt lgkmcnt(0)")
        if (MODE == 17) BODY("v_fma_f32 %0, %4, %5, %0\n v_mov_b32 %1, %4\n v_fma_f32 %2, %4, %5, %2\n v_mov_b32 %3, %5")
        if (MODE == 18) BODY("v_cvt_pk_u8_f32 %0, %4, 0, %0\n v_cvt_pk_u8_f32 %1, %4, 1, %1\n v_cvt_pk_u8_f32 %2, %4, 2, %2\n v_cvt_pk_u8_f32 %3, %4, 3, %3")
        if (MODE == 19) BODY("v_mad_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,0,0]\n s_nop 0\n s_nop 0\n s_nop 0")
        if (MODE == 20) BODY("v_fma_f32 %0, %4, %5, %0\n s_mul_i32 s20, s21, s22\n v_fma_f32 %2, %4, %5, %2\n s_add_u32 s23, s21, s22")
        if (MODE == 21) BODY("v_cvt_pkrtz_f16_f32 %0, %4, %5\n v_cvt_pkrtz_f16_f32 %1, %4, %5\n v_cvt_pkrtz_f16_f32 %2, %4, %5\n v_cvt_pkrtz_f16_f32 %3, %4, %5")
        if (MODE == 22) BODY("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_sub_u32 %2, %4, %2\n v_max_i32 %3, %4, %3")
        if (MODE == 23) BODY("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5")
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
template <int MODE> void run(const char* name, float* out, int per_body = 4) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400, grid = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0001f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)grid * 4 * iters * 64 * per_body;
    printf("%-52s %8.3f ms  %6.2f cycles/wave-instr/SIMD (if 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", out); run<23>("v_fmac_f32 (VOP2)", out); run<1>("v_fma_mix_f32 (f16 src0)", out); run<2>("v_med3_f32", out); run<3>("v_max_f32/v_min_f32", out);
    run<4>("v_mul_f32/v_sub_f32", out); run<5>("v_floor_f32/v_rndne_f32", out); run<6>("v_cvt i32<->f32", out); run<7>("v_perm_b32", out);
    run<8>("v_dot4_u32_u8", out); run<9>("v_mad_u32_u24/v_lshl_add_u32", out); run<10>("v_cmp f32/i32", out); run<11>("v_mov_b32_dpp row_shr", out);
    run<12>("v_cvt_f32_f16", out); run<13>("v_pk_fma_f16", out); run<15>("v_cvt_f32_ubyteN", out); run<14>("fma + cvt_ubyte interleaved", out);
    run<17>("fma + v_mov interleaved", out); run<18>("v_cvt_pk_u8_f32", out); run<16>("ds_bpermute_b32 x4 + wait", out);
    run<20>("fma + salu interleaved (per 4: 2 fma 2 salu)", out); run<21>("v_cvt_pkrtz_f16_f32", out); run<22>("v_add_u32/v_sub_u32/v_max_i32", out);
    return 0;
}
