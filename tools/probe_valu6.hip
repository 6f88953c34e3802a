// Hardware probe: which instructions that read 8- or 16-bit texel data issue at the rate of f32 add/mul/fma (about 2.3
// cycles per wave-instruction with several waves on a SIMD) and which at the rate of conversions and integer work (about 4)?
// Candidates for a cheaper bilinear tap: v_fma_mix_f32 (an f32 fma whose sources may be f16 halves of a register: texels kept
// as f16 need no conversion instruction), v_dot4_u32_u8 and v_dot2_f32_f16 (a four-tap weighted sum in one or two
// instructions), v_pk_fma_f16, v_perm_b32.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_valu6.hip -o tools/probe_valu6 && tools/probe_valu6
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BODY(INS) asm volatile(REP8(REP8(INS "\n")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(s) : "vcc");
#define FMA4 "v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3"
#define CVT4 "v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4"
// D = fma(f32 %4, f16 hi half of %5, f16 lo half of %5)
#define MIX(D) "v_fma_mix_f32 " D ", %4, %5, %5 op_sel:[0,1,0] op_sel_hi:[0,1,1]"
#define MIX4 MIX("%0") "\n" MIX("%1") "\n" MIX("%2") "\n" MIX("%3")
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = 1.0001f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) BODY(FMA4)
        if (MODE == 1) BODY(CVT4)
        if (MODE == 2) BODY(MIX4)
        if (MODE == 3) BODY(MIX("%0") "\n v_fma_f32 %1, %4, %5, %1\n" MIX("%2") "\n v_fma_f32 %3, %4, %5, %3")
        if (MODE == 4) BODY(MIX("%0") "\n v_cvt_f32_ubyte0 %1, %4\n" MIX("%2") "\n v_cvt_f32_ubyte1 %3, %5")
        if (MODE == 5) BODY("v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3")
        if (MODE == 6) BODY("v_dot4_u32_u8 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3")
        if (MODE == 7) BODY("v_dot2_f32_f16 %0, %4, %5, %0\n v_dot2_f32_f16 %1, %4, %5, %1\n v_dot2_f32_f16 %2, %4, %5, %2\n v_dot2_f32_f16 %3, %4, %5, %3")
        if (MODE == 8) BODY("v_dot2_f32_f16 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_dot2_f32_f16 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3")
        if (MODE == 9) BODY("v_pk_fma_f16 %0, %4, %5, %0\n v_pk_fma_f16 %1, %4, %5, %1\n v_pk_fma_f16 %2, %4, %5, %2\n v_pk_fma_f16 %3, %4, %5, %3")
        if (MODE == 10) BODY("v_pk_fma_f16 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_pk_fma_f16 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3")
        if (MODE == 11) BODY("v_perm_b32 %0, %4, %5, %0\n v_perm_b32 %1, %4, %5, %1\n v_perm_b32 %2, %4, %5, %2\n v_perm_b32 %3, %4, %5, %3")
        if (MODE == 12) BODY("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %5\n v_cvt_f32_f16 %2, %4\n v_cvt_f32_f16 %3, %5")
        if (MODE == 13) BODY("v_cvt_pk_u8_f32 %0, %4, 0, %0\n v_cvt_pk_u8_f32 %1, %4, 1, %1\n v_cvt_pk_u8_f32 %2, %4, 2, %2\n v_cvt_pk_u8_f32 %3, %4, 3, %3")
        if (MODE == 14) BODY("v_fma_mixlo_f16 %0, %4, %5, %0\n v_fma_mixlo_f16 %1, %4, %5, %1\n v_fma_mixlo_f16 %2, %4, %5, %2\n v_fma_mixlo_f16 %3, %4, %5, %3")
        if (MODE == 15) BODY("v_floor_f32 %0, %4\n v_floor_f32 %1, %5\n v_floor_f32 %2, %4\n v_floor_f32 %3, %5")
        if (MODE == 16) BODY("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3")
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
template <int MODE> void run(const char* name, float* out, int grid) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0001f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double wave_instr_per_simd = (double)grid / 256.0 * iters * 64 * 4;      // one wave of every block per SIMD
    printf("%-44s %2d waves/SIMD %8.3f ms  %6.2f cycles/wave-instr/SIMD (if 2.4 GHz)\n", name, grid / 256, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int waves : {6, 2, 1}) {
        const int g = 256 * waves;
        run<0>("fma x4", out, g); run<1>("cvt_f32_ubyte x4", out, g); run<2>("fma_mix x4", out, g);
        run<3>("fma_mix fma fma_mix fma", out, g); run<4>("fma_mix cvt fma_mix cvt", out, g);
        run<5>("dot4_u32_u8 x4", out, g); run<6>("dot4 fma dot4 fma", out, g);
        run<7>("dot2_f32_f16 x4", out, g); run<8>("dot2 fma dot2 fma", out, g);
        run<9>("pk_fma_f16 x4", out, g); run<10>("pk_fma_f16 fma pk_fma_f16 fma", out, g);
        run<11>("perm_b32 x4", out, g); run<12>("cvt_f32_f16 x4", out, g); run<13>("cvt_pk_u8_f32 x4", out, g);
        run<14>("fma_mixlo_f16 x4", out, g); run<15>("floor x4", out, g); run<16>("mad_u32_u24 x4", out, g);

    }
    return 0;
}
