// Hardware probe: how do f32 add/mul/fma instructions (about 2.5 cycles per wave-instruction when issued alone) and every
// other vector instruction (about 4) share a SIMD?  Mixes inside one wave's stream, dependent chains, one wave per SIMD,
// and waves of two kinds (pure fma / pure conversion) side by side on a SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_valu4.hip -o tools/probe_valu4 && tools/probe_valu4
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BODY(INS) asm volatile(REP8(REP8(INS "\n")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(s) : "vcc");
#define FMA4 "v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3"
#define CVT4 "v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4"
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = 1.0001f;
    const int wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) BODY(FMA4)
        if (MODE == 1) BODY(CVT4)
        if (MODE == 2) BODY("v_fma_f32 %0, %4, %5, %0\n v_cvt_f32_ubyte0 %1, %4\n v_fma_f32 %2, %4, %5, %2\n v_cvt_f32_ubyte1 %3, %5")
        if (MODE == 3) BODY("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_cvt_f32_ubyte1 %3, %5")
        if (MODE == 4) BODY("v_fma_f32 %0, %4, %5, %0\n v_cvt_f32_ubyte0 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte1 %3, %5")
        if (MODE == 5) BODY("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %0, %4, %5, %0")      // dependent chain
        if (MODE == 6) BODY("v_fma_f32 %0, %4, %5, %0\n v_cvt_f32_ubyte0 %0, %0\n v_fma_f32 %0, %4, %5, %0\n v_cvt_f32_ubyte1 %0, %0")     // dependent, mixed
        if (MODE == 7) BODY("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_cvt_f32_ubyte0 %2, %4\n v_cvt_f32_ubyte1 %3, %5")     // ffcc
        if (MODE == 8) { if (wave & 1) BODY(FMA4) else BODY(CVT4) }                                       // waves of two kinds
        if (MODE == 9) BODY("v_cvt_f32_ubyte0 %0, %4\n v_and_b32 %1, %4, %5\n v_cvt_f32_ubyte2 %2, %4\n v_add_u32 %3, %4, %5")        // two "other" kinds
        if (MODE == 10) BODY("v_mul_f32 %0, %4, %0\n v_max_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_min_f32 %3, %4, %3")
        if (MODE == 11) BODY("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %0, %5, %1\n v_fma_f32 %2, %1, %5, %2\n v_fma_f32 %3, %2, %5, %3")   // each depends on the previous
        if (MODE == 12) BODY("v_fma_f32 %0, %4, %5, %0\n v_cvt_f32_ubyte0 %1, %0\n v_fma_f32 %2, %1, %5, %2\n v_cvt_f32_ubyte1 %3, %2")   // mixed, each depends on the previous
        if (MODE == 13) BODY("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte0 %1, %0\n v_cvt_f32_ubyte0 %2, %1\n v_cvt_f32_ubyte1 %3, %2")     // conversions, each depends on the previous
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
    printf("%-62s %2d waves/SIMD %8.3f ms  %6.2f cycles/wave-instr/SIMD (if 2.4 GHz)\n", name, grid / 256, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int waves : {8, 2, 1}) {
        const int g = 256 * waves;
        run<0>("fma fma fma fma (independent)", out, g); run<1>("cvt cvt cvt cvt (independent)", out, g); run<2>("fma cvt fma cvt", out, g);
        run<7>("fma fma cvt cvt", out, g); run<3>("fma fma fma cvt", out, g); run<4>("fma cvt cvt cvt", out, g);
        run<5>("fma chain on one register", out, g); run<11>("fma, each depends on the previous", out, g);
        run<6>("fma cvt chain on one register", out, g); run<12>("fma cvt, each depends on the previous", out, g); run<13>("cvt, each depends on the previous", out, g);
        run<8>("odd waves fma, even waves cvt", out, g); run<9>("cvt and cvt add (two other kinds)", out, g); run<10>("mul max mul min", out, g);
    }
    return 0;
}
