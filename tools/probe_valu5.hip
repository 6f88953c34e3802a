// Probe: which ingredient of a real kernel's instruction stream stops a SIMD from overlapping an f32 fma of one wave
// with another wave's instruction?  Base stream: fma cvt fma cvt (overlaps: ~2.4 cycles per instruction at 8 waves/SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_valu5.hip -o tools/probe_valu5 && tools/probe_valu5
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BODY(INS) asm volatile(REP8(REP8(INS "\n")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(s), "v"(ldsaddr) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "memory");
#define MIX "v_fma_f32 %0, %4, %5, %0\n v_cvt_f32_ubyte0 %1, %4\n v_fma_f32 %2, %4, %5, %2\n v_cvt_f32_ubyte1 %3, %5"
template <int MODE, int BLOCK> __global__ __launch_bounds__(BLOCK) void k(float* out, int iters, float s) {
    __shared__ float lds[1024];
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = 1.0001f;
    const uint32_t ldsaddr = (threadIdx.x & 255) * 4;
    lds[threadIdx.x & 1023] = a;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) BODY(MIX)
        if (MODE == 1) BODY("v_fma_f32 %0, %4, %5, %0\n s_add_u32 s20, s21, s22\n v_cvt_f32_ubyte0 %1, %4\n s_and_b32 s23, s21, s22\n v_fma_f32 %2, %4, %5, %2\n s_lshl_b32 s24, s21, 2\n v_cvt_f32_ubyte1 %3, %5\n s_sub_u32 s25, s21, s22")
        if (MODE == 2) BODY("v_fma_f32 %0, %4, %5, %0\n v_cvt_f32_ubyte0 %1, %4\n s_mov_b64 s[20:21], exec\n s_and_b64 exec, exec, s[20:21]\n v_fma_f32 %2, %4, %5, %2\n v_cvt_f32_ubyte1 %3, %5\n s_mov_b64 exec, s[20:21]")
        if (MODE == 3) BODY("ds_read_b32 %1, %7\n v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %2, %4, %5, %2\n s_waitcnt lgkmcnt(0)\n v_cvt_f32_ubyte0 %3, %1\n v_cvt_f32_ubyte1 %1, %5")
        if (MODE == 4) BODY("v_fma_f32 %0, %4, %5, %0\n v_cvt_f32_ubyte0 %1, %4\n s_nop 0\n v_fma_f32 %2, %4, %5, %2\n v_cvt_f32_ubyte1 %3, %5\n s_nop 0")
        if (MODE == 5) BODY("v_fma_f32 %0, %4, %5, %0\n v_cvt_f32_ubyte0 %1, %4\n s_cbranch_scc0 1f\n s_nop 0\n1:\n v_fma_f32 %2, %4, %5, %2\n v_cvt_f32_ubyte1 %3, %5")
        if (MODE == 6) BODY("v_fmac_f32 %0, %4, %5\n v_cvt_f32_ubyte0 %1, %4\n v_mul_f32 %2, 0x3b808081, %2\n v_cvt_f32_ubyte1 %3, %5")
        if (MODE == 7) BODY("v_fma_f32 %0, %4, %6, %0\n v_cvt_f32_ubyte0 %1, %4\n v_fma_f32 %2, %4, %6, %2\n v_cvt_f32_ubyte1 %3, %5")
        if (MODE == 8) BODY("v_fma_f32 %0, %4, %5, %0\n v_cmp_lt_f32 vcc, %4, %1\n v_fma_f32 %2, %4, %5, %2\n v_cndmask_b32 %3, %4, %5, vcc")
        if (MODE == 9) BODY("v_fma_f32 %0, %4, %5, %0\n v_readlane_b32 s20, %4, 3\n v_fma_f32 %2, %4, %5, %2\n v_cvt_f32_ubyte1 %3, %5")
    }
    out[(blockIdx.x * BLOCK + threadIdx.x) & 0xfffff] = a + b + c + d + lds[(threadIdx.x + 1) & 1023];
}
template <int MODE, int BLOCK> void run(const char* name, float* out, int nvalu) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 300, grid = 256 * 8 * (256 / BLOCK);
    hipLaunchKernelGGL((k<MODE, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, out, 10, 1.0001f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, out, iters, 1.0001f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double valu_per_simd = 8.0 * iters * 64 * nvalu;
    printf("%-58s block %3d %8.3f ms  %6.2f cycles per VALU instruction per SIMD (if 2.4 GHz)\n", name, BLOCK, ms, ms * 1e-3 * 2.4e9 / valu_per_simd);
}
int main() {
    float* out; (void)hipMalloc(&out, (1 << 20) * 4);
    run<0, 256>("fma cvt fma cvt", out, 4); run<0, 64>("fma cvt fma cvt", out, 4);
    run<1, 256>("+ one SALU after every VALU", out, 4); run<2, 256>("+ exec mask saved, narrowed, restored", out, 4);
    run<3, 256>("+ ds_read and wait (2 fma 2 cvt per read)", out, 4); run<4, 256>("+ s_nop after every two", out, 4);
    run<5, 256>("+ a branch after every two", out, 4); run<6, 256>("fmac (VOP2) / mul with literal", out, 4);
    run<7, 256>("fma with an SGPR operand", out, 4); run<8, 256>("fma cmp fma cndmask", out, 4); run<9, 256>("fma readlane fma cvt", out, 4);
    return 0;
}
