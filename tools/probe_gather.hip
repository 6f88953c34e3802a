// Probe: the texture fetch pattern of a 16:1 minified bilinear sample on its own.  Every lane of a pass reads two
// 8-byte texel pairs (rows y and y+1) 64 bytes apart from its neighbour's, i.e. every fragment touches 128-byte lines
// nobody else in the pass needs.  How many such passes per second does a CU sustain from a 256 KiB texture (L2
// resident, far larger than the 32 KiB L1), and do the cache-policy bits change it?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_gather.hip -o tools/probe_gather && tools/probe_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE> __device__ __forceinline__ uint2 ld8(const uint8_t* base, uint32_t off) {
    uint2 r;
    if (MODE == 0) r = *reinterpret_cast<const uint2*>(base + off);
    if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, %2 nt\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(off), "s"(base) : "memory");
    if (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, %2 sc1\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(off), "s"(base) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, %2 sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(off), "s"(base) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx2 %0, %1, %2 sc0\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}

// stride_x: bytes between neighbouring fragments of a row (64 = 16 texels); W: texture width in texels
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t* tex, uint32_t W, uint32_t H, uint32_t stride_x, uint32_t stride_y, int passes, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    uint32_t seed = blockIdx.x * 977u + (threadIdx.x >> 6) * 131u + 7u;
    uint32_t acc = 0;
    const uint32_t pitch = W * 4;
    for (int p = 0; p < passes; p++) {
        seed = seed * 1664525u + 1013904223u;
        const uint32_t px = (seed >> 8) % 16u, py = (seed >> 16) % 16u, fr = ((seed >> 24) & 3u) * 4u;      // sub-pixel phase, which quarter of the quad
        const uint32_t row = ((fr + (uint32_t)lane / 16u) * stride_y + py) % (H - 1), col = ((uint32_t)lane % 16u) * stride_x + px * 4u;
        const uint32_t off = row * pitch + (col % (pitch - 8));
        const uint2 a = ld8<MODE>(tex, off), b = ld8<MODE>(tex, off + pitch);
        acc += a.x ^ a.y ^ b.x ^ b.y;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE> void run(const char* name, const uint8_t* tex, uint32_t W, uint32_t sx, uint32_t sy, uint32_t* out, int waves) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int passes = 4000, grid = 256 * waves;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, tex, W, W, sx, sy, 10, out);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, tex, W, W, sx, sy, passes, out);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double passes_per_cu = (double)grid / 256.0 * 4 * passes;
    printf("%-10s tex %4u stride %3u/%2u rows  %d waves/SIMD %8.3f ms  %6.1f ns/pass/CU  %6.1f ns/pass/SIMD\n", name, W, sx, sy, waves, ms,
           ms * 1e6 / passes_per_cu, ms * 1e6 / passes_per_cu * 4);
}

int main() {
    uint32_t* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (uint32_t W : {256u, 64u, 1024u}) {
        std::vector<uint8_t> h((size_t)W * W * 4, 1);
        uint8_t* tex; (void)hipMalloc(&tex, h.size()); (void)hipMemcpy(tex, h.data(), h.size(), hipMemcpyHostToDevice);
        const uint32_t sx = W == 64u ? 16u : 64u, sy = W == 64u ? 4u : 16u;
        for (int waves : {8, 4}) {
            run<0>("plain", tex, W, sx, sy, out, waves); run<1>("nt", tex, W, sx, sy, out, waves); run<2>("sc1", tex, W, sx, sy, out, waves);
            run<3>("sc0 sc1", tex, W, sx, sy, out, waves); run<4>("sc0", tex, W, sx, sy, out, waves);
        }
        if (W == 256u) { run<0>("plain", tex, W, 4u, 1u, out, 8); run<0>("plain", tex, W, 8u, 2u, out, 8); run<0>("plain", tex, W, 16u, 4u, out, 8); run<0>("plain", tex, W, 32u, 8u, out, 8); }
        (void)hipFree(tex);
    }
    return 0;
}
