// Probe: cost of folding per-workgroup digit histograms into one global histogram with atomic adds nobody waits for
// (B workgroups x 256 threads, every thread adds to K of 256*K words), against the same kernel without the adds.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_atomic.hip -o tools/probe_atomic && tools/probe_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K, bool NONZERO_ONLY> __global__ __launch_bounds__(256) void k(uint32_t* g, uint32_t* sink) {
    __shared__ uint32_t h[K ? K : 1][256]; if (K == 0) h[0][threadIdx.x] = threadIdx.x;
    for (int p = 0; p < K; p++) h[p][threadIdx.x] = (threadIdx.x * 7u + blockIdx.x + p) & 3u;
    __syncthreads();
    for (int p = 0; p < K; p++) {
        const uint32_t v = h[p][threadIdx.x];
        if (!NONZERO_ONLY || v) atomicAdd(&g[p * 256 + threadIdx.x], v);
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = h[0][1];
}
template <int K, bool NZ> void run(const char* name, uint32_t* g, uint32_t* sink, int blocks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<K, NZ>), dim3(blocks), dim3(256), 0, 0, g, sink);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<K, NZ>), dim3(blocks), dim3(256), 0, 0, g, sink);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %5d workgroups  %7.2f us per launch\n", name, blocks, ms * 1000 / 20);
}
int main() {
    uint32_t *g, *sink; (void)hipMalloc(&g, 4 * 256 * 4); (void)hipMalloc(&sink, 65536 * 4); (void)hipMemset(g, 0, 4096);
    for (int blocks : {256, 1000, 4000, 16000}) {
        run<0, false>("no adds", g, sink, blocks);
        run<1, false>("1 x 256 words", g, sink, blocks);
        run<2, false>("2 x 256 words", g, sink, blocks);
        run<2, true>("2 x 256 words, 3 of 4", g, sink, blocks);
        run<3, false>("3 x 256 words", g, sink, blocks);
    }
    return 0;
}
