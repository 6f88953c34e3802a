// Probe: the chained sorting pass (rh_sort_pass) on its own: 2.25 M random (tile, cluster) pairs over 8160 tiles, checked
// against a stable CPU sort, timed, with wave clock stamps of its phases for a few parts.
//   hipcc --offload-arch=gfx950 -O3 -DRH_SORT_STAMPS -I renderer_amd/csrc tools/probe_sort.hip -o tools/probe_sort && tools/probe_sort
#include "../renderer_amd/csrc/rh_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <numeric>

__global__ void count_digits(const uint32_t* keys, uint32_t n, rh_bin_state* st, uint32_t passes) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (uint32_t p = 0; p < passes; p++) atomicAdd(&st->digit_count[p][blockIdx.x % RH_DIGIT_COPIES][(keys[i] >> (8 * p)) & 255u], 1u);
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 2250000u, tiles = 8160, passes = 2;
    std::vector<uint32_t> hk(n), hv(n);
    uint32_t seed = 12345;
    for (uint32_t i = 0; i < n; i++) { seed = seed * 1664525u + 1013904223u; hk[i] = (seed >> 8) % tiles; hv[i] = i; }
    uint32_t *ka, *va, *kb, *vb, *status; rh_bin_state* st; unsigned long long* host_total;
    const uint32_t parts = rhk::sort_parts_for(n);
    (void)hipMalloc(&ka, n * 4); (void)hipMalloc(&va, n * 4); (void)hipMalloc(&kb, n * 4); (void)hipMalloc(&vb, n * 4);
    const size_t status_bytes = (size_t)passes * rhk::sort_status_words(n) * 4;
    (void)hipMalloc(&status, status_bytes); (void)hipMalloc(&st, sizeof(rh_bin_state));
    (void)hipHostMalloc(&host_total, 16); host_total[0] = host_total[1] = 0;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t resident = argc > 2 ? (uint32_t)atoi(argv[2]) : rhk::sort_pass_resident_parts(256);
    printf("resident parts %u\n", resident);
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        (void)hipMemcpy(ka, hk.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(va, hv.data(), n * 4, hipMemcpyHostToDevice);
        (void)hipMemset(status, 0, status_bytes); (void)hipMemset(st, 0, sizeof(rh_bin_state));
        unsigned long long total = n; (void)hipMemcpy(&st->total, &total, 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(count_digits, dim3((n + 255) / 256), dim3(256), 0, 0, ka, n, st, passes);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        const int where = rhk::sort_passes_chained(ka, va, kb, vb, n, 13, st, status, host_total, resident, 0);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        if (rep == 0) {
            std::vector<uint32_t> gk(n), gv(n);
            (void)hipMemcpy(gk.data(), where ? kb : ka, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(gv.data(), where ? vb : va, n * 4, hipMemcpyDeviceToHost);
            std::vector<uint32_t> idx(n); std::iota(idx.begin(), idx.end(), 0u);
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return hk[a] < hk[b]; });
            size_t bad = 0; for (uint32_t i = 0; i < n; i++) if (gk[i] != hk[idx[i]] || gv[i] != idx[i]) bad++;
            printf("n %u parts %u: %zu of %u pairs differ from the stable CPU sort, fault %llu\n", n, parts, bad, n, host_total[1]);
        }
    }
    printf("two passes: best %.1f us\n", best * 1000);
#ifdef RH_SORT_STAMPS
    std::vector<unsigned long long> stamps(4096 * 8);
    (void)hipMemcpyFromSymbol(stamps.data(), HIP_SYMBOL(rh_sort_stamps), stamps.size() * 8);
    unsigned long long t0 = ~0ull; for (uint32_t p = 0; p < parts; p++) t0 = std::min(t0, stamps[p * 8]);
    printf("part: start, keys counted, local sort done, look-back done, barrier, written (us after the first start, 100 MHz clock); hops\n");
    for (uint32_t p = 0; p < parts; p += std::max(1u, parts / 24)) {
        printf("%4u:", p); for (int k = 0; k < 6; k++) printf(" %7.2f", (double)(stamps[p * 8 + k] - t0) / 100.0); printf("  %llu\n", stamps[p * 8 + 6]);
    }
#endif
    return 0;
}
