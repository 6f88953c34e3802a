// Where the time of the LZ4F compression kernel goes: compresses one captured frame per stream (every stream its own
// copy of the bytes) and prints the kernel's duration, a checksum of stream 0's output and, per phase of a sequence,
// the share of the waves' cycles (stamps compiled in with RH_LZ4_STAMPS).
//   hipcc --offload-arch=gfx950 -O3 -DRH_LZ4_STAMPS -I renderer_amd/csrc tools/probe_lz4.hip -o tools/probe_lz4
//   tools/probe_lz4 <file with raw frame bytes> [streams=1024] [MiB per stream=8]
// Without a file the bytes are synthetic: RGBA pixels with constant alpha whose colour repeats the previous pixel's
// about every third pixel, which is what a blended 4K frame of noise-textured quads looks like to LZ4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "rh_codec.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "";
    const uint32_t n_streams = argc > 2 ? (uint32_t)atoi(argv[2]) : 1024u;
    const size_t bytes = (size_t)(argc > 3 ? atoi(argv[3]) : 8) << 20;
    std::vector<uint8_t> frame(bytes);
    FILE* f = *path ? fopen(path, "rb") : nullptr;
    if (f) {
        fseek(f, 128, SEEK_SET);                                  // (an .npy header; harmless for a raw file)
        if (fread(frame.data(), 1, bytes, f) != bytes) { fprintf(stderr, "short file\n"); return 1; }
        fclose(f);
    } else {
        uint64_t x = 88172645463325252ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 11); };
        for (size_t i = 0; i < bytes; i += 4) {
            const bool again = i >= 4 && rnd() % 3u == 0u;
            for (int c = 0; c < 3; c++) frame[i + c] = again ? frame[i - 4 + c] : (uint8_t)rnd();
            frame[i + 3] = 255;
        }
    }
    uint8_t *d_src, *d_out; rh_lz4_stream_state* d_states; rh_lz4_job* d_jobs; unsigned long long* d_len;
    const size_t src_stride = bytes + 256, out_stride = bytes + bytes / 255 + 65536;
    CHECK(hipMalloc(&d_src, src_stride * n_streams + 64)); CHECK(hipMalloc(&d_out, out_stride * n_streams));
    CHECK(hipMalloc(&d_states, sizeof(rh_lz4_stream_state) * n_streams)); CHECK(hipMalloc(&d_jobs, sizeof(rh_lz4_job) * n_streams));
    CHECK(hipMalloc(&d_len, 8 * n_streams));
    CHECK(hipMemset(d_states, 0, sizeof(rh_lz4_stream_state) * n_streams));
    CHECK(hipMemcpy(d_src, frame.data(), bytes, hipMemcpyHostToDevice));
    for (uint32_t i = 1; i < n_streams; i++) CHECK(hipMemcpy(d_src + src_stride * i, d_src, bytes, hipMemcpyDeviceToDevice));
    std::vector<rh_lz4_job> jobs(n_streams);
    for (uint32_t i = 0; i < n_streams; i++) {
        memset(&jobs[i], 0, sizeof(rh_lz4_job));
        jobs[i].stream = i; jobs[i].n_writes = 1; jobs[i].write_ptr[0] = d_src + src_stride * i; jobs[i].write_len[0] = bytes;
        jobs[i].out = d_out + out_stride * i; jobs[i].out_len = d_len + i;
    }
    CHECK(hipMemcpy(d_jobs, jobs.data(), sizeof(rh_lz4_job) * n_streams, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipMemset(d_states, 0, sizeof(rh_lz4_stream_state) * n_streams));
#ifdef RH_LZ4_STAMPS
        unsigned long long zero[24] = {0}; CHECK(hipMemcpyToSymbol(HIP_SYMBOL(rh_lz4_stamp_acc), zero, sizeof(zero)));
#endif
        CHECK(hipEventRecord(e0, 0));
        if (rhk::lz4f_compress_launch(d_jobs, n_streams, d_states, 0) != 0) { fprintf(stderr, "launch failed\n"); return 1; }
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    unsigned long long len0; CHECK(hipMemcpy(&len0, d_len, 8, hipMemcpyDeviceToHost));
    std::vector<uint8_t> out(len0); CHECK(hipMemcpy(out.data(), d_out, len0, hipMemcpyDeviceToHost));
    uint64_t h = 1469598103934665603ull; for (uint8_t b : out) { h ^= b; h *= 1099511628211ull; }
    printf("streams %u  MiB/stream %zu  kernel %.2f ms  (%.1f MB/s per stream, %.1f GB/s in)  out %llu bytes (%.4f)  fnv %016llx\n", n_streams, bytes >> 20, best,
           bytes / 1e3 / best, bytes * (double)n_streams / 1e6 / best, len0, (double)len0 / bytes, (unsigned long long)h);
#ifdef RH_LZ4_STAMPS
    unsigned long long acc[24]; CHECK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(rh_lz4_stamp_acc), sizeof(acc)));
    const char* names[12] = {"window + sequence + hash", "election + candidate bytes", "resolution + inserts", "short form", "general loads + catch-up", "emit + match length", "block frame, last literals, dictionary", "straight piece: up to the hits", "straight piece: lengths + sequence", "straight piece: given up", "-", "-"};
    double total = 0; for (int k = 0; k < 12; k++) total += (double)acc[k];
    for (int k = 0; k < 10; k++) printf("  %-40s %5.1f %%   %7.0f cycles per KiB of input  %6.2f times per KiB, %5.0f cycles each\n", names[k], 100.0 * acc[k] / total, (double)acc[k] / n_streams / (bytes / 1024.0),
                                        (double)acc[12 + k] / n_streams / (bytes / 1024.0), acc[12 + k] ? (double)acc[k] / acc[12 + k] : 0.0);
#endif
    return 0;
}
